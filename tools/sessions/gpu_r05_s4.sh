#!/bin/bash
# Round-5 GPU session 4: the bench line once more on the collected build (now with profiles/r05_pmc_traffic_summary.json present: roofline.traffic non-null),
# and the one-rank RCCL test
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05s4; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_rccl_single_rank_gpu.py -q --timeout=280 -p no:cacheprovider > $O/pytest_rccl.log 2>&1; echo "== rccl test exit $?"; tail -n 5 $O/pytest_rccl.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "== bench exit $?"; tail -n 1 $O/bench.log | cut -c1-400
python3 -c "
import json; d=json.loads(open('$O/bench.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_note'][:120], d['clock_power'])"
