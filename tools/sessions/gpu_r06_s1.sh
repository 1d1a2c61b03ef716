#!/bin/bash
# Round-6 session 1: smoke, every GPU test (new: contract precision fp32c everywhere, outlier-channel stress, phase-kernel statistics), bench line with the fp32c leg.
# usage: gpurun --timeout 2700 -- 'bash tools/sessions/gpu_r06_s1.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r06s1
rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "== smoke exit $?"; tail -n 2 $O/smoke.log
rm -f gpurun_out/parity_log.jsonl
timeout 2000 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "== pytest exit $?"; tail -n 25 $O/pytest_gpu.log | cut -c1-240
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "== bench exit $?"; tail -n 1 $O/bench.log | cut -c1-600
