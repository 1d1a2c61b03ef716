#!/bin/bash
# every GPU test + smoke (what the driver runs at round end), then the driver's bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05tests; rm -rf $O; mkdir -p $O; rm -f gpurun_out/parity_log.jsonl
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "== smoke exit $?"; tail -n 2 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "== pytest exit $?"; tail -n 8 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "== bench exit $?"; tail -n 1 $O/bench.log | cut -c1-600
