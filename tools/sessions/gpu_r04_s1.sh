#!/bin/bash
# Round-4 GPU session 1: every GPU test (new: refexec fixture, both parity metrics at full size, builtin permlane swaps, cached switches) + the bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s1; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_log.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "== pytest exit $?"; tail -n 8 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "== bench exit $?"; tail -n 1 $O/bench.log | cut -c1-600
