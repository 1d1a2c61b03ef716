#!/bin/bash
# round-2 GPU session A: all GPU tests, per-launch log of the benched configuration, short bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_log.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "== pytest exit $?"; tail -n 25 gpurun_out/pytest_gpu.log
timeout 600 python tools/launch_log.py --tag r02a > gpurun_out/launch_log_run.log 2>&1
echo "== launch_log exit $?"; tail -n 30 gpurun_out/launch_log_run.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1
echo "== bench exit $?"; tail -n 3 gpurun_out/bench.log
