#!/bin/bash
# round-2 GPU session F: full GPU suite, launch log, bench (after the flash XCD remap)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_log.jsonl
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "== pytest exit $?"; tail -n 30 gpurun_out/pytest_gpu.log
timeout 600 python tools/launch_log.py --tag r02f > gpurun_out/launch_log_run.log 2>&1
echo "== launch_log exit $?"; tail -n 24 gpurun_out/launch_log_run.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/bench.log 2>&1
echo "== bench exit $?"; tail -n 1 gpurun_out/bench.log | cut -c1-250; grep -o '"stages".*' gpurun_out/bench.log
