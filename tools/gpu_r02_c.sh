#!/bin/bash
# round-2 GPU session C: targeted tests (cross fold, small GroupNorm, e2e stages), launch log, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -k "cross or groupnorm or stages_vs_golden or infer_vs_golden or infer_dpt or full_sd21" > gpurun_out/pytest_gpu_c.log 2>&1
echo "== pytest exit $?"; tail -n 30 gpurun_out/pytest_gpu_c.log
timeout 600 python tools/launch_log.py --tag r02c > gpurun_out/launch_log_run.log 2>&1
echo "== launch_log exit $?"; tail -n 30 gpurun_out/launch_log_run.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/bench.log 2>&1
echo "== bench exit $?"; tail -n 2 gpurun_out/bench.log | cut -c1-200; grep -o '"stages".*' gpurun_out/bench.log
