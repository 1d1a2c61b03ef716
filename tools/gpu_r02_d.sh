#!/bin/bash
# round-2 GPU session D: cross-fold + e2e tests, launch log, full bench line (CPU baseline at 768, MFMA peak)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -k "cross or stages_vs_golden or infer_vs_golden or full_sd21" > gpurun_out/pytest_gpu_d.log 2>&1
echo "== pytest exit $?"; tail -n 12 gpurun_out/pytest_gpu_d.log
timeout 600 python tools/launch_log.py --tag r02d > gpurun_out/launch_log_run.log 2>&1
echo "== launch_log exit $?"; tail -n 24 gpurun_out/launch_log_run.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
echo "== bench exit $?"; tail -n 2 gpurun_out/bench.log
