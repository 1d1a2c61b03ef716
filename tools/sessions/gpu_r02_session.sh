#!/bin/bash
# One round-2 GPU-box session: every GPU test (both element-type libraries), the per-launch log of the benched configuration, the bench line.
# usage (from the build container): gpurun --timeout 3300 -- 'bash tools/gpu_r02_session.sh [pytest -k expression]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_log.jsonl
if [ -n "$1" ]; then K=(-k "$1"); else K=(); fi
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider "${K[@]}" > gpurun_out/pytest_gpu.log 2>&1
echo "== pytest exit $?"; tail -n 30 gpurun_out/pytest_gpu.log
timeout 600 python tools/launch_log.py --tag r02 > gpurun_out/launch_log_run.log 2>&1
echo "== launch_log exit $?"; tail -n 24 gpurun_out/launch_log_run.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
echo "== bench exit $?"; tail -n 1 gpurun_out/bench.log
