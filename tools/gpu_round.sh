#!/bin/bash
# One GPU-box session: parity tests (all, not -x), short bench, optional rocprof.  Logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_log.jsonl
(rocminfo | grep -E "gfx|Marketing" | head -4; nproc; free -g | head -2) > gpurun_out/env.log 2>&1
WHAT="${1:-all}"
if [[ "$WHAT" == "all" || "$WHAT" == *kernels* ]]; then
  timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=300 -p no:cacheprovider > gpurun_out/pytest_kernels.log 2>&1
  echo "== kernels exit $?"; tail -n 45 gpurun_out/pytest_kernels.log
fi
if [[ "$WHAT" == "all" || "$WHAT" == *e2e* ]]; then
  timeout 1800 python -m pytest tests/test_e2e_gpu.py -m gpu -q --timeout=900 -p no:cacheprovider > gpurun_out/pytest_e2e.log 2>&1
  echo "== e2e exit $?"; tail -n 45 gpurun_out/pytest_e2e.log
fi
if [[ "$WHAT" == "all" || "$WHAT" == *bench* ]]; then
  timeout 1500 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1
  echo "== bench exit $?"; tail -n 12 gpurun_out/bench.log
fi
