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
if [[ "$WHAT" == *prof* ]]; then
  export TMPDIR=/tmp
  ROOTD=$(pwd)
  (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof" -- python "$ROOTD/bench.py" --steps 2 --warmup 1 --no-cpu --no-profile > "$ROOTD/gpurun_out/prof_bench.log" 2>&1)
  echo "== prof exit $?"; tail -n 3 gpurun_out/prof_bench.log
  find gpurun_out/prof -name "*kernel_stats*" | head; F=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -40 "$F"
  # keep the merge small: drop the raw trace, keep stats
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
