#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "flash" > gpurun_out/pytest_gpu_g.log 2>&1; echo "== pytest exit $?"; tail -n 5 gpurun_out/pytest_gpu_g.log
for i in 1 2; do
timeout 300 python tools/attn_bench.py 2>&1 | grep "T="
GENPERCEPT_FLASH_RING2=1 timeout 300 python tools/attn_bench.py 2>&1 | grep "T="
done
