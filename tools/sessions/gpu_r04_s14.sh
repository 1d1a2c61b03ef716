#!/bin/bash
# Round-4 GPU session 14: pgemm_kernel with its LDS-DMA through buffer resources (default now) vs global_load_lds (ABL 1 = GENPERCEPT_IGEMM_DBG=512);
# GEMM parity tests first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s14; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or linear or qkv or geglu or pointwise" --timeout=600 -p no:cacheprovider > $O/pytest_gemm.log 2>&1; echo "== gemm tests exit $?"; tail -n 4 $O/pytest_gemm.log
S="gemm:36864,320,320 gemm:36864,2560,320 gemm:36864,320,1280 gemm:9216,640,640 gemm:9216,5120,640 gemm:9216,640,2560 gemm:2304,1280,1280 gemm:2304,10240,1280 gemm:2304,1280,5120 gemm:589824,256,512 gemm:2359296,128,256"
for rep in 1 2 3; do
  for V in "mubuf:0" "flat:512"; do
    echo "== ${V%%:*} rep $rep"; GENPERCEPT_IGEMM_DBG=${V##*:} timeout 200 tools/kbench iters=20 cold=1 check=$((rep==1)) $S | grep -vE "^#" | tee -a $O/kbench_${V%%:*}.log
  done
done
