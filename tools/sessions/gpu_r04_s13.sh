#!/bin/bash
# Round-4 GPU session 13: conv3x3_halo3_kernel with its LDS-DMA through buffer resources (default now) vs the global_load_lds form of r2 / r3
# (dbg bit 31 in GP_HALO_ABLATIONS builds: tools/kbench_abl); conv parity tests on the product library first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s13; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "conv" --timeout=600 -p no:cacheprovider > $O/pytest_conv.log 2>&1; echo "== conv tests exit $?"; tail -n 5 $O/pytest_conv.log
S="conv:4,768,768,128,128 conv:4,384,384,256,256 conv:4,192,192,512,512 conv:4,96,96,512,512 conv:4,768,768,256,128 conv:4,96,96,320,320 conv:4,96,96,640,320"
for rep in 1 2 3; do
  for V in "mubuf:0" "flat:-2147483648"; do
    echo "== ${V%%:*} rep $rep"; GENPERCEPT_IGEMM_DBG=${V##*:} timeout 200 tools/kbench_abl iters=20 cold=1 check=$((rep==1)) $S | grep -vE "^#" | tee -a $O/kbench_${V%%:*}.log
  done
done
