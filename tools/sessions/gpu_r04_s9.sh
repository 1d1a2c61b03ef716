#!/bin/bash
# Round-4 GPU session 9: conv3x3_halo5_kernel (two workgroups per CU): parity tests, then kbench against conv3x3_halo3_kernel, interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s9; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "halo5" --timeout=300 -p no:cacheprovider > $O/pytest_halo5.log 2>&1; echo "== halo5 tests exit $?"; tail -n 8 $O/pytest_halo5.log
S="conv:4,768,768,128,128 conv:4,384,384,256,256 conv:4,192,192,512,512 conv:4,96,96,512,512 conv:4,768,768,256,128 conv:4,96,96,320,320 conv:4,96,96,640,320"
for rep in 1 2; do
  for V in "halo3:$((2<<28))" "halo5:$((1<<28))"; do
    echo "== ${V%%:*} rep $rep"; GENPERCEPT_IGEMM_DBG=${V##*:} timeout 200 tools/kbench iters=20 cold=1 check=$((rep==1)) $S | grep -vE "^#" | tee -a $O/kbench_${V%%:*}.log
  done
done
