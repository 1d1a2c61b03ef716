#!/bin/bash
# Round-4 GPU session 4: halo4 with the next step's weight fragments read under the second MFMA batch (kbench A/B vs halo3) + its tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s4; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=300 -p no:cacheprovider -k "halo4" > $O/pytest_halo4.log 2>&1
echo "== halo4 tests exit $?"; tail -n 3 $O/pytest_halo4.log
S="conv:4,384,384,256,256 conv:4,768,768,256,128 conv:4,384,384,512,256 conv:8,192,192,512,512"
for rep in 1 2 3; do
  for V in "halo3:$((2<<20))" "halo4x2:$((1<<20))"; do
    echo "== ${V%%:*} rep $rep"; GENPERCEPT_IGEMM_DBG=${V##*:} timeout 200 tools/kbench iters=20 cold=1 check=0 $S | grep -vE "^#" | tee -a $O/kbench_${V%%:*}.log
  done
done
