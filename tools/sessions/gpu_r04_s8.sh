#!/bin/bash
# Round-4 GPU session 8: (1) device pre / post with bicubic resampling and float-tensor inputs (tests/test_prepost_gpu.py);
# (2) diagnostic for the next kernel: conv3x3_halo3_kernel (plain) without its LDS fragment reads -- weights (dbg bit 26), pixels (bit 27), both --
#     beside the r3 ablations "no MFMA" (abl 2) and "no epilogue" (abl 1024); tools/kbench_abl, interleaved, two repetitions
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s8; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_prepost_gpu.py -q --timeout=600 -p no:cacheprovider > $O/pytest_prepost.log 2>&1; echo "== prepost exit $?"; tail -n 5 $O/pytest_prepost.log
S="conv:4,768,768,128,128 conv:4,384,384,256,256 conv:4,192,192,512,512"
for rep in 1 2; do
  for V in "base:0" "no_wfrag:$((1<<26))" "no_xfrag:$((2<<26))" "no_frag:$((3<<26))" "no_mfma:$((2<<9))" "no_epilogue:$((1024<<9))"; do
    echo "== ${V%%:*} rep $rep"; GENPERCEPT_IGEMM_DBG=${V##*:} timeout 200 tools/kbench_abl iters=20 cold=1 check=0 $S | grep -vE "^#" | tee -a $O/kbench_${V%%:*}.log
  done
done
