#!/bin/bash
# Round-4 GPU session 7: scheduling experiments on the dominant kernel (conv3x3_halo3_kernel, plain): front-loaded fragment reads (dbg bit 24),
# static priority for waves 4-7 (bit 25), both; kbench interleaved, three repetitions
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s7; rm -rf $O; mkdir -p $O
S="conv:4,384,384,256,256 conv:4,192,192,512,512 conv:4,96,96,512,512 conv:4,768,768,256,128 conv:4,96,96,320,320"
for rep in 1 2 3; do
  for V in "base:0" "front:$((1<<24))" "prio:$((2<<24))" "both:$((3<<24))"; do
    echo "== ${V%%:*} rep $rep"; GENPERCEPT_IGEMM_DBG=${V##*:} timeout 200 tools/kbench iters=20 cold=1 check=$((rep==1)) $S | grep -vE "^#" | tee -a $O/kbench_${V%%:*}.log
  done
done
