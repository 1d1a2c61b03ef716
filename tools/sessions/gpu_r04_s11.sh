#!/bin/bash
# Round-4 GPU session 11: (1) is LDS array time additive to the MFMA time?  conv3x3_halo3_kernel with every fragment read issued twice (dbg bit 30,
# tools/kbench_abl) beside the plain kernel; (2) pipeline A/B: GENPERCEPT_HALO5=1 (two workgroups per CU where the tile count allows) vs default
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s11; rm -rf $O; mkdir -p $O
S="conv:4,768,768,128,128 conv:4,384,384,256,256 conv:4,192,192,512,512"
for rep in 1 2; do
  for V in "base:0" "double_frag:$((1<<30))"; do
    echo "== ${V%%:*} rep $rep"; GENPERCEPT_IGEMM_DBG=${V##*:} timeout 200 tools/kbench_abl iters=20 cold=1 check=0 $S | grep -vE "^#" | tee -a $O/kbench_${V%%:*}.log
  done
done
for E in "default:" "halo5:GENPERCEPT_HALO5=1" "default2:" "halo5_2:GENPERCEPT_HALO5=1"; do
  echo "== bench ${E%%:*}"; env ${E#*:} timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-fp16 2>&1 | tail -1 > $O/bench_${E%%:*}.log
  python3 -c "import json,sys; d=json.load(open('$O/bench_${E%%:*}.log')); print(d['value'], d['ms_per_step'], d['stages'], d['roofline']['achieved'], d['parity']['bf16'] if 'parity' in d else '')" | cut -c1-400
done
