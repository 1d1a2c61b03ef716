#!/bin/bash
# Round-6 session 3: (1) same-box A/B of the GroupNorm fusion policy after r5's cheaper fused transform: apply fused into convs with up to 1 (default) / 2 / 4
# output slices (GENPERCEPT_GN_FUSE_MAX_SLICES), alternating; (2) the SQ-counter collection script, to validate it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r06s3
rm -rf $O; mkdir -p $O
for E in "d1:" "s2:GENPERCEPT_GN_FUSE_MAX_SLICES=2" "s4:GENPERCEPT_GN_FUSE_MAX_SLICES=4" "d2:" "s2b:GENPERCEPT_GN_FUSE_MAX_SLICES=2"; do
  env ${E#*:} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-fp16 --no-fp32c 2>&1 | tail -1 > $O/bench_${E%%:*}.log
  python3 -c "import json; d=json.load(open('$O/bench_${E%%:*}.log')); print('ab ${E%%:*}', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['stages']['ms_encode'], d['stages']['ms_unet'], d['stages']['ms_head'], d['stages']['kernel_launches'], d['clock_power']['sclk_mhz_mean'])"
done
bash tools/sessions/gpu_r06_pmc.sh
