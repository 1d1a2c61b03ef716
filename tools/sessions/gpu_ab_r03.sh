#!/bin/bash
# same-box A/B of this round's switches through the whole pipeline (bench.py, 20 steps, stage times from a level-1 pass)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-fp16 2>&1 | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); s=l['stages']
print('%-36s %7.3f img/s  %7.3f ms  enc %.3f unet %.3f head %.3f' % ('$*', l['value'], l['ms_per_step'], s['ms_encode'], s['ms_unet'], s['ms_head']))"; }
for r in 1 2; do
run A=default
run GENPERCEPT_XFOLD_LDS=0
run GENPERCEPT_NO_QKV_FUSE=1
run GENPERCEPT_NO_CONV_FEW=1
run GENPERCEPT_GN_APPLY_OLD=1
run GENPERCEPT_QKV_FUSE_MAX_ROWS=100000
done 2>&1 | tee gpurun_out/ab_r03.log
