#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-fp16 2>&1 | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); s=l['stages']
print('%-36s %7.3f img/s  %7.3f ms  enc %.3f unet %.3f head %.3f util %.4f' % ('$*', l['value'], l['ms_per_step'], s['ms_encode'], s['ms_unet'], s['ms_head'], s['unet_mfma_util']))"; }
for r in 1 2; do
run A=default
run GENPERCEPT_NO_CONV_IMG=1
done 2>&1 | tee gpurun_out/ab2_r03.log
