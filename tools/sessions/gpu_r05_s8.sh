#!/bin/bash
# Round-5 GPU session 8: residual rows of a tile requested before the MFMAs of the tile's last step (res_prefetch) instead of at the start of the
# epilogue: conv parity tests (with residual), e2e tests, and a same-box pipeline A/B against the library built before the change, alternating x3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTD=$(pwd); O=gpurun_out/r05s8; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "conv" --timeout=500 -p no:cacheprovider > $O/pytest_k.log 2>&1; echo "== conv tests exit $?"; tail -n 3 $O/pytest_k.log
for E in "new:" "base:GENPERCEPT_HIP_LIB=$ROOTD/genpercept_amd/lib/base/libgenpercept_hip.so" "new2:" "base2:GENPERCEPT_HIP_LIB=$ROOTD/genpercept_amd/lib/base/libgenpercept_hip.so" "new3:" "base3:GENPERCEPT_HIP_LIB=$ROOTD/genpercept_amd/lib/base/libgenpercept_hip.so"; do
  env ${E#*:} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-fp16 2>&1 | tail -1 > $O/bench_${E%%:*}.log
  python3 -c "import json; d=json.load(open('$O/bench_${E%%:*}.log')); print('${E%%:*}', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['stages']['ms_encode'], d['stages']['ms_unet'], d['stages']['ms_head'], d['clock_power']['sclk_mhz_mean'], d['roofline']['sum_ms'])"
done
timeout 600 python -m pytest tests/test_e2e_gpu.py -q --timeout=500 -p no:cacheprovider -k "full_size or batch or stages" > $O/pytest_e2e.log 2>&1; echo "== e2e exit $?"; tail -n 3 $O/pytest_e2e.log
