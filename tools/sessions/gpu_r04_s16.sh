#!/bin/bash
# Round-4 GPU session 16: conv3x3_halo6_kernel (Winograd F(2,3) along x): all parity tests, then the pipeline A/B (GENPERCEPT_WINO=1), alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s16; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "halo6" --timeout=300 -p no:cacheprovider > $O/pytest_halo6.log 2>&1; echo "== halo6 tests exit $?"; tail -n 6 $O/pytest_halo6.log | cut -c1-220
for E in "default:" "wino:GENPERCEPT_WINO=1" "default2:" "wino2:GENPERCEPT_WINO=1"; do
  env ${E#*:} timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-fp16 2>&1 | tail -1 > $O/bench_${E%%:*}.log
  python3 -c "import json; d=json.load(open('$O/bench_${E%%:*}.log')); print('${E%%:*}', d['value'], d['ms_per_step'], d['stages']['ms_encode'], d['stages']['ms_unet'], d['stages']['ms_head'], d['parity']['bf16']['mean_abs'] if d.get('parity') and d['parity'].get('bf16') else None)"
done
