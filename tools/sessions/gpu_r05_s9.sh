#!/bin/bash
# Round-5 GPU session 9: the x2-upsample convs as four phase convolutions (conv3x3_halo3_kernel<..., PH>): kernel parity tests, the e2e / full-size tests,
# and a pipeline A/B against the nine-tap upsample kernel on the same library (GENPERCEPT_NO_UP_PHASES=1), alternating x3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05s9; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "upsample" --timeout=500 -p no:cacheprovider > $O/pytest_k.log 2>&1; echo "== upsample kernel tests exit $?"; tail -n 12 $O/pytest_k.log
for E in "phases:" "ninetap:GENPERCEPT_NO_UP_PHASES=1" "phases2:" "ninetap2:GENPERCEPT_NO_UP_PHASES=1" "phases3:" "ninetap3:GENPERCEPT_NO_UP_PHASES=1"; do
  env ${E#*:} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-fp16 2>&1 | tail -1 > $O/bench_${E%%:*}.log
  python3 -c "import json; d=json.load(open('$O/bench_${E%%:*}.log')); print('${E%%:*}', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['stages']['ms_encode'], d['stages']['ms_unet'], d['stages']['ms_head'], d['clock_power']['sclk_mhz_mean'], d['roofline']['sum_ms'], d['roofline']['frac'])"
done
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_fullsize_parity_gpu.py tests/test_refexec_gpu.py -q --timeout=800 -p no:cacheprovider > $O/pytest_e2e.log 2>&1; echo "== e2e / full-size / refexec exit $?"; tail -n 6 $O/pytest_e2e.log
