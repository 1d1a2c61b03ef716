#!/bin/bash
# Round-5 GPU session 6: packed-fp32 GELU in the GEGLU epilogue of pgemm_kernel: GEGLU tests, kbench of the three GEGLU shapes against the library built
# before the change (genpercept_amd/lib/base via LD_LIBRARY_PATH), interleaved; pipeline A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTD=$(pwd); O=gpurun_out/r05s6; rm -rf $O; mkdir -p $O
G="gemm:36864,2560,320,3 gemm:9216,5120,640,3 gemm:2304,10240,1280,3"
for rep in 1 2 3; do
  for V in new base; do
    E=""; [ $V = base ] && E="LD_LIBRARY_PATH=$ROOTD/genpercept_amd/lib/base"
    env $E timeout 120 tools/kbench iters=40 cold=1 check=$((rep==1)) $G | grep -vE "^#" | sed "s/^/$V /" | tee -a $O/kbench_geglu.log
  done
done
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "geglu or gemm" --timeout=500 -p no:cacheprovider > $O/pytest_k.log 2>&1; echo "== tests exit $?"; tail -n 3 $O/pytest_k.log
for E in "new:" "base:GENPERCEPT_HIP_LIB=$ROOTD/genpercept_amd/lib/base/libgenpercept_hip.so" "new2:" "base2:GENPERCEPT_HIP_LIB=$ROOTD/genpercept_amd/lib/base/libgenpercept_hip.so"; do
  env ${E#*:} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-fp16 2>&1 | tail -1 > $O/bench_${E%%:*}.log
  python3 -c "import json; d=json.load(open('$O/bench_${E%%:*}.log')); print('${E%%:*}', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['stages']['ms_encode'], d['stages']['ms_unet'], d['stages']['ms_head'], d['clock_power']['sclk_mhz_mean'])"
done
