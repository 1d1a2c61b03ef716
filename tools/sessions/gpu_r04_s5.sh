#!/bin/bash
# Round-4 GPU session 5: 4-deep ring of the 128-row persistent GEMM: GEMM tests, kbench A/B (ring 3 vs 4) on the UNet's shapes, pipeline A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s5; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=300 -p no:cacheprovider -k "gemm or qkv or geglu or statistics" > $O/pytest_gemm.log 2>&1
echo "== gemm tests exit $?"; tail -n 3 $O/pytest_gemm.log
S="gemm:2304,1280,1280 gemm:2304,1280,1280,0,1 gemm:9216,640,640 gemm:9216,640,640,0,1 gemm:2304,1280,5120,0,1 gemm:9216,640,2560,0,1 gemm:2304,10240,1280,3 gemm:9216,5120,640,3 gemm:576,10240,1280,3 gemm:36864,320,320,0,1 gemm:2304,1280,2560 gemm:9216,640,1920"
for rep in 1 2; do
  for V in "ring3:$((1<<23))" "ring4:0"; do
    echo "== ${V%%:*} rep $rep"; GENPERCEPT_IGEMM_DBG=${V##*:} timeout 200 tools/kbench iters=30 cold=1 check=$((2-rep)) $S | grep -vE "^#" | tee -a $O/kbench_${V%%:*}.log
  done
done
for E in "ring4:" "ring3:GENPERCEPT_PGEMM_RING3=1" "ring4b:" "ring3b:GENPERCEPT_PGEMM_RING3=1"; do
  echo "== bench ${E%%:*}"; env ${E#*:} timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-fp16 2>&1 | tail -1 > $O/bench_${E%%:*}.log
  python3 -c "import json,sys; d=json.load(open('$O/bench_${E%%:*}.log')); print(d['value'], d['ms_per_step'], d['stages'])"
done
