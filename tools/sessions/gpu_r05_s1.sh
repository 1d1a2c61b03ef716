#!/bin/bash
# Round-5 GPU session 1: (1) kbench A/B of the 12-row tile variant of conv3x3_halo3_kernel (dbg bits 20-21: 1 force, 2 forbid) on the 96 x 96 x 512 shapes it
# was built for and on two shapes it must NOT be chosen for; the fused GroupNorm+SiLU conv with packed fp32 math; where the short-K persistent GEMMs
# spend their time (pgemm ablations: 2 no MFMA, 4 no stores, 8 no DMA, 10, 14); (2) the parity tests of what changed this round; (3) a bench line.
# usage: gpurun --timeout 900 -- 'bash tools/sessions/gpu_r05_s1.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05s1; rm -rf $O; mkdir -p $O
S="conv:4,96,96,512,512 conv:8,96,96,512,512 conv:4,96,96,320,320 conv:4,192,192,512,512"
for rep in 1 2; do
  for V in "tr4:$((2<<20))" "tr3:$((1<<20))" "auto:0"; do
    echo "== ${V%%:*} rep $rep"; GENPERCEPT_IGEMM_DBG=${V##*:} timeout 200 tools/kbench iters=30 cold=1 check=$((rep==1)) $S | grep -vE "^#" | tee -a $O/kbench_${V%%:*}.log
  done
done
echo "== fused GroupNorm+SiLU conv (packed math)"; timeout 200 tools/kbench iters=20 cold=1 check=1 convg:4,768,768,128,128,1 convg:4,768,768,256,128,1 conv:4,768,768,128,128 | grep -vE "^#" | tee $O/kbench_fused.log
G="gemm:36864,320,320,0,1 gemm:9216,640,640,0,1 gemm:2304,1280,1280,0,1 gemm:36864,960,320 gemm:36864,320,1280,0,1"
for A in 0 2 4 8 10 14; do
  echo "== pgemm ablation $A"; GENPERCEPT_IGEMM_DBG=$((512*A)) timeout 200 tools/kbench iters=30 cold=1 check=0 $G | grep -vE "^#" | sed "s/^/abl$A /" | tee -a $O/kbench_pgemm_abl.log
done
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "12row or fused_groupnorm or groupnorm or halo_kernel or conv_epilogue or conv3x3_s1" --timeout=600 -p no:cacheprovider > $O/pytest_kernels.log 2>&1
echo "== kernel tests exit $?"; tail -n 4 $O/pytest_kernels.log
timeout 900 python -m pytest tests/test_refexec_gpu.py tests/test_fullsize_parity_gpu.py tests/test_fp16_range_gpu.py -q --timeout=800 -p no:cacheprovider > $O/pytest_parity.log 2>&1
echo "== parity tests exit $?"; tail -n 6 $O/pytest_parity.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-fp16 > $O/bench.log 2>&1; echo "== bench exit $?"; tail -n 1 $O/bench.log | cut -c1-700
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05s1/bench.log").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "min/med/max", d.get("ms_per_step_min"), d.get("ms_per_step_median"), d.get("ms_per_step_max"))
print("clock_power", d.get("clock_power")); print("stages", d.get("stages")); print("roofline frac", d["roofline"]["frac"], d["roofline"]["sum_ms"])
PY
