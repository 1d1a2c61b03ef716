#!/bin/bash
# Round-4 GPU session 2: the new 512-pixel-tile conv kernel (tests, then kbench A/B against halo3 on the pipeline's plain shapes), the fp16
# range stress test, every GPU test, the bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s2; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_log.jsonl
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=300 -p no:cacheprovider -k "halo4" > $O/pytest_halo4.log 2>&1
echo "== halo4 tests exit $?"; tail -n 12 $O/pytest_halo4.log
S="conv:4,768,768,256,128 conv:4,384,384,128,256 conv:4,384,384,256,256 conv:4,384,384,512,256 conv:4,192,192,256,512 conv:4,192,192,512,512 conv:4,96,96,512,512 conv:8,192,192,512,512"
for rep in 1 2; do
  echo "== halo3 (forbid halo4), rep $rep"; GENPERCEPT_IGEMM_DBG=$((2<<20)) timeout 200 tools/kbench iters=20 cold=1 check=$((2-rep)) $S | grep -vE "^#" | tee -a $O/kbench_halo3.log
  echo "== halo4 (forced), rep $rep";      GENPERCEPT_IGEMM_DBG=$((1<<20)) timeout 200 tools/kbench iters=20 cold=1 check=$((2-rep)) $S | grep -vE "^#" | tee -a $O/kbench_halo4.log
done
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "== pytest exit $?"; tail -n 12 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "== bench exit $?"; tail -n 1 $O/bench.log | cut -c1-700
