#!/bin/bash
# pgemm ablations (GENPERCEPT_IGEMM_DBG = 512 * ABL; 2: no MFMA, 4: no output stores, 8: no DMA) + long-K main-loop rate
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
S="gemm:36864,2560,320,3 gemm:36864,2560,320 gemm:36864,640,320 gemm:36864,320,1280,0,1 gemm:2304,1280,1280 gemm:2304,10240,1280,3"
for A in 0 2 4 8; do echo "== ABL $A"; GENPERCEPT_IGEMM_DBG=$((512*A)) timeout 120 tools/kbench iters=20 cold=1 check=0 $S | grep gemm; done
echo "== long K"; timeout 120 tools/kbench iters=10 cold=1 check=0 gemm:36864,2560,5120 gemm:36864,1280,5120,0,0,0,0,4 gemm:4608,1280,10240 gemm:2304,1280,10240 | grep gemm
