#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python tools/conv_bench.py --tiles ${TILES:-4,1} --dbg ${DBG:-0,1,2,3,4,5,6} --shapes ${SHAPES:-vae128,vae512} > gpurun_out/ablate.log 2>&1
cat gpurun_out/ablate.log
