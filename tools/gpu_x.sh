#!/bin/bash
# scratch session
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for r in 1 2; do
echo "== BASE $r"; LD_LIBRARY_PATH=$PWD/tools/probe/base_lib timeout 100 tools/kbench iters=40 cold=0 check=0 attn:4,9216,5 attn:4,2304,10 attn:4,576,20 | grep -vE "^#"
echo "== NEW $r"; timeout 100 tools/kbench iters=40 cold=0 check=0 attn:4,9216,5 attn:4,2304,10 attn:4,576,20 | grep -vE "^#"
done
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attn or attention or flash" 2>&1 | tail -3
