#!/bin/bash
# scratch session
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attn or attention or flash" 2>&1 | tail -3
timeout 300 python tools/attn_bench.py 2>&1 | tail -12
