cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo NEW; timeout 100 tools/kbench iters=10 check=0 conv:8,24,24,1280,1280 conv:8,12,12,1280,1280 conv:8,24,24,2560,1280 gemm:576,1280,1280 gemm:576,1280,5120,0,1 | grep -E "conv|gemm"
echo OLD; GENPERCEPT_NO_CONV_IMG=1 timeout 100 tools/kbench iters=10 check=0 conv:8,24,24,1280,1280 conv:8,12,12,1280,1280 conv:8,24,24,2560,1280 | grep conv
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
