cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_xfold_ab.sh
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "fold or layernorm" 2>&1 | tail -3
