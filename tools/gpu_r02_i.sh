#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -k "infer_eval_loop" > gpurun_out/pytest_gpu_i.log 2>&1; echo "== pytest exit $?"; tail -n 15 gpurun_out/pytest_gpu_i.log
bash tools/gpu_r02_prof.sh
