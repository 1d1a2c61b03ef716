#!/bin/bash
# round-2 GPU session E: pre/post, threading, error-path tests; run.py-equivalent timing host vs device pre/post
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -k "prepost or preprocess or postprocess or two_engines or failed_call or pipeline_surface or split" > gpurun_out/pytest_gpu_e.log 2>&1
echo "== pytest exit $?"; tail -n 40 gpurun_out/pytest_gpu_e.log
timeout 900 python tools/prepost_bench.py > gpurun_out/prepost_bench.log 2>&1
echo "== prepost_bench exit $?"; tail -n 3 gpurun_out/prepost_bench.log
