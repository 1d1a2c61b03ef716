#!/bin/bash
# usage: gpu_pytest.sh <pytest args...>   (log -> gpurun_out/pytest_sel.log)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1500 python -m pytest "$@" 2>&1 | tail -25 | tee gpurun_out/pytest_sel.log
