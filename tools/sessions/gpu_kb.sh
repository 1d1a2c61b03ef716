#!/bin/bash
# generic kbench session: args are passed through; output to gpurun_out/kb_<tag>.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
TAG=$1; shift
timeout 600 tools/kbench "$@" > gpurun_out/kb_$TAG.log 2>&1; echo "exit $?"; cat gpurun_out/kb_$TAG.log
