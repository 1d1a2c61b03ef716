#!/bin/bash
# A/B of environment settings within ONE box: bench lines for "default" and each "VAR=VALUE" argument, default repeated at the end.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() { echo "== $1"; env $1 python bench.py --steps 4 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages'])"; }
run "X=1"
for a in "$@"; do run "$a"; done
run "X=1"
