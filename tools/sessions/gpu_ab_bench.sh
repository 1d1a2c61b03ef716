#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() { echo "== $1"; env $2 python bench.py --steps 4 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['stages'])"; }
run default "X=1"
run "$1" "$1=1"
run default "X=1"
