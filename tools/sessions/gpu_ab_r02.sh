#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo "== $1"; env $1 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages']['ms_encode'], d['stages']['ms_unet'], d['stages']['ms_head'], d['stages']['kernel_launches'])"; }
run "X=1"
run "GENPERCEPT_NO_PGEMM=1"
run "GENPERCEPT_NO_SPLITK=1"
run "GENPERCEPT_NO_CROSS_FOLD=1"
run "GENPERCEPT_NO_GN_SMALL=1"
run "X=2"
