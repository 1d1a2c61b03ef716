#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for px in 16384 5000 1000 0; do
  echo "== GN fuse below px $px"
  GENPERCEPT_GN_FUSE_BELOW_PX=$px timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages'])"
done
echo "== max slices 4, px 0"
GENPERCEPT_GN_FUSE_MAX_SLICES=0 GENPERCEPT_GN_FUSE_BELOW_PX=0 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages'])"
