#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu --no-fp16 "$@" 2>&1 | tail -1 > gpurun_out/bench_quick.log
python3 - <<'PY'
import json
l=json.loads(open('gpurun_out/bench_quick.log').read())
print('value', l['value'], 'ms', l['ms_per_step'], 'stages', l['stages'])
print('roofline', {k: l['roofline'][k] for k in ('achieved','frac','sum_ms','launches','family_achieved')})
PY
bash tools/gpu_ktrace2.sh
