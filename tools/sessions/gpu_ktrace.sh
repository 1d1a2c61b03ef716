#!/bin/bash
# Kernel-only durations (rocprofv3 --kernel-trace) of a conv_bench invocation: host launch overhead (~10 us per Python call) hides
# kernels shorter than that from event timing.  usage: gpu_ktrace.sh "<conv_bench args>"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/ktrace; rm -rf gpurun_out/ktrace/*
export TMPDIR=/tmp
ROOTD=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/ktrace" -- python "$ROOTD/tools/conv_bench.py" --iters 5 --rounds 4 $1 > "$ROOTD/gpurun_out/ktrace/run.log" 2>&1)
F=$(find gpurun_out/ktrace -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'igemm' not in n and 'pgemm' not in n and 'halo' not in n: continue
    k = (n[:46], r['Grid_Size_X'], r['Grid_Size_Y'])
    agg.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    print('%-48s grid %8s x %s  n=%3d  median %8.1f us  min %8.1f us' % (k[0], k[1], k[2], len(v), statistics.median(v), min(v)))
PY
find gpurun_out/ktrace -name "*.csv" -size +5M -delete
