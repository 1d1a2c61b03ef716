#!/bin/bash
# effective shader clock of the halo3 conv with and without its LDS-DMA traffic: GRBM_GUI_ACTIVE / kernel duration (rocprofv3 --pmc + --kernel-trace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/halo_clock; mkdir -p $O; rm -rf $O/*
export TMPDIR=/tmp; ROOTD=$(pwd)
SPEC=${SPEC:-conv:4,192,192,512,512}
for A in ${ABLS:-0 24 2}; do
  (cd /tmp && GENPERCEPT_IGEMM_DBG=$((512*A)) timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$ROOTD/$O/a$A" -- "$ROOTD/tools/kbench" iters=10 cold=1 check=0 $SPEC > "$ROOTD/$O/a$A.log" 2>&1)
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for d in sorted(glob.glob(O + "/a*/")):
    cnt = collections.defaultdict(lambda: [0, 0.0]); dur = [0, 0.0]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if 'halo3' not in r.get('Kernel_Name', ''): continue
            a = cnt[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if 'halo3' not in r.get('Kernel_Name', ''): continue
            dur[0] += 1; dur[1] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    if not dur[0]: print(d, 'no trace'); continue
    us = dur[1] / dur[0] / 1e3
    line = '%s n=%d dur=%.1f us' % (d.rstrip('/').split('/')[-1], dur[0], us)
    for k, v in cnt.items(): line += '  %s=%.4g' % (k, v[1] / v[0])
    if 'GRBM_GUI_ACTIVE' in cnt: line += '  clock=%.3f GHz' % (cnt['GRBM_GUI_ACTIVE'][1] / cnt['GRBM_GUI_ACTIVE'][0] / us / 1e3)
    print(line)
PY
find $O -name "*.csv" -size +1M -delete
