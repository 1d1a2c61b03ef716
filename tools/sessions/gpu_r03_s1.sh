#!/bin/bash
# round 3, GPU session 1: torch-free baseline of the UNet's GEMM / small-map conv / attention kernels (tools/kbench), hot and weight-cold,
# every tile configuration, + PMC passes on three representative persistent-GEMM shapes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r03s1; rm -rf gpurun_out/r03s1/*
export TMPDIR=/tmp; ROOTD=$(pwd); O=gpurun_out/r03s1
H="0,1,2,4,6"
timeout 300 tools/kbench iters=20 \
  gemm:36864,320,320,0,0,0,$H gemm:36864,320,320,0,1,0,0 gemm:36864,640,320,0,0,0,$H gemm:36864,2560,320,3,0,0,0,4 gemm:36864,320,1280,0,1,0,$H \
  gemm:36864,320,960,0,0,0,0 \
  gemm:9216,640,640,0,0,0,$H gemm:9216,1280,640,0,0,0,$H gemm:9216,5120,640,3,0,0,0,4 gemm:9216,640,2560,0,1,0,$H \
  gemm:2304,1280,1280,0,0,0,$H gemm:2304,2560,1280,0,0,0,$H gemm:2304,10240,1280,3,0,0,0,4 gemm:2304,1280,5120,0,1,0,$H \
  gemm:576,1280,1280,0,0,0,$H gemm:576,10240,1280,3,0,0,0,4 gemm:576,1280,5120,0,1,0,$H \
  > $O/gemm.log 2>&1; echo "gemm exit $?"; cat $O/gemm.log
timeout 300 tools/kbench iters=10 \
  conv:4,24,24,1280,1280,0,0,1,2,4 conv:4,24,24,2560,1280,0,0,1,4 conv:4,24,24,1920,1280,0,0 conv:4,12,12,1280,1280,0,0,1,2 conv:4,12,12,2560,1280,0,0,2 \
  conv:4,48,48,640,640,0,0,1,4 conv:4,48,48,1280,640,0,0 conv:4,96,96,320,320,0,0,4 conv:4,96,96,640,320,0,0 conv:4,24,24,1280,1280,1,0 \
  attn:4,9216,5 attn:4,2304,10 attn:4,576,20 attn:4,144,20 \
  > $O/conv.log 2>&1; echo "conv exit $?"; cat $O/conv.log
PM="iters=3 cold=1 check=0 gemm:36864,320,320 gemm:2304,1280,1280 gemm:36864,2560,320,3 gemm:36864,320,1280,0,1"
run_pmc() {
  (cd /tmp && timeout 120 rocprofv3 --pmc $2 --output-format csv -d "$ROOTD/$O/$1" -- "$ROOTD/tools/kbench" $PM > "$ROOTD/$O/$1.log" 2>&1)
}
run_pmc sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
run_pmc tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
run_pmc grbm "GRBM_GUI_ACTIVE"
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r03s1/**/*counter_collection.csv", recursive=True)):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if 'pgemm' not in r.get('Kernel_Name', ''): continue
        k = (r['Kernel_Name'][:28], r.get('Grid_Size', ''), r['Counter_Name'])
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
    for k, v in agg.items(): print(k[0], 'grid', k[1], k[2], 'n=%d avg=%.5g' % (v[0], v[1] / v[0]))
PY
find gpurun_out/r03s1 -name "*.csv" -size +2M -delete
