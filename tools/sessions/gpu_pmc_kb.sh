#!/bin/bash
# PMC passes over a kbench invocation: usage gpu_pmc_kb.sh <tag> <kernel-name-substring> <kbench args...>
cd "${GRAFT_REPO_ROOT:-/root/repo}"; TAG=$1; KSUB=$2; shift 2
O=gpurun_out/pmc_$TAG; mkdir -p $O; rm -rf $O/*
export TMPDIR=/tmp; ROOTD=$(pwd)
run_pmc() { (cd /tmp && timeout 120 rocprofv3 --pmc $2 --output-format csv -d "$ROOTD/$O/$1" -- "$ROOTD/tools/kbench" "$@" > "$ROOTD/$O/$1.log" 2>&1); }
ARGS=("$@")
p() { local name=$1; shift; (cd /tmp && timeout 120 rocprofv3 --pmc "$@" --output-format csv -d "$ROOTD/$O/$name" -- "$ROOTD/tools/kbench" "${ARGS[@]}" > "$ROOTD/$O/$name.log" 2>&1); }
p sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
p sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT
p sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT
p tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
p grbm GRBM_GUI_ACTIVE
python3 - "$O" "$KSUB" <<'PY'
import csv, glob, sys, collections
O, ksub = sys.argv[1], sys.argv[2]
agg = collections.OrderedDict()
for f in sorted(glob.glob(O + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if ksub not in r.get('Kernel_Name', ''): continue
        k = (r['Kernel_Name'].split('(')[0][-40:], r.get('Grid_Size', ''), r['Counter_Name'])
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for k, v in agg.items(): print(k[0], 'grid', k[1], k[2], 'n=%d avg=%.6g' % (v[0], v[1] / v[0]))
PY
find $O -name "*.csv" -size +1M -delete
