#!/bin/bash
# Round-4 GPU session 10: PMC of conv3x3_halo3_kernel vs conv3x3_halo5_kernel (two workgroups per CU) on one shape: where do the cycles go?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s10; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; ROOTD=$(pwd)
SHAPE=${1:-conv:4,384,384,256,256}
pmc() {  # tag dbg
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT" \
             "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
    N=$(echo $SET | cut -c1-12 | tr ' ' '_')
    (cd /tmp && GENPERCEPT_IGEMM_DBG=$2 timeout 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$ROOTD/$O/pmc_$1_$N" -- "$ROOTD/tools/kbench" iters=5 cold=1 check=0 $SHAPE > "$ROOTD/$O/pmc_$1_$N.log" 2>&1) || tail -3 "$ROOTD/$O/pmc_$1_$N.log"
  done
}
pmc halo3 $((2<<28)); pmc halo5 $((1<<28))
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
out = {}
for tag in ("halo3", "halo5"):
    cnt = collections.OrderedDict(); dur = [0, 0.0]
    for d in sorted(glob.glob(f"{O}/pmc_{tag}_*/")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if 'conv3x3_halo' not in r.get('Kernel_Name', ''): continue
                a = cnt.setdefault(r['Counter_Name'], [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if 'conv3x3_halo' not in r.get('Kernel_Name', ''): continue
                dur[0] += 1; dur[1] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    us = dur[1] / max(dur[0], 1) / 1e3
    out[tag] = {"dur_us": us, **{k: v[1] / v[0] for k, v in cnt.items()}}
    print(tag, 'dur %.1f us' % us, ' '.join('%s=%.5g' % (k, v[1] / v[0]) for k, v in cnt.items()))
json.dump(out, open(f"{O}/pmc_summary.json", "w"), indent=1)
PY
find $O -name "*.csv" -size +1M -delete
