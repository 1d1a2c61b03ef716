#!/bin/bash
# PMC passes (one rocprofv3 --pmc run per counter set) over a conv_bench invocation.  usage: gpu_pmc.sh "<conv_bench args>"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/pmc2; rm -rf gpurun_out/pmc2/*
export TMPDIR=/tmp
ROOTD=$(pwd)
ARGS="$1"
python tools/conv_bench.py --iters 10 $ARGS > gpurun_out/pmc2/timing.log 2>&1; cat gpurun_out/pmc2/timing.log
run_pmc() {
  (cd /tmp && timeout 300 rocprofv3 --pmc $2 --output-format csv -d "$ROOTD/gpurun_out/pmc2/$1" -- python "$ROOTD/tools/conv_bench.py" --iters 2 $ARGS > "$ROOTD/gpurun_out/pmc2/$1.log" 2>&1)
}
run_pmc sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
run_pmc sq2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT"
run_pmc sq3 "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_WAVES SQ_INSTS_SMEM"
run_pmc tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
run_pmc grbm "GRBM_GUI_ACTIVE"
for f in $(find gpurun_out/pmc2 -name "*counter_collection.csv"); do python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r.get('Kernel_Name','')[:44], r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for k, v in agg.items():
    if 'halo' in k[0] or 'igemm' in k[0]:
        print(k[0], k[1], 'n=%d avg=%.4g' % (v[0], v[1] / v[0]))
PY
done
find gpurun_out/pmc2 -name "*.csv" -size +2M -delete
