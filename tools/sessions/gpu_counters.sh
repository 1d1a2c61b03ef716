#!/bin/bash
# Counter study of the conv kernel: timing per tile config, then PMC passes (separate runs, --pmc only).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
ROOTD=$(pwd)
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
python tools/conv_bench.py --iters 10 --tiles 1,4 --shapes vae128,vae256,vae512,vae512_96,unet320,unet640 > gpurun_out/pmc/conv_bench.log 2>&1
python tools/conv_bench.py --iters 10 --tiles 2,1 --shapes unet1280_24,unet1280_12,unet2560_12,lin320,ff320,ff1280,lin1280_24 >> gpurun_out/pmc/conv_bench.log 2>&1
cat gpurun_out/pmc/conv_bench.log
run_pmc() { # name, counters
  (cd /tmp && timeout 300 rocprofv3 --pmc $2 --output-format csv -d "$ROOTD/gpurun_out/pmc/$1" -- python "$ROOTD/tools/conv_bench.py" --iters 2 --tiles 4 --shapes vae128,vae512 > "$ROOTD/gpurun_out/pmc/$1.log" 2>&1)
  echo "pmc $1 exit $?"
}
run_pmc sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
run_pmc sq2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VALU"
run_pmc tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
run_pmc tcp1 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
run_pmc grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
for f in $(find gpurun_out/pmc -name "*counter_collection.csv"); do echo "== $f"; python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r.get('Kernel_Name','')[:40], r.get('Grid_Size', r.get('Grid_Size_X','')), r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for k, v in agg.items():
    if 'igemm' in k[0]:
        print(k, 'n=%d avg=%.4g' % (v[0], v[1] / v[0]))
PY
done
find gpurun_out/pmc -name "*.csv" -size +5M -delete
