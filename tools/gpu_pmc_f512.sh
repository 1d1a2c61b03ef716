#!/bin/bash
# PMC passes over flash_attn512_kernel (separate runs, --pmc only): tools/attn_bench.py --hd512-only
ROOTD=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc512
run_pmc() { # name, counters
  (cd /tmp && timeout 200 rocprofv3 --pmc $2 --output-format csv -d "$ROOTD/gpurun_out/pmc512/$1" -- python "$ROOTD/tools/attn_bench.py" --hd512-only > "$ROOTD/gpurun_out/pmc512/$1.log" 2>&1)
  echo "pmc $1 exit $?"
}
run_pmc sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU"
run_pmc sq2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
run_pmc grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
for f in $(find gpurun_out/pmc512 -name "*counter_collection.csv"); do echo "== $f"; python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r.get('Kernel_Name','')[:28], r.get('Grid_Size', r.get('Grid_Size_X','')), r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for k, v in agg.items():
    if 'flash_attn512' in k[0]:
        print(k, 'n=%d avg=%.5g' % (v[0], v[1] / v[0]))
PY
done
find gpurun_out/pmc512 -name "*.csv" -size +2M -delete
