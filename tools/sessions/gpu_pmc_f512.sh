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
python - <<'PY'
import csv, glob, json, collections
out = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmc512/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "flash_attn512" not in r.get("Kernel_Name", ""):
            continue
        k = "grid_%s" % r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        a = out.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
res = {"note": "rocprofv3 --pmc (separate passes) over tools/attn_bench.py --hd512-only; per-launch averages of flash_attn512_kernel; grid = threads "
               "(65536 = batch 4 x 9216 tokens on 256 workgroups)", "kernels": {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in out.items()}}
json.dump(res, open("gpurun_out/pmc512/summary.json", "w"), indent=1)
PY
find gpurun_out/pmc512 -name "*.csv" -size +2M -delete
