#!/bin/bash
# SQ counters of the CONTRACT precision (bench.py --precision fp32c): matrix-pipe busy of the F32O halo conv, flash_attn64_split_kernel, the generic GEMM, the c_* kernels.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r06pmc_fp32c
rm -rf $O; mkdir -p $O
BID=$(python3 -c "import bench; print(bench.source_build_id())")
(cd /tmp && timeout 500 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$O/sq" -- python "$ROOTD/bench.py" --steps 1 --warmup 1 --no-cpu --no-profile --no-fp16 --no-fp32c --precision fp32c > "$O/sq.log" 2>&1)
echo "sq exit $?"
python3 - "$BID" <<'PY'
import csv, glob, json, collections, sys
O = "gpurun_out/r06pmc_fp32c"
def short(name):
    n = name.replace("void ", "")
    return n[: n.index(">") + 1] if "<" in n and n.index("<") < (n.index("(") if "(" in n else 10**9) else n.split("(")[0]
f = glob.glob(f"{O}/sq/**/*counter_collection.csv", recursive=True)
out = {"note": "contract precision (bench.py --precision fp32c --steps 1 --warmup 1: 2 passes of B = 4, 768 x 768): rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES "
               "SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT; mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)", "build_id": sys.argv[1], "kernels": {}}
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r.get("Dispatch_Id", ""))
for k, v in agg.items():
    d = {"dispatches": len(cnt[k])}; d.update({c: v[c] for c in sorted(v)})
    gui, mf = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui > 0: d["mfma_busy_fraction"] = round(mf / (gui / 8.0 * 1024.0), 4)
    out["kernels"][k] = d
json.dump(out, open(f"{O}/pmc_mfma_fp32c.json", "w"), indent=1)
for k, d in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:12]:
    print(f"{d.get('mfma_busy_fraction', 0):6.3f}  n={d['dispatches']:4d}  {k[:110]}")
PY
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
