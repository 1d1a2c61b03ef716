#!/bin/bash
# Round-6 counter session: SQ counters of the SHIPPED build on the bench command, per kernel, stamped with the build id bench.py checks:
#   pass 1  GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT   -> mfma_busy_fraction, VALU per MFMA-busy cycle
#   pass 2 / 3  FETCH_SIZE, WRITE_SIZE (separate passes; FETCH_SIZE doubled on gfx950 by the reader)           -> roofline.traffic
# --pmc passes carry no trace domains (gpurun refuses the combination).  usage: gpurun --timeout 1500 -- 'bash tools/sessions/gpu_r06_pmc.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r06pmc
rm -rf $O; mkdir -p $O
BID=$(python3 -c "import bench; print(bench.source_build_id())")
ARGS="--steps 1 --warmup 1 --no-cpu --no-profile --no-fp16 --no-fp32c"
(cd /tmp && timeout 500 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$O/sq" -- python "$ROOTD/bench.py" $ARGS > "$O/sq.log" 2>&1)
echo "sq exit $?"; tail -n 2 $O/sq.log | cut -c1-200
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --output-format csv -d "$O/traffic_$C" -- python "$ROOTD/bench.py" $ARGS > "$O/traffic_$C.log" 2>&1)
  echo "traffic $C exit $?"
done
python3 - "$BID" <<'PY'
import csv, glob, json, collections, re, sys
O = "gpurun_out/r06pmc"
bid = sys.argv[1]
def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0] if "<" not in n else n[: n.index(">") + 1] if n.index("<") < (n.index("(") if "(" in n else 10**9) else n.split("(")[0]
# ---- SQ pass
f = glob.glob(f"{O}/sq/**/*counter_collection.csv", recursive=True)
sq = {"note": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT over `bench.py --steps 1 --warmup 1` (2 passes of "
      "B = 4, 768 x 768, bf16); per kernel: dispatches and counter sums over them.  mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs) "
      "(the r3 formula, profiles/r03_pmc_halo3_clock_flash64.json); valu_per_mfma_busy_cycle = SQ_INSTS_VALU (wave instructions) / SQ_VALU_MFMA_BUSY_CYCLES.",
      "build_id": bid, "kernels": {}}
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r.get("Dispatch_Id", r.get("Dispatch_ID", "")))
    for k, v in agg.items():
        d = {"dispatches": len(cnt[k])}
        d.update({c: v[c] for c in sorted(v)})
        gui, mf = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if gui > 0:
            d["mfma_busy_fraction"] = round(mf / (gui / 8.0 * 1024.0), 4)
        if mf > 0:
            d["valu_per_mfma_busy_cycle"] = round(v.get("SQ_INSTS_VALU", 0.0) / mf, 4)
        sq["kernels"][k] = d
json.dump(sq, open(f"{O}/pmc_mfma_summary.json", "w"), indent=1)
top = sorted(sq["kernels"].items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:14]
for k, d in top:
    print(f"{d.get('mfma_busy_fraction', 0):6.3f}  n={d['dispatches']:4d}  {k[:110]}")
# ---- traffic passes
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 = 2 passes of B=4 768x768; sum_kb as reported (FETCH_SIZE to be doubled on gfx950)", "build_id": bid}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{O}/traffic_{c}/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"])
    out[c] = {k: {"dispatches": v[0], "sum_kb": v[1]} for k, v in agg.items()}
json.dump(out, open(f"{O}/pmc_traffic_summary.json", "w"), indent=1)
PY
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete; du -sh $O
