#!/bin/bash
# round-2 profiling session: (1) rocprofv3 kernel stats of the bench command, (2) PMC counter sets for the FUSED GroupNorm+SiLU halo conv and
# its plain twin on 128->128@768^2 B=4, (3) HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the bench pass.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r02prof
rm -rf $O; mkdir -p $O
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python "$ROOTD/bench.py" --steps 2 --warmup 1 --no-cpu --no-profile > "$O/stats_bench.log" 2>&1)
echo "== stats exit $?"; tail -n 1 $O/stats_bench.log | cut -c1-200
F=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $O/kernel_stats.csv && head -12 $O/kernel_stats.csv | cut -c1-200
find $O/stats -name "*.csv" -size +5M -delete
pmc_set() { # tag, conv_bench args
  python tools/conv_bench.py --iters 10 $2 > $O/pmc_$1_timing.log 2>&1; cat $O/pmc_$1_timing.log
  i=0
  for CS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16" \
            "SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --pmc $CS --output-format csv -d "$O/pmc_$1_$i" -- python "$ROOTD/tools/conv_bench.py" --iters 2 --rounds 2 $2 > "$O/pmc_$1_$i.log" 2>&1)
    echo "pmc $1 set $i exit $?"
  done
}
pmc_set fused "--gn 1 --shapes vae128"
pmc_set plain "--tiles 5 --shapes vae128"
python - <<'PY'
import csv, glob, json, collections, os
O = "gpurun_out/r02prof"
out = {}
for tag in ("fused", "plain"):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob(f"{O}/pmc_{tag}_*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if "halo3" not in r.get("Kernel_Name", ""):
                continue
            a = agg.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    out[tag] = {k: v[1] / v[0] for k, v in agg.items()}
    out[tag]["dispatches_averaged"] = max((v[0] for v in agg.values()), default=0)
    out[tag]["timing"] = open(f"{O}/pmc_{tag}_timing.log").read().strip().splitlines()[-1]
json.dump(out, open(f"{O}/pmc_halo3_fused_vs_plain.json", "w"), indent=1)
for tag, d in out.items():
    print(tag, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items()})
PY
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d "$O/traffic_$C" -- python "$ROOTD/bench.py" --steps 1 --warmup 1 --no-cpu --no-profile > "$O/traffic_$C.log" 2>&1)
  echo "traffic $C exit $?"
done
python - <<'PY'
import csv, glob, json, collections
O = "gpurun_out/r02prof"
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 = 2 passes of B=4 768x768; sum_kb as reported (FETCH_SIZE to be doubled on gfx950)"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{O}/traffic_{c}/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"])
    out[c] = {k: {"dispatches": v[0], "sum_kb": v[1]} for k, v in agg.items()}
json.dump(out, open(f"{O}/pmc_traffic_summary.json", "w"), indent=1)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = out.get(c, {})
    print(c, "total GB (raw):", round(sum(v["sum_kb"] for v in d.values()) * 1024 / 1e9, 2))
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]["sum_kb"])[:6]:
        print("   ", k[:60], v["dispatches"], round(v["sum_kb"] * 1024 / 1e9, 3), "GB")
PY
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete; du -sh $O
