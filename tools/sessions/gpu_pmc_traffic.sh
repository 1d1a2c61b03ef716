#!/bin/bash
# HBM traffic of the dominant kernels from PMC counters (separate --pmc passes, no tracing), plus configs 2/3 bench lines.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/traffic; rm -rf gpurun_out/traffic/*
export TMPDIR=/tmp; ROOTD=$(pwd)
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d "$ROOTD/gpurun_out/traffic/$C" -- python "$ROOTD/bench.py" --steps 1 --warmup 1 --no-cpu --no-profile > "$ROOTD/gpurun_out/traffic/$C.log" 2>&1)
  echo "pmc $C exit $?"
done
python - <<'PY'
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/traffic/{c}/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"])
    out[c] = {k: {"dispatches": v[0], "sum_kb": v[1]} for k, v in agg.items()}
json.dump(out, open("gpurun_out/traffic/summary.json", "w"), indent=1)
for c, d in out.items():
    tot = sum(v["sum_kb"] for v in d.values())
    print(c, "total GB over 2 passes (raw KB*1024):", tot * 1024 / 1e9)
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]["sum_kb"])[:8]:
        print("   ", k[:60], v["dispatches"], round(v["sum_kb"] * 1024 / 1e9, 3), "GB")
PY
find gpurun_out/traffic -name "*.csv" -size +3M -delete
python bench.py --steps 3 --warmup 1 --no-cpu --mode normal 2>&1 | tail -1 > gpurun_out/bench_normal.log; cut -c1-330 gpurun_out/bench_normal.log
python bench.py --steps 3 --warmup 1 --no-cpu --head dpt 2>&1 | tail -1 > gpurun_out/bench_dpt.log; cut -c1-330 gpurun_out/bench_dpt.log
