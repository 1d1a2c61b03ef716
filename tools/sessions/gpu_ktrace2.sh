#!/bin/bash
# per-dispatch kernel trace (rocprofv3 --kernel-trace, no in-engine events) of two bench passes: true in-pipeline kernel durations and gaps
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/kt; rm -rf gpurun_out/kt/*
export TMPDIR=/tmp; ROOTD=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/kt/trace" -- python "$ROOTD/bench.py" --steps 1 --warmup 1 --no-cpu --no-profile --no-fp16 ${BENCH_ARGS} > "$ROOTD/gpurun_out/kt/bench.log" 2>&1)
echo "exit $?"; tail -1 gpurun_out/kt/bench.log | cut -c1-200
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt/trace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last pass: find the last rgb_conv_in
idx = [i for i, r in enumerate(rows) if "rgb_conv_in" in r["Kernel_Name"]]
start = idx[-1]
out = open("gpurun_out/kt/last_pass.tsv", "w")
prev_end = None
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) if prev_end else 0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    out.write(f"{(e - s) / 1e3:.2f}\t{gap / 1e3:.2f}\t{r.get('Grid_Size_X', r.get('Grid_Size', ''))}\t{r.get('Workgroup_Size_X', '')}\t{name}\n")
    prev_end = e
out.close()
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[start:])
print("last pass: kernels", len(rows) - start, "sum kernel ms", tot / 1e6, "wall ms", (int(rows[-1]["End_Timestamp"]) - int(rows[start]["Start_Timestamp"])) / 1e6)
PY
find gpurun_out/kt -name "*.csv" -size +3M -delete
