#!/bin/bash
# Round-5 collection session: smoke, every GPU test, the bench line (driver's command), per-dispatch kernel trace + stats of the bench command,
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, stamped with the build id bench.py checks), the other benched configurations.
# usage: gpurun --timeout 2400 -- 'bash tools/sessions/gpu_r05_final.sh [skip-tests]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r05final
rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "== smoke exit $?"; tail -n 2 $O/smoke.log
if [ "$1" != "skip-tests" ]; then
  rm -f gpurun_out/parity_log.jsonl
  timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
  echo "== pytest exit $?"; tail -n 6 $O/pytest_gpu.log
fi
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "== bench exit $?"; tail -n 1 $O/bench.log | cut -c1-400
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python "$ROOTD/bench.py" --steps 2 --warmup 1 --no-cpu --no-profile --no-fp16 > "$O/stats_bench.log" 2>&1)
echo "== stats exit $?"
F=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $O/kernel_stats.csv && head -6 $O/kernel_stats.csv | cut -c1-160
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r05final/stats/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rgb_conv_in" in r["Kernel_Name"]]
start = idx[-1]
out = open("gpurun_out/r05final/kernel_trace_last_pass.tsv", "w")
out.write("# one bench pass (batch 4, 768x768, bf16), rocprofv3 --kernel-trace: kernel duration us, gap to the previous kernel's end us, grid, workgroup, kernel\n")
prev_end = None
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
    if name.startswith("at::") or "rocclr" in name: continue
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    out.write(f"{(e - s) / 1e3:.2f}\t{gap:.2f}\t{r.get('Grid_Size_X', r.get('Grid_Size', ''))}\t{r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))}\t{name}\n")
    prev_end = e
out.close()
PY
wc -l $O/kernel_trace_last_pass.tsv
BID=$(python3 -c "import bench; print(bench.source_build_id())")
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --output-format csv -d "$O/traffic_$C" -- python "$ROOTD/bench.py" --steps 1 --warmup 1 --no-cpu --no-profile --no-fp16 > "$O/traffic_$C.log" 2>&1)
  echo "traffic $C exit $?"
done
python3 - "$BID" <<'PY'
import csv, glob, json, collections, sys
O = "gpurun_out/r05final"
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 = 2 passes of B=4 768x768; sum_kb as reported (FETCH_SIZE to be doubled on gfx950)", "build_id": sys.argv[1]}
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
PY
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-fp16 --mode normal 2>&1 | tail -1 > $O/bench_normal.log; cut -c1-160 $O/bench_normal.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-fp16 --head dpt 2>&1 | tail -1 > $O/bench_dpt.log; cut -c1-160 $O/bench_dpt.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-fp16 --batch 8 2>&1 | tail -1 > $O/bench_b8.log; cut -c1-160 $O/bench_b8.log
timeout 300 python tools/launch_log.py --tag r05 > $O/launch_log_run.log 2>&1; echo "== launch_log exit $?"; tail -n 3 $O/launch_log_run.log
# same-box A/B against the library of the start of the round's kernel work where present (genpercept_amd/lib/base), alternating
if [ -f genpercept_amd/lib/base/libgenpercept_hip.so ]; then
  for E in "new:" "base:GENPERCEPT_HIP_LIB=$ROOTD/genpercept_amd/lib/base/libgenpercept_hip.so" "new2:" "base2:GENPERCEPT_HIP_LIB=$ROOTD/genpercept_amd/lib/base/libgenpercept_hip.so"; do
    env ${E#*:} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-fp16 2>&1 | tail -1 > $O/bench_ab_${E%%:*}.log
    python3 -c "import json; d=json.load(open('$O/bench_ab_${E%%:*}.log')); print('ab ${E%%:*}', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['stages']['ms_encode'], d['stages']['ms_unet'], d['stages']['ms_head'], d['clock_power']['sclk_mhz_mean'])"
  done
fi
timeout 300 python tools/mfma_lds_probe.py > $O/mfma_lds_probe.json 2> $O/probe.err; echo "== probe exit $?"
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete; du -sh $O
