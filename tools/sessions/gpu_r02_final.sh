#!/bin/bash
# Final round-2 GPU session: smoke, every GPU test, bench line, per-launch log, rocprofv3 kernel stats of the bench command, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate --pmc passes), the other two benched configurations, the multi-step archs.
# usage: gpurun --timeout 3000 -- 'bash tools/gpu_r02_final.sh <commit>'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r02final
rm -rf $O; mkdir -p $O; rm -f gpurun_out/parity_log.jsonl
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "== smoke exit $?"; tail -n 2 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "== pytest exit $?"; tail -n 6 $O/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1; echo "== bench exit $?"; tail -n 1 $O/bench.log
timeout 300 python tools/launch_log.py --tag r02 > $O/launch_log_run.log 2>&1; echo "== launch_log exit $?"; tail -n 22 $O/launch_log_run.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python "$ROOTD/bench.py" --steps 2 --warmup 1 --no-cpu --no-profile > "$O/stats_bench.log" 2>&1)
echo "== stats exit $?"
F=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $O/kernel_stats.csv && head -8 $O/kernel_stats.csv | cut -c1-180
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --output-format csv -d "$O/traffic_$C" -- python "$ROOTD/bench.py" --steps 1 --warmup 1 --no-cpu --no-profile > "$O/traffic_$C.log" 2>&1)
  echo "traffic $C exit $?"
done
python - "$1" <<'PY'
import csv, glob, json, collections, sys
O = "gpurun_out/r02final"
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 = 2 passes of B=4 768x768; sum_kb as reported (FETCH_SIZE to be doubled on gfx950)", "commit": sys.argv[1] if len(sys.argv) > 1 else "?"}
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
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]["sum_kb"])[:5]:
        print("   ", k[:60], v["dispatches"], round(v["sum_kb"] * 1024 / 1e9, 3), "GB")
PY
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --mode normal 2>&1 | tail -1 > $O/bench_normal.log; cut -c1-200 $O/bench_normal.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --head dpt 2>&1 | tail -1 > $O/bench_dpt.log; cut -c1-200 $O/bench_dpt.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --precision fp16 2>&1 | tail -1 > $O/bench_fp16.log; cut -c1-200 $O/bench_fp16.log
timeout 300 python tools/multistep_bench.py --archs marigold > $O/multistep_marigold.log 2>&1; tail -n 1 $O/multistep_marigold.log
timeout 300 python tools/multistep_bench.py --archs rgb_blending --denoise-steps 4 > $O/multistep_blend.log 2>&1; tail -n 1 $O/multistep_blend.log
timeout 200 python tools/attn_bench.py > $O/attn_bench.log 2>&1; cat $O/attn_bench.log | grep "T="
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete; du -sh $O
