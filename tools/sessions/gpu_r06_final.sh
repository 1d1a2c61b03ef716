#!/bin/bash
# Round-6 collection session: smoke, every GPU test, the bench line (driver's command: bf16 + fp16 + fp32c legs), per-dispatch kernel trace + stats of the
# bench command, SQ counters (matrix-pipe busy per kernel) and HBM traffic in separate --pmc passes (stamped with the build id bench.py checks), the bench
# line AGAIN with those summaries in place (roofline.traffic / roofline.mfma_busy_fraction filled), the other benched configurations, launch logs (bf16 and
# contract precision), kernel stats of the contract precision.
# usage: gpurun --timeout 3000 -- 'bash tools/sessions/gpu_r06_final.sh [skip-tests]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r06final
rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "== smoke exit $?"; tail -n 3 $O/smoke.log
if [ "$1" != "skip-tests" ]; then
  rm -f gpurun_out/parity_log.jsonl
  timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
  echo "== pytest exit $?"; tail -n 6 $O/pytest_gpu.log | cut -c1-200
  cp gpurun_out/parity_log.jsonl $O/parity_log.jsonl
fi
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "== bench exit $?"; tail -n 1 $O/bench.log | cut -c1-300
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python "$ROOTD/bench.py" --steps 2 --warmup 1 --no-cpu --no-profile --no-fp16 --no-fp32c > "$O/stats_bench.log" 2>&1)
echo "== stats exit $?"
F=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $O/kernel_stats.csv && head -6 $O/kernel_stats.csv | cut -c1-160
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r06final/stats/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rgb_conv_in" in r["Kernel_Name"]]
start = idx[-1]
out = open("gpurun_out/r06final/kernel_trace_last_pass.tsv", "w")
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
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_fp32c" -- python "$ROOTD/bench.py" --steps 2 --warmup 1 --no-cpu --no-profile --no-fp16 --no-fp32c --precision fp32c > "$O/stats_fp32c_bench.log" 2>&1)
echo "== stats fp32c exit $?"
F=$(find $O/stats_fp32c -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $O/kernel_stats_fp32c.csv && head -8 $O/kernel_stats_fp32c.csv | cut -c1-160
bash tools/sessions/gpu_r06_pmc.sh > $O/pmc_session.log 2>&1; echo "== pmc exit $?"; tail -n 16 $O/pmc_session.log | cut -c1-150
cp gpurun_out/r06pmc/pmc_mfma_summary.json profiles/r06_pmc_mfma_summary.json; cp gpurun_out/r06pmc/pmc_traffic_summary.json profiles/r06_pmc_traffic_summary.json
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_with_pmc.log 2>&1; echo "== bench (summaries in place) exit $?"; tail -n 1 $O/bench_with_pmc.log | cut -c1-200
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-fp16 --no-fp32c --mode normal 2>&1 | tail -1 > $O/bench_normal.log; cut -c1-160 $O/bench_normal.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-fp16 --no-fp32c --head dpt 2>&1 | tail -1 > $O/bench_dpt.log; cut -c1-160 $O/bench_dpt.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-fp16 --no-fp32c --batch 8 2>&1 | tail -1 > $O/bench_b8.log; cut -c1-160 $O/bench_b8.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-fp16 --no-fp32c --head dpt --precision fp32c 2>&1 | tail -1 > $O/bench_dpt_fp32c.log; cut -c1-160 $O/bench_dpt_fp32c.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-fp16 --no-fp32c --mode normal --precision fp32c 2>&1 | tail -1 > $O/bench_normal_fp32c.log; cut -c1-160 $O/bench_normal_fp32c.log
timeout 300 python tools/launch_log.py --tag r06 > $O/launch_log_run.log 2>&1; echo "== launch_log exit $?"; tail -n 3 $O/launch_log_run.log
timeout 300 python tools/launch_log.py --tag r06 --precision fp32c > $O/launch_log_fp32c_run.log 2>&1; echo "== launch_log fp32c exit $?"; tail -n 3 $O/launch_log_fp32c_run.log
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete; du -sh $O
