#!/bin/bash
# copy the outputs of tools/sessions/gpu_r05_final.sh (gpurun_out/r05final) into profiles/ under their round-5 names
cd "$(dirname "$0")/../.."; O=gpurun_out/r05final
tail -1 $O/bench.log > profiles/r05_bench_depth_b4_768_session.json
cp $O/kernel_stats.csv profiles/r05_bench_b4_768_kernel_stats.csv
cp $O/kernel_trace_last_pass.tsv profiles/r05_kernel_trace_b4_768.tsv
cp $O/pmc_traffic_summary.json profiles/r05_pmc_traffic_summary.json
cp $O/bench_normal.log profiles/r05_bench_normal_b4_768.json
cp $O/bench_dpt.log profiles/r05_bench_dpt_b4_768.json
cp $O/bench_b8.log profiles/r05_bench_depth_b8_768.json
cp gpurun_out/launch_log_r05_vae_b4_768.txt profiles/r05_launch_log_b4_768.txt
[ -s $O/mfma_lds_probe.json ] && cp $O/mfma_lds_probe.json profiles/r05_mfma_lds_probe.json
python3 - <<'PY'
import json, bench
d = json.load(open('profiles/r05_bench_depth_b4_768_session.json'))
print('session line:', d['value'], 'img/s', d['ms_per_step'], 'ms; fp16', d['value_fp16'], d['stages'])
print('build id', bench.source_build_id(), 'pmc', json.load(open('profiles/r05_pmc_traffic_summary.json'))['build_id'])
PY
