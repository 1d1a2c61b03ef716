#!/bin/bash
# copy the outputs of tools/gpu_r04_final.sh (gpurun_out/r04final) into profiles/ under their round-4 names
cd "$(dirname "$0")/.."; O=gpurun_out/r04final
tail -1 $O/bench.log > profiles/r04_bench_depth_b4_768_session.json
cp $O/kernel_stats.csv profiles/r04_bench_b4_768_kernel_stats.csv
cp $O/kernel_trace_last_pass.tsv profiles/r04_kernel_trace_b4_768.tsv
cp $O/pmc_traffic_summary.json profiles/r04_pmc_traffic_summary.json
cp $O/bench_normal.log profiles/r04_bench_normal_b4_768.json
cp $O/bench_dpt.log profiles/r04_bench_dpt_b4_768.json
cp $O/bench_b8.log profiles/r04_bench_depth_b8_768.json
cp gpurun_out/launch_log_r04_vae_b4_768.txt profiles/r04_launch_log_b4_768.txt
[ -s $O/mfma_lds_probe.json ] && cp $O/mfma_lds_probe.json profiles/r04_mfma_lds_probe.json
python3 - <<'PY'
import json, bench
d = json.load(open('profiles/r04_bench_depth_b4_768_session.json'))
print('session line:', d['value'], 'img/s', d['ms_per_step'], 'ms; fp16', d['value_fp16'], d['stages'])
print('build id', bench.source_build_id(), 'pmc', json.load(open('profiles/r04_pmc_traffic_summary.json'))['build_id'])
PY
