#!/bin/bash
# Round-4 GPU session 18: sustained (seconds, not milliseconds) runs of three inner loops with socket power and clock sampled beside them:
# halo3's loop (64 x 64 tile, two waves per SIMD, 0.5 reads per MFMA), the 128 x 128-per-wave loop (one wave per SIMD, half the fragment traffic per
# flop), and MFMAs alone -- TFLOP/s at the power limit and watts per TFLOP/s are what decides between structures in the pipeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s18; rm -rf $O; mkdir -p $O
( while true; do echo "{\"unix_time\": $(date +%s.%N), \"smi\": $(rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n')}" >> $O/smi.jsonl; sleep 0.25; done ) & SMI=$!
timeout 60 tools/probe/wave1_tile_probe 3 > $O/sustained.jsonl 2> $O/err.log
kill $SMI; wait $SMI 2>/dev/null
cat $O/sustained.jsonl | tail -30; wc -l $O/smi.jsonl
