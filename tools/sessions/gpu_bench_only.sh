#!/bin/bash
# the driver's bench command alone (after profiles/r03_pmc_traffic_summary.json of the same build is in place: roofline.traffic is then non-null)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_only.log 2>&1; echo "== bench exit $?"; tail -n 1 gpurun_out/bench_only.log | cut -c1-300
