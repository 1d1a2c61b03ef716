#!/bin/bash
# Round-6 last session: SQ counters + HBM traffic of the FINAL sources (build id stamped), then the driver's bench command with those summaries in place.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r06last
rm -rf $O; mkdir -p $O
bash tools/sessions/gpu_r06_pmc.sh > $O/pmc_session.log 2>&1; echo "== pmc exit $?"; tail -n 16 $O/pmc_session.log | cut -c1-150
cp gpurun_out/r06pmc/pmc_mfma_summary.json profiles/r06_pmc_mfma_summary.json; cp gpurun_out/r06pmc/pmc_traffic_summary.json profiles/r06_pmc_traffic_summary.json
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "== bench exit $?"; tail -n 1 $O/bench.log | cut -c1-300
