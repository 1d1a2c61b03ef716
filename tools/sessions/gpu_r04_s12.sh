#!/bin/bash
# Round-4 GPU session 12: the MFMA + LDS-read probe (tools/mfma_lds_probe.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s12; rm -rf $O; mkdir -p $O
timeout 300 python tools/mfma_lds_probe.py > $O/mfma_lds_probe.json 2> $O/probe.err; echo "exit $?"; tail -3 $O/probe.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04s12/mfma_lds_probe.json"))
for r in d["rows"]: print(r)
for r in d["build_up_at_2_waves_per_simd"]: print(r)
PY
