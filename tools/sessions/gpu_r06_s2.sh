#!/bin/bash
# Round-6 session 2: (1) the tightened regression gates on the product build (full-size, e2e, multi-step, outlier stress); (2) gate sensitivity: the
# same full-size bf16 tests against a library whose halo conv stores one mantissa bit less (tools/build_round_abl.sh) -- they must FAIL.
# usage: gpurun --timeout 1800 -- 'bash tools/sessions/gpu_r06_s2.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r06s2
rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_log.jsonl
timeout 1200 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_e2e_gpu.py tests/test_multistep_gpu.py tests/test_outlier_stress_gpu.py tests/test_refexec_gpu.py -q --timeout=900 -p no:cacheprovider > $O/pytest_gates.log 2>&1
echo "== gates exit $?"; tail -n 8 $O/pytest_gates.log | cut -c1-200
cp gpurun_out/parity_log.jsonl $O/parity_product.jsonl; rm -f gpurun_out/parity_log.jsonl
GENPERCEPT_HIP_LIB=$ROOTD/genpercept_amd/lib/abl_round/libgenpercept_hip.so timeout 600 python -m pytest tests/test_fullsize_parity_gpu.py -k "bf16 and (768_depth or dpt_disparity)" -q --timeout=900 -p no:cacheprovider > $O/pytest_abl_round.log 2>&1
echo "== abl_round exit $? (expected: 1 = the gates caught the doubled rounding)"; tail -n 6 $O/pytest_abl_round.log | cut -c1-200
cp gpurun_out/parity_log.jsonl $O/parity_abl_round.jsonl
python3 - <<'PY'
import json
def load(p):
    return {json.loads(l)["test"]: json.loads(l) for l in open(p)}
a, b = load("gpurun_out/r06s2/parity_product.jsonl"), load("gpurun_out/r06s2/parity_abl_round.jsonl")
out = {"what": "gate sensitivity (VERDICT r5 item 5): tests/test_fullsize_parity_gpu.py, bf16 library, product build vs a build whose conv3x3_halo kernels "
               "round their stored outputs to one mantissa bit less (tools/build_round_abl.sh: -DGP_ROUND_ABL=1 on conv_halo.hip only); gates = 1.25x the product's values",
       "rows": {}}
for k in b:
    if k in a and "mean_abs" in b[k]:
        out["rows"][k] = {"product": {m: a[k][m] for m in ("mean_abs", "rel_rms") if m in a[k]}, "halo_rounding_doubled": {m: b[k][m] for m in ("mean_abs", "rel_rms") if m in b[k]}}
        print(k, out["rows"][k])
json.dump(out, open("gpurun_out/r06s2/gate_sensitivity.json", "w"), indent=1)
PY
