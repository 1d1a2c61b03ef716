#!/bin/bash
# Round-5 GPU session 2: (1) per-kernel durations (rocprofv3 --kernel-trace --stats on kbench) of the fused GroupNorm conv with and without the SiLU
# (FUSED 2 / 1) next to the plain conv: how much of the fused kernel's penalty is the transform's VALU load?  (2) tests of cross_fold with several rows
# per wave at C = 1280 and of gn_small_reg_kernel; (3) per-launch log of one pass.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r05s2; rm -rf $O; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $ROOTD/tools/kbench iters=20 cold=1 check=0 convg:4,768,768,128,128,1 convg:4,768,768,128,128,0 conv:4,768,768,128,128 convg:4,768,768,256,128,1 convg:4,768,768,256,128,0 conv:4,768,768,256,128 > $O/kbench_prof.log 2>&1)
F=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $O/kernel_stats.csv && cut -c1-200 $O/kernel_stats.csv | head -14
grep -E "^conv" $O/kbench_prof.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r05s2/prof/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# consecutive runs of the same kernel name = one kbench spec: average duration per run
runs, cur = [], None
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if cur and cur[0] == n: cur[1].append(d)
    else:
        cur = [n, [d]]; runs.append(cur)
for n, ds in runs:
    if len(ds) >= 5 and ("halo3" in n or "gn_" in n): print(f"{n:62s} n={len(ds):3d} avg {sum(ds)/len(ds):8.1f} us  min {min(ds):8.1f}")
PY
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "two_token_fold or groupnorm" --timeout=500 -p no:cacheprovider > $O/pytest_k.log 2>&1; echo "== tests exit $?"; tail -n 3 $O/pytest_k.log
timeout 300 python tools/launch_log.py --tag r05s2 > $O/launch_log_run.log 2>&1; echo "== launch_log exit $?"; tail -n 3 $O/launch_log_run.log
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
