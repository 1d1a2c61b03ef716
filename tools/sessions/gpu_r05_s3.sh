#!/bin/bash
# Round-5 GPU session 3: the fused GroupNorm+SiLU conv with the halo issued first (certified one step earlier) and the 32 left-over items as a sixth
# part in tap 2: parity tests, kernel-only durations (rocprofv3 --kernel-trace on kbench), and a same-box pipeline A/B against the library built
# before the change (genpercept_amd/lib/base, selected with GENPERCEPT_HIP_LIB), alternating.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r05s3; rm -rf $O; mkdir -p $O
for L in new base; do
  E=""; [ $L = base ] && E="LD_LIBRARY_PATH=$ROOTD/genpercept_amd/lib/base"
  (cd /tmp && env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$L -- $ROOTD/tools/kbench iters=20 cold=1 check=0 convg:4,768,768,128,128,1 convg:4,768,768,256,128,1 convg:4,768,768,128,128,0 > $O/kbench_$L.log 2>&1)
  grep -E "^conv" $O/kbench_$L.log
done
python3 - <<'PY'
import csv, glob, collections, statistics
for L in ("new", "base"):
    f = glob.glob(f"gpurun_out/r05s3/prof_{L}/**/*kernel_trace.csv", recursive=True)
    g = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "halo3" in n: g[(n, r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, ds in sorted(g.items()):
        ds = sorted(ds); h = len(ds) // 2
        print(L, k[0], "n", len(ds), "median lower half %.1f upper half %.1f" % (statistics.median(ds[:h]), statistics.median(ds[h:])))
PY
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "fused_groupnorm or heaviest or conv_epilogue" --timeout=500 -p no:cacheprovider > $O/pytest_k.log 2>&1; echo "== tests exit $?"; tail -n 3 $O/pytest_k.log
timeout 600 python -m pytest tests/test_e2e_gpu.py -q --timeout=500 -p no:cacheprovider > $O/pytest_e2e.log 2>&1; echo "== e2e exit $?"; tail -n 3 $O/pytest_e2e.log
for E in "new:" "base:GENPERCEPT_HIP_LIB=$ROOTD/genpercept_amd/lib/base/libgenpercept_hip.so" "new2:" "base2:GENPERCEPT_HIP_LIB=$ROOTD/genpercept_amd/lib/base/libgenpercept_hip.so"; do
  env ${E#*:} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-fp16 2>&1 | tail -1 > $O/bench_${E%%:*}.log
  python3 -c "import json; d=json.load(open('$O/bench_${E%%:*}.log')); print('${E%%:*}', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['stages']['ms_encode'], d['stages']['ms_unet'], d['stages']['ms_head'], d['clock_power']['sclk_mhz_mean'])"
done
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
