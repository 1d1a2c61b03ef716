#!/bin/bash
# Round-4 GPU session 3: why is the 512-pixel-tile conv 5 % behind?  kbench A/B (halo3 / halo4 two pixel sets / halo4 three pixel sets), PMC of
# halo3 vs halo4 on one shape, and pipeline A/Bs (GroupNorm fusion policy, halo4 auto).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s3; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; ROOTD=$(pwd)
S="conv:4,384,384,256,256 conv:4,768,768,256,128 conv:4,384,384,512,256"
for rep in 1 2; do
  for V in "halo3:$((2<<20))" "halo4x2:$((1<<20))" "halo4x3:$(((1<<20)+(1<<22)))"; do
    echo "== ${V%%:*} rep $rep"; GENPERCEPT_IGEMM_DBG=${V##*:} timeout 200 tools/kbench iters=20 cold=1 check=0 $S | grep -vE "^#" | tee -a $O/kbench_${V%%:*}.log
  done
done
pmc() {  # tag dbg
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT" \
             "GRBM_GUI_ACTIVE"; do
    N=$(echo $SET | cut -c1-12 | tr ' ' '_')
    (cd /tmp && GENPERCEPT_IGEMM_DBG=$2 timeout 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$ROOTD/$O/pmc_$1_$N" -- "$ROOTD/tools/kbench" iters=5 cold=1 check=0 conv:4,384,384,256,256 > "$ROOTD/$O/pmc_$1_$N.log" 2>&1)
  done
}
pmc halo3 $((2<<20)); pmc halo4 $((1<<20))
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for tag in ("halo3", "halo4"):
    cnt = collections.OrderedDict(); dur = [0, 0.0]
    for d in sorted(glob.glob(f"{O}/pmc_{tag}_*/")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if 'conv3x3_halo' not in r.get('Kernel_Name', ''): continue
                a = cnt.setdefault(r['Counter_Name'], [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if 'conv3x3_halo' not in r.get('Kernel_Name', ''): continue
                dur[0] += 1; dur[1] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    us = dur[1] / max(dur[0], 1) / 1e3
    print(tag, 'dur %.1f us' % us, ' '.join('%s=%.5g' % (k, v[1] / v[0]) for k, v in cnt.items()))
PY
find $O -name "*.csv" -size +1M -delete
for E in "default:" "nofuse:GENPERCEPT_GN_FUSE_MAX_SLICES=0" "halo4:GENPERCEPT_HALO4=1" "nofuse_halo4:GENPERCEPT_GN_FUSE_MAX_SLICES=0 GENPERCEPT_HALO4=1" "default2:"; do
  echo "== bench ${E%%:*}"; env ${E#*:} timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-fp16 2>&1 | tail -1 > $O/bench_${E%%:*}.log
  python3 -c "import json,sys; d=json.load(open('$O/bench_${E%%:*}.log')); print(d['value'], d['ms_per_step'], d['stages'], d['roofline']['achieved'], d['roofline']['launches'])"
done
