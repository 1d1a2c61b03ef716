#!/bin/bash
# rocprofv3 kernel stats over a kbench invocation: usage gpu_kstats.sh <kbench args...>
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/kstats; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; ROOTD=$(pwd)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/$O" -- "$ROOTD/tools/kbench" "$@" > "$ROOTD/$O/run.log" 2>&1)
F=$(find $O -name "*kernel_stats.csv" | head -1); cut -d, -f1-4,6,7 "$F" | grep -v "fill_kernel\|ref_gemm\|cmp_" | head -12 | cut -c1-200
