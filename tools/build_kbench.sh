#!/bin/bash
# builds tools/kbench (torch-free kernel micro-benchmark) against the in-tree bf16 library; the binary travels with the gpurun snapshot
set -e
cd "$(dirname "$0")/.."
python -m genpercept_amd.build >/dev/null
DEFS=""
grep -q gp_gemm_qkv include/genpercept_hip.h && DEFS="-DKBENCH_HAVE_QKV"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 $DEFS tools/kbench.cpp -o tools/kbench \
  -Lgenpercept_amd/lib -lgenpercept_hip -Wl,-rpath,'$ORIGIN/../genpercept_amd/lib'
echo built tools/kbench
