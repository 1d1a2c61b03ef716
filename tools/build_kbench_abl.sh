#!/bin/bash
# builds tools/kbench_abl: kbench against a bf16 library whose conv_halo.hip carries the profiling ablations (-DGP_HALO_ABLATIONS=1: IGemmParams::dbg
# bits 9-19 and 24-27); the product library under genpercept_amd/lib/ is not touched
set -e
cd "$(dirname "$0")/.."
python -m genpercept_amd.build >/dev/null
L=genpercept_amd/lib/abl; mkdir -p $L
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DGP_HALO_ABLATIONS=1 -c genpercept_amd/csrc/conv_halo.hip -o $L/conv_halo.o
OBJS=$(ls genpercept_amd/lib/obj_bf16/*.o | grep -v "/conv_halo.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $L/conv_halo.o -o $L/libgenpercept_hip.so
DEFS=""
grep -q gp_gemm_qkv include/genpercept_hip.h && DEFS="-DKBENCH_HAVE_QKV"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 $DEFS tools/kbench.cpp -o tools/kbench_abl \
  -L$L -lgenpercept_hip -Wl,-rpath,'$ORIGIN/../genpercept_amd/lib/abl'
echo built tools/kbench_abl
