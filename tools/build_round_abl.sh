#!/bin/bash
# builds genpercept_amd/lib/abl_round/libgenpercept_hip.so: the bf16 library with ONE kernel's rounding doubled -- conv_halo.hip compiled with
# -DGP_ROUND_ABL=1 (common.h: every fp32 -> bf16 conversion of that translation unit drops one more mantissa bit) -- to show that the regression gates
# of tests/test_fullsize_parity_gpu.py fail for it (profiles/r06_gate_sensitivity.json).  The product library is not touched.
set -e
cd "$(dirname "$0")/.."
python -m genpercept_amd.build >/dev/null
L=genpercept_amd/lib/abl_round; mkdir -p $L
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DGP_ROUND_ABL=1 -c genpercept_amd/csrc/conv_halo.hip -o $L/conv_halo.o
OBJS=$(ls genpercept_amd/lib/obj_bf16/*.o | grep -v "/conv_halo.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $L/conv_halo.o -o $L/libgenpercept_hip.so
echo built $L/libgenpercept_hip.so
