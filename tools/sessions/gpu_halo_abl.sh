#!/bin/bash
# halo3 conv ablations (GENPERCEPT_IGEMM_DBG = 512 * ABL, see conv_halo.hip; library built with GENPERCEPT_HIPCC_FLAGS=-DGP_HALO_ABLATIONS=2)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
S="${SPECS:-conv:4,768,768,128,128 conv:4,384,384,256,256 conv:4,192,192,512,512}"
{
for A in ${ABLS:-0 4 8 16 24 0}; do echo "== ABL $A"; GENPERCEPT_IGEMM_DBG=$((512*A)) timeout 120 tools/kbench iters=20 cold=1 check=${CHECK:-0} $S | grep -vE "^#"; done
} > gpurun_out/halo_abl.log 2>&1
cat gpurun_out/halo_abl.log
