#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
X="xfold:36864,320,5 xfold:9216,640,10 xfold:2304,1280,20 xfold:576,1280,20 xfold:73728,320,5"
echo NEW; timeout 100 tools/kbench iters=20 $X | grep xfold
echo OLD; GENPERCEPT_XFOLD_LDS=0 timeout 100 tools/kbench iters=20 $X | grep xfold
