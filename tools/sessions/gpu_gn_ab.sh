G="gn:4,589824,128,1 gn:4,147456,256,1 gn:4,147456,512,1 gn:4,36864,512,1 gn:4,9216,320,1 gn:4,9216,640,1 gn:4,9216,960,1 gn:4,2304,1280,1 gn:4,2304,1920,1 gn:4,576,2560,1 gn:4,9216,320,0 xfold:36864,320,5 xfold:9216,640,10 xfold:2304,1280,20"
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
echo NEW; timeout 200 tools/kbench iters=20 $G
echo OLD; GENPERCEPT_GN_APPLY_OLD=1 timeout 200 tools/kbench iters=20 $G
