#!/bin/bash
# Round-5 GPU session 5: cheap tuning sweeps through existing switches (kbench, weight-cold): K-slice count of conv_img_kernel on the 12x12 / 24x24 maps
# (GENPERCEPT_CONV_IMG_S), K / V ring depth of flash_attn64 at the UNet's short sequences (GENPERCEPT_FLASH_RING3)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05s5; rm -rf $O; mkdir -p $O
for S in 0 4 6 8 10 12 16; do
  echo "== conv_img S=$S (0 = heuristic)"; GENPERCEPT_CONV_IMG_S=$S timeout 120 tools/kbench iters=40 cold=1 check=0 conv:4,12,12,1280,1280 conv:4,12,12,2560,1280 | grep -vE "^#" | sed "s/^/S$S /" | tee -a $O/conv_img_12.log
done
for S in 0 2 3 4 5 6 8; do
  echo "== conv_img S=$S"; GENPERCEPT_CONV_IMG_S=$S timeout 120 tools/kbench iters=40 cold=1 check=0 conv:4,24,24,1280,1280 conv:4,24,24,2560,1280 conv:4,24,24,640,1280 | grep -vE "^#" | sed "s/^/S$S /" | tee -a $O/conv_img_24.log
done
for R in 0 1; do
  E=""; [ $R = 1 ] && E="GENPERCEPT_FLASH_RING3=1"
  echo "== flash ring3=$R"; env $E timeout 120 tools/kbench iters=40 check=0 attn:4,144,20 attn:4,576,20 attn:4,2304,10 attn:4,9216,5 | grep -vE "^#" | sed "s/^/ring3=$R /" | tee -a $O/flash.log
done
