#!/bin/bash
# ARCHIVED (round 6): the WD_ABL compile-time ablation masks of the f32-A kernel left csrc/det_gemm_split.hip with the other laboratory code; this script works at
# commit 8ca8714 (round 5).  The planes kernel has its own: tools/split_planes_ablation_build.py + tools/split_planes_ablation.sh.
# Cumulative compile-time ablations of the split-operand ring kernel (timing only; results are wrong by construction):
#   tools/split_ablation.sh "9600 1024 1024"      (builds csrc/variants/lib_abl<N>.so for the WD_ABL masks below when missing)
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=${1:-"9600 1024 1024"}
for mask in 0 1 2 4 8 3 7 15 16 32 48; do
  lib=$R/waymo_2d_tracking_amd/csrc/variants/lib_abl$mask.so
  [ -f $lib ] || bash $R/tools/build_variant.sh abl$mask "-DWD_ABL=$mask" det_gemm_split.hip > /dev/null
  echo -n "WD_ABL=$mask  "
  WT_LIB_PATH=$lib python3 $R/tools/gemm_split_one.py $SHAPE 20 0 2>&1 | grep gemm_split | tr '\n' ' '
  WT_LIB_PATH=$lib python3 $R/tools/gemm_split_stamps.py $SHAPE 0 2>&1 | grep "^wall"
done
