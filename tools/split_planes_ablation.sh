#!/bin/bash
# GPU box: timing-only ablations of gemm_split_planes_kernel (results wrong by construction); build the variant libraries first (here, on the CPU):
#   python tools/split_planes_ablation_build.py     - an experiment copy of
# csrc/det_gemm_split.hip with -DWD_PL_ABL=mask: 1 no W loads / waits, 2 no LDS-DMA, 4 no barrier, 8 no MFMAs, 16 no fragment reads
cd "$(dirname "$0")/.."
O=gpurun_out/r06_planes/ablation.txt
mkdir -p gpurun_out/r06_planes
: > $O
for m in 0 1 2 3 4 7 16; do
  for shape in "9600 1024 1024" "38400 512 512"; do
    WT_LIB_PATH=$PWD/waymo_2d_tracking_amd/csrc/variants/lib_plabl$m.so timeout 120 python tools/gemm_planes_one.py $shape 20 0 2>&1 | grep "us |" | sed "s/^/[WD_PL_ABL=$m] /" >> $O
  done
done
cat $O
