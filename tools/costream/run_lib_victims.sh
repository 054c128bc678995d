#!/bin/bash
# GPU box: the library kernels left in the frame as victims next to the split-operand kernel (two frames in flight put them beside it)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_costream
mkdir -p $O
export WT_EXPERIMENT=1
for mt in 4 5 6; do
  for agg in res2 res4; do
    echo "== aggressor $agg MT=$mt"
    WD_SPLIT_MT=$mt AGGRESSOR=$agg timeout 600 python tools/costream/victims_table.py lib_stem_conv7x7_s2 lib_offset_conv3x3_s2 lib_rpn_predictor_gemm lib_max_pool lib_softmax offset_conv 2>&1 | grep -v amdgpu.ids | tail -8
  done
done > $O/part4_library_victims.txt 2>&1
cat $O/part4_library_victims.txt
