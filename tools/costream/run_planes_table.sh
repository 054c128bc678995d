#!/bin/bash
# GPU box: every product kernel as a victim next to the round-6 planes kernel (fewer registers than the f32-A kernel at the same tile height)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_costream
mkdir -p $O
export WT_EXPERIMENT=1
for mt in 4 5 6; do
  for agg in planes_res2 planes_res4; do
    echo "== aggressor $agg MT=$mt"
    WD_SPLIT_MT=$mt AGGRESSOR=$agg timeout 600 python tools/costream/victims_table.py 2>&1 | grep -v amdgpu.ids | tail -30
  done
done > $O/part3_planes.txt 2>&1
