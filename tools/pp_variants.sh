#!/bin/bash
# persistent deformable-conv variants side by side (tools only): usage pp_variants.sh name1 name2 ...  (csrc/variants/lib_pp_<name>.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pp_variants.txt
: > $OUT
for n in "$@"; do
  echo "== $n" >> $OUT
  WT_LIB_PATH=$R/waymo_2d_tracking_amd/csrc/variants/lib_pp_$n.so SHAPES=${SHAPES:-res4} STDS=${STDS:-0.2,1,2} python3 $R/tools/deform_r3_bench.py 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
