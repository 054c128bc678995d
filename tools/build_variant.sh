#!/bin/bash
# build csrc/variants/lib_<name>.so with extra hipcc flags for ONE unit (kernel experiments; not shipped); load it with WT_LIB_PATH
# usage: tools/build_variant.sh name "-DFLAG ..." [unit.hip, default det_deform_pp.hip]
set -e
cd "$(dirname "$0")/../waymo_2d_tracking_amd/csrc"
mkdir -p variants
unit=${3:-det_deform_pp.hip}
base=${unit%.hip}
extra=""
case $unit in det_deform_pp.hip) extra="-fno-slp-vectorize";; det_gemm.hip|det_backward.hip|det_deform_bwd.hip) extra="-munsafe-fp-atomics";; sort_*|ensemble.hip|det_tail.hip|det_preprocess.hip) extra="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math $extra $2 -c $unit -o variants/${base}_$1.o
objs=$(ls *.o | grep -v "^${base}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$1.so $objs variants/${base}_$1.o
echo variants/lib_$1.so
