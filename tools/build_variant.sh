#!/bin/bash
# build csrc/variants/lib_<name>.so with extra hipcc flags for det_deform_pp.hip (kernel experiments; not shipped)
# usage: tools/build_variant.sh name "-DFLAG ..."
set -e
cd "$(dirname "$0")/../waymo_2d_tracking_amd/csrc"
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math $2 -c det_deform_pp.hip -o variants/pp_$1.o
objs=$(ls *.o | grep -v det_deform_pp.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$1.so $objs variants/pp_$1.o
echo variants/lib_$1.so
