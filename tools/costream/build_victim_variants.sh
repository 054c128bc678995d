#!/bin/bash
# Round 6, VERDICT weak #3: victim-side experiments for the co-residency corruption.  The experiment switches (-DWD_VICTIM_SYNC / _M0 / _LB / _CHECK) are NOT
# in the product sources: tools/costream/victim_variants.patch adds them to COPIES of det_gconv.hip / det_deform.hip, which are rebuilt with the flags
# below and linked with the product's other objects into csrc/variants/lib_<name>.so (load with WT_LIB_PATH).  a_f32 = aggressor with every bf16 MFMA
# replaced by an f32 MFMA (needs the round-5 WD_ABL switch: build it from commit 8ca8714).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT/waymo_2d_tracking_amd/csrc"
mkdir -p variants/src
cp det_gconv.hip det_deform.hip common.h variants/src/
(cd "$ROOT" && sed 's#waymo_2d_tracking_amd/csrc/#waymo_2d_tracking_amd/csrc/variants/src/#g' tools/costream/victim_variants.patch | patch -p1 -s)
sed -i 's#"../../include/#"../../../../include/#' variants/src/det_gconv.hip variants/src/det_deform.hip variants/src/common.h
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math"
build() {   # name, flags, units...
    name=$1; flags=$2; shift 2
    skip=""
    extra=""
    for u in "$@"; do
        b=${u%.hip}
        $HIPCC $flags -c variants/src/$u -o variants/${b}_$name.o &
        skip="$skip -e ^${b}.o\$"
        extra="$extra variants/${b}_$name.o"
    done
    wait
    objs=$(ls *.o | grep -v '\.dbg\.o$' | grep -v $skip)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$name.so $objs $extra
    rm -f $extra
    echo variants/lib_$name.so
}
[ -n "$SERIES2" ] || build v_fz "-mllvm -amdgpu-waitcnt-forcezero" det_gconv.hip det_deform.hip
[ -n "$SERIES2" ] || build v_sync "-DWD_VICTIM_SYNC=1" det_gconv.hip det_deform.hip
[ -n "$SERIES2" ] || build v_m0 "-DWD_VICTIM_M0=1" det_gconv.hip
[ -n "$SERIES2" ] || build v_lb1 "-DWD_VICTIM_LB=1" det_gconv.hip det_deform.hip
[ -n "$SERIES2" ] || build v_check "-DWD_VICTIM_CHECK=1" det_gconv.hip
# round 6, second series: is the victims' LDS data LATE?  delay behind the fill barrier / no LDS-DMA at all
build v_delay1 "-DWD_VICTIM_DELAY=1" det_gconv.hip det_deform.hip
build v_delay8 "-DWD_VICTIM_DELAY=8" det_gconv.hip det_deform.hip
build v_nodma "-DWD_VICTIM_NODMA=1" det_gconv.hip
