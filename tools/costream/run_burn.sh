#!/bin/bash
# GPU box: the two known victims next to the pure-register matrix-instruction burner (debug library) - is anything but the matrix instructions needed?
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_costream
mkdir -p $O
export WT_LIB_PATH=$PWD/waymo_2d_tracking_amd/csrc/libwaymotrack_debug.so WT_EXPERIMENT=1
for victim in deform64 gconv; do
  for kind in ${KINDS:-1 2 3 4 5 6 7 8 9}; do
    for iters in 400; do
      echo "== victim $victim aggressor burn kind $kind ($iters iterations x 16 MFMAs, 512 workgroups)"
      AGGRESSOR=burn$kind BURN_ITERS=$iters VICTIM=$victim timeout 300 python tools/archive/diag_victim.py 2>&1 | grep -v amdgpu.ids | tail -5
    done
  done
done > $O/burn${TAG}.txt 2>&1
if [ -n "$TABLE" ]; then for k in $TABLE; do echo "== all victims next to burn kind $k"; AGGRESSOR=burn$k timeout 900 python tools/costream/victims_table.py 2>&1 | grep -v amdgpu.ids | tail -30; done > $O/burn_table.txt 2>&1; fi
