#!/bin/bash
# GPU box: second series of victim-side variants - is the victims' LDS data late (delay behind the fill / slab barriers), does the grouped conv fail without LDS-DMA?
# aggressor: split kernel MT = 2 (product library's kernel, via WT_EXPERIMENT) and the zero-operand burner (debug library: only with the product victims)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_costream
mkdir -p $O
V=waymo_2d_tracking_amd/csrc/variants
export WT_EXPERIMENT=1
for lib in product v_delay1 v_delay8 v_nodma; do
  for victim in deform64 gconv; do
    if [ $lib = v_nodma ]; then [ $victim = gconv ] || continue; fi
    L=""; [ $lib = product ] || L=$PWD/$V/lib_$lib.so
    echo "== lib $lib victim $victim aggressor split MT=2"
    WT_LIB_PATH=$L WD_SPLIT_MT=2 AGGRESSOR=split2 VICTIM=$victim timeout 300 python tools/archive/diag_victim.py 2>&1 | grep -v amdgpu.ids | tail -2
  done
done > $O/part5_series2.txt 2>&1
cat $O/part5_series2.txt
