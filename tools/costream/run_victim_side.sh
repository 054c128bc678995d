#!/bin/bash
# GPU box: victim-side experiments (part 1) and the widened victim table (part 2).  Output: gpurun_out/r06_costream/*.txt
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_costream
mkdir -p $O
V=waymo_2d_tracking_amd/csrc/variants
export WT_EXPERIMENT=1
part1() {
for lib in product v_fz v_sync v_m0 v_lb1 v_check a_f32; do
  for victim in deform64 gconv; do
    if [ $lib = v_m0 ] || [ $lib = v_check ]; then [ $victim = gconv ] || continue; fi
    L=""; [ $lib = product ] || L=$PWD/$V/lib_$lib.so
    echo "== lib $lib victim $victim aggressor split MT=2" 
    WT_LIB_PATH=$L WD_SPLIT_MT=2 AGGRESSOR=split2 VICTIM=$victim timeout 300 python tools/archive/diag_victim.py 2>&1 | tail -8
  done
done
}
part2() {
for mt in 4 5 6; do
  for agg in res2 res4; do
    echo "== aggressor $agg MT=$mt"
    WD_SPLIT_MT=$mt AGGRESSOR=$agg timeout 600 python tools/costream/victims_table.py 2>&1 | tail -30
  done
done
echo "== control: no aggressor"
AGGRESSOR=none REPS=50 timeout 600 python tools/costream/victims_table.py 2>&1 | tail -30
echo "== positive control: MT=2 aggressor, all victims"
WD_SPLIT_MT=2 AGGRESSOR=res2 timeout 600 python tools/costream/victims_table.py 2>&1 | tail -30
}
case "$1" in 1) part1 > $O/part1.txt 2>&1;; 2) part2 > $O/part2.txt 2>&1;; *) part1 > $O/part1.txt 2>&1; part2 > $O/part2.txt 2>&1;; esac
