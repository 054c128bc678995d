#!/bin/bash
# GPU box: frames in flight (hipGraph lanes on separate streams) A/B for the bench stages.  usage: tools/inflight_ab.sh name bench-args... (repeatable via ;)
cd "$(dirname "$0")/.."
export WT_EXPERIMENT=1
O=gpurun_out/r06_inflight
mkdir -p $O
run() {
  name=$1; shift
  timeout 500 python bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err
  echo "$name rc=$?"
  python - "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('    %.2f %s  %.1f ms/step  verified %s' % (d['value'], d['unit'], d['ms_per_step'], (d.get('verified') or {}).get('ok')))
except Exception as e:
    print('    no line', e)
PY
}
case "$1" in
  sweep)
    run e2e_if3 --inflight 3 --steps 10 --warmup 3
    run e2e_if4 --inflight 4 --steps 10 --warmup 3
    run detect_if2 --stage detect --inflight 2 --steps 10 --warmup 3
    run detect_if1 --stage detect --inflight 1 --steps 10 --warmup 3
    run tta_if2 --stage detect --tta x1.5,hflip --auto-contrast --inflight 2 --steps 6 --warmup 2
    run tta_if1 --stage detect --tta x1.5,hflip --auto-contrast --inflight 1 --steps 6 --warmup 2
    run jpeg_if2 --from-jpeg --inflight 2 --steps 10 --warmup 3
    ;;
  lanes)
    run e2e_l2_a --inflight 2 --steps 10 --warmup 3
    run e2e_l3_a --inflight 3 --steps 10 --warmup 3
    run e2e_l2_b --inflight 2 --steps 10 --warmup 3
    run e2e_l3_b --inflight 3 --steps 10 --warmup 3
    run e2e_l1 --inflight 1 --steps 10 --warmup 3
    ;;
  planes)
    run e2e_f32act_a --steps 10 --warmup 3
    WD_SPLIT_PLANES=1 run e2e_planes_a --steps 10 --warmup 3
    run e2e_f32act_b --steps 10 --warmup 3
    WD_SPLIT_PLANES=1 run e2e_planes_b --steps 10 --warmup 3
    WD_SPLIT_PLANES=1 WD_SPLIT_PLANES_MIN_CH=512 run e2e_planes_res3_too --steps 10 --warmup 3
    ;;
  xmap)
    run e2e_xmap0_a --steps 10 --warmup 3
    WD_SPLIT_XMAP=1 run e2e_xmap1_a --steps 10 --warmup 3
    run e2e_xmap0_b --steps 10 --warmup 3
    WD_SPLIT_XMAP=1 run e2e_xmap1_b --steps 10 --warmup 3
    ;;
  instr)
    run e2e_instr_all_a --steps 10 --warmup 3
    WT_BENCH_INSTRUMENT_STEPS=2 run e2e_instr_2_a --steps 10 --warmup 3
    WT_BENCH_INSTRUMENT_STEPS=0 run e2e_instr_0_a --steps 10 --warmup 3
    run e2e_instr_all_b --steps 10 --warmup 3
    WT_BENCH_INSTRUMENT_STEPS=2 run e2e_instr_2_b --steps 10 --warmup 3
    WT_BENCH_INSTRUMENT_STEPS=0 run e2e_instr_0_b --steps 10 --warmup 3
    ;;
  posmajor)
    for mt in 4 5 6; do
      WD_SPLIT_MT=$mt python tools/conv_split_one.py 1000 256 7 7 256 2>&1 | grep conv3x3 | sed "s/^/position-major  /"
      WD_SPLIT_MT=$mt WD_SPLIT_NO_POSMAJOR=1 python tools/conv_split_one.py 1000 256 7 7 256 2>&1 | grep conv3x3 | sed "s/^/plain row order /"
    done
    run e2e_posmajor_a --steps 10 --warmup 3
    WD_SPLIT_NO_POSMAJOR=1 run e2e_plainrows_a --steps 10 --warmup 3
    run e2e_posmajor_b --steps 10 --warmup 3
    WD_SPLIT_NO_POSMAJOR=1 run e2e_plainrows_b --steps 10 --warmup 3
    run train_posmajor_a --stage train --steps 8 --warmup 4
    WD_SPLIT_NO_POSMAJOR=1 run train_plainrows_a --stage train --steps 8 --warmup 4
    run train_posmajor_b --stage train --steps 8 --warmup 4
    WD_SPLIT_NO_POSMAJOR=1 run train_plainrows_b --stage train --steps 8 --warmup 4
    ;;
  nosplitk)
    run e2e_planner_a --steps 10 --warmup 3
    WD_SPLIT_SPLITK=1 run e2e_no_kslices_a --steps 10 --warmup 3
    run e2e_planner_b --steps 10 --warmup 3
    WD_SPLIT_SPLITK=1 run e2e_no_kslices_b --steps 10 --warmup 3
    ;;
  defer)
    run e2e_defer_a --steps 10 --warmup 3
    run e2e_nodefer_a --no-defer-track --steps 10 --warmup 3
    run e2e_defer_b --steps 10 --warmup 3
    run e2e_nodefer_b --no-defer-track --steps 10 --warmup 3
    run detect_only --stage detect --steps 10 --warmup 3
    ;;
  *) run "$@" ;;
esac
