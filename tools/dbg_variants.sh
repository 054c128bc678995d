#!/bin/bash
# build det_deform.hip with -DWD_DBG=n (variants of the LDS kernel) and time res3/res4
for d in ${VARIANTS:-0 5 7}; do
  touch waymo_2d_tracking_amd/csrc/det_deform.hip
  WD_HIPCC_FLAGS=-DWD_DBG=$d python -m waymo_2d_tracking_amd.build > /dev/null 2>&1
  echo "== WD_DBG=$d"
  WD_DEFORM_PATCH=lds python -m pytest tests/test_gpu_detops.py -q -m gpu -k "deform" 2>&1 | tail -1
  WD_DEFORM_PATCH=lds OFF_STD=0.3 python tools/deform_bench.py | grep "res3 \|res4 "
done
