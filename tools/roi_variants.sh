#!/bin/bash
# ROIAlign kernels side by side (tools only).  usage: roi_variants.sh "name|lib-or-empty|KERNEL|ORDER" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/roi_variants.txt
: > $OUT
for v in "$@"; do
  IFS='|' read name lib kern ord <<< "$v"
  echo "== $name (lib=$lib WD_ROI_KERNEL=$kern WD_ROI_ORDER=$ord)" >> $OUT
  export WD_ROI_KERNEL=$kern WD_ROI_ORDER=$ord
  if [ -n "$lib" ]; then export WT_LIB_PATH=$R/waymo_2d_tracking_amd/csrc/variants/$lib; else unset WT_LIB_PATH; fi
  python3 $R/tools/roi_probe.py 2>&1 | grep -v amdgpu.ids >> $OUT
  SETS=6 python3 - 2>&1 <<PY | grep -v amdgpu.ids >> $OUT
import json, sys
src = open('$R/tools/hbm_roofline.py').read().split("del pyramids")[0]
exec(src)
k = list(out.values())[0]
print('cold (6 pyramids): %.1f us  frac %.3f' % (k['us'], k['frac_of_8TBs']))
PY
done
cat $OUT
