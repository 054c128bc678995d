#!/bin/bash
# memory-side counters of the ROIAlign kernels on the cold roofline set (tools only).  usage: roi_pmc.sh "name|lib-or-empty|KERNEL|ORDER" ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/roi_pmc.txt
: > $OUT
cat > /tmp/roi_only.py <<PY
import json, sys
sys.path.insert(0, '$R')
src = open('$R/tools/hbm_roofline.py').read().split("del pyramids")[0]
exec(src)
k = list(out.values())[0]
print('cold (6 pyramids): %.1f us  frac %.3f' % (k['us'], k['frac_of_8TBs']))
PY
for v in "$@"; do
  IFS='|' read name lib kern ord <<< "$v"
  export WD_ROI_KERNEL=$kern WD_ROI_ORDER=$ord
  if [ -n "$lib" ]; then export WT_LIB_PATH=$R/waymo_2d_tracking_amd/csrc/variants/$lib; else unset WT_LIB_PATH; fi
  echo "== $name" >> $OUT
  i=0
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
             "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES"; do
    i=$((i+1))
    rm -rf /tmp/roipmc
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/roipmc -- python3 /tmp/roi_only.py > /tmp/roipmc.log 2>&1
    f=$(find /tmp/roipmc -name "*counter_collection.csv" | head -1)
    python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'roi_' in n:
        acc[n.split('(')[0][-28:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print('  ', k, ' '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())), 'n=%d' % len(next(iter(d.values()))))
PY
    grep "cold" /tmp/roipmc.log >> $OUT || tail -5 /tmp/roipmc.log >> $OUT
  done
done
cat $OUT
