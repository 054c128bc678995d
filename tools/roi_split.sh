#!/bin/bash
# per-kernel split of one wd_roi_pool_fpn_f32 call (row kernel / ordering / fallback), both processing orders -> stdout
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for o in 0 1; do
  WD_ROI_ORDER=$o rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/roi_$o -- python3 $R/tools/roi_bench.py > /tmp/roi_$o.log 2>&1
  echo "WD_ROI_ORDER=$o: $(grep roi_pool_fpn /tmp/roi_$o.log | tail -1)"
  python3 - "$(find /tmp/roi_$o -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'roi' in r['Name']:
        print('   %-28s calls %4s  avg %8.1f us  total %9.1f us' % (r['Name'].split('::')[-1][:28], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
PY
done
