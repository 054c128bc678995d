#!/bin/bash
# per-kernel times of the deformable backward at the res4 training shape (tools/deform_bwd_bench.py under rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dbw
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dbw -- python3 $R/tools/${DBW_SCRIPT:-deform_bwd_bench.py} > /tmp/dbw.log 2>&1
head -12 /tmp/dbw.log | grep -v amdgpu.ids
python3 - "$(find /tmp/prof_dbw -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%9.1f us avg %6s calls %6.1f%%  %s' % (float(r['AverageNs']) / 1e3, r['Calls'], float(r['Percentage']), r['Name'][:110]))
PY
