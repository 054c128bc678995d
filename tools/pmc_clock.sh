#!/bin/bash
# effective shader clock under a kernel = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmcclk
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmcclk -- python3 $R/tools/deform_one.py > /tmp/pmcclk.log 2>&1
python3 - "$(find /tmp/pmcclk -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmcclk -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if 'deform_conv3x3_pp' in r['Kernel_Name']:
        cnt[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    if 'deform_conv3x3_pp' in r['Kernel_Name']:
        dur[r.get('Dispatch_Id', r.get('Correlation_Id'))] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for d, c in cnt.items():
    us = dur.get(d)
    if us:
        gui = c.get('GRBM_GUI_ACTIVE', 0) / 8
        print('dispatch %s: %.1f us, %.0f cycles -> %.2f GHz, MFMA busy %.0f cycles per SIMD = %.2f of elapsed' % (
            d, us, gui, gui / us / 1e3, c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024, c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / gui))
PY
