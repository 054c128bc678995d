#!/bin/bash
# build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/probe/lds_bank_probe tools/probe/lds_bank_probe.hip
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmcprobe
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcprobe -- $R/tools/probe/lds_bank_probe > /tmp/pmcprobe.log 2>&1
grep "waves" /tmp/pmcprobe.log > /tmp/names.txt
f=$(find /tmp/pmcprobe -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
names = [l.strip() for l in open('/tmp/names.txt')]
for (d, c), n in zip(by.items(), names):
    print('%-90s insts %.0f active %.0f conflict %.0f  (%.2f cycles/inst)' % (n[:90], c.get('SQ_INSTS_LDS', 0), c.get('SQ_LDS_IDX_ACTIVE', 0), c.get('SQ_LDS_BANK_CONFLICT', 0), c.get('SQ_LDS_IDX_ACTIVE', 0) / max(1, c.get('SQ_INSTS_LDS', 1))))
PY
