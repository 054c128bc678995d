#!/bin/bash
# LDS conflict counters of pp-kernel build variants (csrc/variants/lib_<name>.so); experiments only
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export WD_DEFORM_PATCH=pp
for v in base "$@"; do
  if [ $v = base ]; then unset WT_LIB_PATH; else export WT_LIB_PATH=$R/waymo_2d_tracking_amd/csrc/variants/lib_$v.so; fi
  rm -rf /tmp/pv_$v
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/pv_$v -- python3 $R/tools/deform_one.py > /tmp/pv_$v.log 2>&1
  f=$(find /tmp/pv_$v -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if 'deform_conv3x3_pp' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], '  '.join('%s=%.4g' % (k.replace('SQ_', ''), sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
done
