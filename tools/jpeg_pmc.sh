#!/bin/bash
# SQ / LDS counters of the JPEG entropy kernels (tools/jpeg_bench.py) -> gpurun_out/jpeg/pmc_pass*.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/jpeg
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcj_$i -- python3 $R/tools/jpeg_bench.py 4 > /tmp/pmcj_$i.log 2>&1
  f=$(find /tmp/pmcj_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$OUT/pmc_pass$i.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
seen = collections.Counter()
for r in rows:
    if 'jpeg_' in r['Kernel_Name']:
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
with open(sys.argv[2], 'w') as o:
    for k, d in acc.items():
        for c, v in d.items():
            o.write('%-28s %-24s n=%d first=%.6g mean=%.6g max=%.6g\n' % (k, c, len(v), v[0], sum(v) / len(v), max(v)))
PY
done
cat $OUT/pmc_pass*.txt
