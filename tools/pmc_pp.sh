#!/bin/bash
# SQ / LDS counters of the ping-pong deformable conv (tools/deform_one.py, res4); summaries -> gpurun_out/pmc_pp/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_pp
mkdir -p $OUT
export WD_DEFORM_PATCH=${WD_DEFORM_PATCH:-pp}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcpp_$i -- python3 $R/tools/deform_one.py > /tmp/pmcpp_$i.log 2>&1
  f=$(find /tmp/pmcpp_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$OUT/pass$i.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'deform_conv3x3' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
with open(sys.argv[2], 'w') as o:
    for k, d in acc.items():
        for c, v in d.items():
            o.write('%s %s n=%d mean=%.6g\n' % (k, c, len(v), sum(v) / len(v)))
PY
done
cat $OUT/pass*.txt
