#!/bin/bash
# SQ counters of the fused deformable backward kernels (tools only): where do the wave cycles go?  Separate --pmc passes, kernel trace only.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/deform_bwd_pmc.txt
: > $OUT
export DBW_ONE=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  rm -rf /tmp/dbwpmc
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/dbwpmc -- python3 $R/tools/deform_fused_bwd_bench.py > /tmp/dbwpmc.log 2>&1
  f=$(find /tmp/dbwpmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r'(deform_d\w+<\d+>)', r['Kernel_Name'])
    if m:
        acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    print('  %-30s' % k, ' '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())), 'n=%d' % len(next(iter(d.values()))))
PY
done
cat $OUT
