#!/bin/bash
# MFMA-pipe utilisation of the FC GEMMs (own v2 kernel vs hipBLASLt) from counters: busy cycles / (SIMDs x elapsed cycles)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmcg
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d /tmp/pmcg -- python3 $R/tools/fc_bench.py > /tmp/pmcg.log 2>&1
f=$(find /tmp/pmcg -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'gemm_nt_v2' in n or 'Cijk' in n:
        acc[n[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    gui = m.get('GRBM_GUI_ACTIVE', 0)
    print('%-62s launches %3d  MFMA_BUSY %.4g  GUI_ACTIVE %.4g  -> busy share %.3f (of 1024 SIMD x elapsed cycles)  %s' % (
        k, len(d.get('GRBM_GUI_ACTIVE', [])), m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), gui,
        m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024.0 * gui) if gui else 0, {c: '%.4g' % v for c, v in m.items() if c not in ('GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES')}))
PY
