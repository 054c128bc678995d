#!/bin/bash
# HBM-side traffic of the roofline kernel on the res4 shape (tools/deform_one.py) from the L2's memory-side request
# counters (FETCH_SIZE / WRITE_SIZE are derived from exactly these; the derived names hang this rocprofv3 build).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_e2e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d /tmp/prof_pmc -- python3 $R/tools/deform_one.py > /tmp/prof_pmc.log 2>&1
python3 - "$(find /tmp/prof_pmc -name '*counter_collection.csv' | head -1)" $OUT/pmc_traffic.json <<'PY'
import csv, json, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'deform_conv3x3' in n:
        key = re.search(r'deform_conv3x3\w*(<[^>]*>)?', n).group(0)
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    rd = (m['TCC_EA0_RDREQ_sum'] - m['TCC_EA0_RDREQ_32B_sum']) * 64 + m['TCC_EA0_RDREQ_32B_sum'] * 32
    wr = m['TCC_EA0_WRREQ_64B_sum'] * 64 + (m['TCC_EA0_WRREQ_sum'] - m['TCC_EA0_WRREQ_64B_sum']) * 32
    out[k] = dict(launches=len(d['TCC_EA0_RDREQ_sum']), counters=m, fetch_bytes_raw=rd,
                  fetch_bytes_gfx950_corrected=2 * rd, write_bytes=wr, traffic_bytes_per_launch=2 * rd + wr)
json.dump(out, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
print(json.dumps(out))
PY
# round 5: the split-operand GEMM on the shape the bench line's roofline names (res4 1x1: M 9600, N 1024, K 1024, residual + bias + ReLU epilogue),
# same counters, merged into the same file under the bench tag of that shape
timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d /tmp/prof_pmc_split -- python3 $R/tools/gemm_split_one.py 9600 1024 1024 6 1 > /tmp/prof_pmc_split.log 2>&1
python3 - "$(find /tmp/prof_pmc_split -name '*counter_collection.csv' | head -1)" $OUT/pmc_traffic.json <<'PY'
import csv, json, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_split_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {c: sum(v) / len(v) for c, v in acc.items()}
rd = (m['TCC_EA0_RDREQ_sum'] - m['TCC_EA0_RDREQ_32B_sum']) * 64 + m['TCC_EA0_RDREQ_32B_sum'] * 32
wr = m['TCC_EA0_WRREQ_64B_sum'] * 64 + (m['TCC_EA0_WRREQ_sum'] - m['TCC_EA0_WRREQ_64B_sum']) * 32
out = json.load(open(sys.argv[2]))
M, N, K = 9600, 1024, 1024
out['gemm_split_kernel: 1x1 conv / GEMM M=%d N=%d K=%d' % (M, N, K)] = dict(
    launches=len(acc['TCC_EA0_RDREQ_sum']), counters=m, fetch_bytes_raw=rd, fetch_bytes_gfx950_corrected=2 * rd, write_bytes=wr,
    traffic_bytes_per_launch=2 * rd + wr,
    algorithmic_bytes_per_launch=4 * (M * K + 2 * M * N) + 6 * N * K,       # A once, residual in + out once (f32), packed weight planes once (3 x bf16)
    note='with the residual + bias + ReLU epilogue, as in the backbone')
json.dump(out, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
print(json.dumps(out[list(out)[-1]]))
PY
