#!/bin/bash
# Round profile of the default bench command: rocprofv3 kernel stats + a separate PMC pass (HBM traffic of the roofline
# kernel).  Run on the GPU box; results land in gpurun_out/prof_e2e/ and are copied to profiles/ by hand.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_e2e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /tmp/prof_stats.log
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/prof_pmc.json 2> /tmp/prof_pmc.log
python3 - "$(find /tmp/prof_pmc -name '*counter_collection.csv' | head -1)" $OUT/pmc_traffic.json <<'PY'
import csv, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'deform_conv3x3' in n or 'roi_pool' in n or 'gemm_nt' in n:
        key = n.split('(')[0].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
        acc[key + ' grid=' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in acc.items():
    out[k] = {c: dict(launches=len(v), mean=sum(v) / len(v)) for c, v in d.items()}
json.dump(out, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
PY
cd $R && python3 bench.py --steps 5 --warmup 2 > $OUT/bench_line.json 2> /dev/null
tail -c 600 $OUT/bench_line.json
