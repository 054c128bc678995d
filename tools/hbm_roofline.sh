#!/bin/bash
# Counter-backed HBM rooflines of the bandwidth-bound detector kernels: tools/hbm_roofline.py un-profiled (HIP-event times,
# algorithmic bytes) + two rocprofv3 --pmc passes (reads / writes at the L2's memory side; raw TCC_EA0_* counters - the derived
# FETCH_SIZE / WRITE_SIZE names hang this rocprofv3 build).  gfx950: FETCH bytes of wide coalesced reads are reported at 1/2
# (MI355X_MICROARCH.md, HBM section) -> x2 correction on reads.  Output: gpurun_out/hbm_roofline/summary.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/hbm_roofline
mkdir -p $OUT
python3 $R/tools/hbm_roofline.py > $OUT/timing.json 2> $OUT/timing.err
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d /tmp/hbm_rd -- python3 $R/tools/hbm_roofline.py > /tmp/hbm_rd.log 2>&1
timeout 600 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d /tmp/hbm_wr -- python3 $R/tools/hbm_roofline.py > /tmp/hbm_wr.log 2>&1
python3 - "$(find /tmp/hbm_rd -name '*counter_collection.csv' | head -1)" "$(find /tmp/hbm_wr -name '*counter_collection.csv' | head -1)" $OUT/timing.json $OUT/summary.json <<'PY'
import csv, json, sys, collections
def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name']
        for key in ('roi_pool_wg_kernel', 'roi_bucket_order_kernel', 'roi_pool_row_kernel', 'roi_pool_sep_kernel', 'nms_mask_kernel', 'nms_sweep_col_kernel', 'preprocess_kernel'):
            if key in n:
                acc[key + ' grid=' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} | {'launches': len(next(iter(d.values())))} for k, d in acc.items()}
rd, wr = load(sys.argv[1]), load(sys.argv[2])
timing = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
out = {'timing_and_algorithmic_bytes': timing, 'kernels': {}}
for k in sorted(set(rd) | set(wr)):
    r, w = rd.get(k, {}), wr.get(k, {})
    fetch_raw = (r.get('TCC_EA0_RDREQ_sum', 0) - r.get('TCC_EA0_RDREQ_32B_sum', 0)) * 64 + r.get('TCC_EA0_RDREQ_32B_sum', 0) * 32
    write = w.get('TCC_EA0_WRREQ_64B_sum', 0) * 64 + (w.get('TCC_EA0_WRREQ_sum', 0) - w.get('TCC_EA0_WRREQ_64B_sum', 0)) * 32
    out['kernels'][k] = dict(launches=r.get('launches', w.get('launches')), fetch_bytes_raw=fetch_raw,
                             fetch_bytes_gfx950_corrected=2 * fetch_raw, write_bytes=write,
                             traffic_bytes_per_launch=2 * fetch_raw + write)
json.dump(out, open(sys.argv[4], 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
