#!/bin/bash
# Which detector kernels slow down while the SORT chunk kernel is resident?  Kernel trace of the default bench; every detector kernel
# launch is classified by whether it overlaps a sort_streams_kernel interval; per kernel name: mean duration inside / outside.
# usage: bash tools/sort_overlap_trace.sh [extra bench flags]  -> gpurun_out/sort_overlap_trace.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sot
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sot -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > /tmp/sot.json 2>/tmp/sot.log
python3 - "$(find /tmp/prof_sot -name '*kernel_trace.csv' | head -1)" > $R/gpurun_out/sort_overlap_trace.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
sorts = [(r['s'], r['e']) for r in rows if 'sort_streams_kernel' in r['Kernel_Name']]
sorts = sorts[2:]                                       # steady state
if not sorts:
    print('no sort kernel')
    sorts = []
    t_lo = rows[len(rows) // 2]['s']
else:
    t_lo = sorts[0][0] - 400_000_000
print('%d SORT chunk kernels, mean %.2f ms' % (len(sorts), sum(e - s for s, e in sorts) / max(1, len(sorts)) / 1e6))
def inside(r):
    return any(r['s'] < e and r['e'] > s for s, e in sorts)
acc = collections.defaultdict(lambda: [[0, 0.0], [0, 0.0]])
tot_in = tot_in_expected = 0.0
for r in rows:
    if r['s'] < t_lo or 'sort_streams' in r['Kernel_Name']:
        continue
    k = r['Kernel_Name'][:90]
    a = acc[k][1 if inside(r) else 0]
    a[0] += 1; a[1] += (r['e'] - r['s']) / 1e3
print('%-92s %8s %10s %8s %10s %8s' % ('kernel', 'n out', 'us out', 'n in', 'us in', 'extra ms'))
out = []
for k, (o, i) in acc.items():
    if o[0] and i[0]:
        mo, mi = o[1] / o[0], i[1] / i[0]
        out.append(((mi - mo) * i[0] / 1e3, k, o[0], mo, i[0], mi))
out.sort(reverse=True)
total = sum(x[0] for x in out)
for extra, k, no, mo, ni, mi in out[:25]:
    print('%-92s %8d %10.1f %8d %10.1f %8.3f' % (k, no, mo, ni, mi, extra))
# idle time of the detector's own kernels (everything except the tracker's) over the steady-state window
TRK = ('sort_streams', 'rank_classes', 'gather_ranked', 'scan2_kernel', 'finalize_kernel', 'resolve_births', 'advance_streams', 'totals_kernel', 'stream_prefix', 'global_ids')
det = [r for r in rows if r['s'] >= t_lo and not any(t in r['Kernel_Name'] for t in TRK)]
marks = [r['s'] for r in det if 'preprocess_kernel' in r['Kernel_Name']]
if len(marks) > 12:
    w0, w1 = marks[1], marks[-1]
    nfr = len(marks) - 2
    sel = [r for r in det if w0 <= r['s'] < w1]
    busy, cur_s, cur_e = 0, None, None
    for r in sel:
        if cur_e is None or r['s'] > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = r['s'], r['e']
        else:
            cur_e = max(cur_e, r['e'])
    busy += cur_e - cur_s
    gaps = sorted(((b['s'] - a['e']) / 1e3, a['Kernel_Name'][:50], b['Kernel_Name'][:50]) for a, b in zip(sel, sel[1:]) if b['s'] > a['e'])
    print('detector kernels over %d frames: wall %.3f ms/frame, busy %.3f ms/frame, idle %.3f ms/frame; largest gaps (us):' % (nfr, (w1 - w0) / 1e6 / nfr, busy / 1e6 / nfr, (w1 - w0 - busy) / 1e6 / nfr))
    for g in gaps[-6:]:
        print('   %8.1f us after %s before %s' % g)
import json, os
if len(marks) > 12:
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in sel:
        d = per[r['Kernel_Name'][:100]]
        d[0] += 1; d[1] += (r['e'] - r['s']) / 1e3
    json.dump({k: [v[0] / nfr, v[1] / nfr] for k, v in per.items()}, open(os.environ.get('SOT_DUMP', '/tmp/sot_dump.json'), 'w'))
print('sum over all kernels of (mean inside - mean outside) x launches inside: %.2f ms over %d SORT intervals = %.2f ms per chunk' % (total, len(sorts), total / max(1, len(sorts))))
PY
cat $R/gpurun_out/sort_overlap_trace.txt; tail -1 /tmp/sot.json | cut -c1-140
