#!/bin/bash
# steady-state kernel breakdown of one frame of the bench (kernel trace, last frames only).  INFLIGHT=1 (default here): the anatomy of ONE frame, kernel
# durations not stretched by a second frame sharing the chip; INFLIGHT=2: the default bench - kernel time per frame then sums OVERLAPPING kernels
# (kernel time / wall = the average number of kernels in flight)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export WT_BENCH_NO_EXACT=1   # the trace must end with the headline pipeline, not the exact-f32 secondary run
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_e2e -- python3 $R/bench.py --inflight ${INFLIGHT:-1} --steps 3 --warmup 2 --no-cpu-baseline > /tmp/e2e.json 2>/tmp/e2e.log
python3 - "$(find /tmp/prof_e2e -name '*kernel_trace.csv' | head -1)" $R/gpurun_out/e2e_frame_sequence.txt > $R/gpurun_out/e2e_steady.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'preprocess_kernel' in r['Kernel_Name']]
nf = 20
sel = rows[marks[-nf - 1]:marks[-1]]
acc = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for r in sel:
    d = acc[r['Kernel_Name'][:110]]
    dt = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    d[0] += 1; d[1] += dt
tot = sum(v[1] for v in acc.values())
wall = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6
gaps = sorted(((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3, a['Kernel_Name'][:60], b['Kernel_Name'][:60]) for a, b in zip(sel, sel[1:]))
print('%d frames: kernel time %.2f ms/frame, wall %.2f ms/frame, %d launches/frame' % (nf, tot / nf, wall / nf, len(sel) / nf))
print('idle between kernels: %.2f ms/frame; largest gaps (us):' % (sum(max(g[0], 0) for g in gaps) / 1e3 / nf))
for g in gaps[-8:]:
    print('   %8.1f us after %s before %s' % g)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:60]:
    print('%7.3f ms/f %7.1f calls/f %8.1f us  %s' % (v[1] / nf, v[0] / nf, v[1] / v[0] * 1e3, k))
# launch sequence of the last full frame (name, grid, us)
with open(sys.argv[2], 'w') as fp:
    one = rows[marks[-2]:marks[-1]]
    t0 = int(one[0]['Start_Timestamp'])
    for r in one:
        fp.write('%9.1f %8.1f us  grid %-10s wg %-5s %s\n' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                 r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')), r['Kernel_Name'][:100]))
PY
cat $R/gpurun_out/e2e_steady.txt
