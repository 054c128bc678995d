#!/bin/bash
# steady-state kernel breakdown of one training step: kernel trace of bench.py --stage train, last steps only
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tr -- python3 $R/bench.py --stage train --steps 6 --warmup 4 --no-cpu-baseline > /tmp/tr.json 2>/tmp/tr.log
python3 - "$(find /tmp/prof_tr -name '*kernel_trace.csv' | head -1)" > $R/gpurun_out/train_steady.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# step boundaries: the roi_pool backward kernel runs exactly 3 times per step (one per cascade stage)
marks = [i for i, r in enumerate(rows) if 'roi_pool_fpn_bwd_kernel' in r['Kernel_Name']]
steps = len(marks) // 3
first = marks[(steps - 4) * 3]            # start of the 4th-from-last step's backward ... use the last 3 full steps
last = marks[(steps - 1) * 3]
sel = rows[first:last]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    d = acc[r['Kernel_Name'][:400]]
    d[0] += 1; d[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
tot = sum(v[1] for v in acc.values())
wall = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6
print('3 steps: kernel time %.1f ms/step, wall %.1f ms/step' % (tot / 3, wall / 3))
# idle time between kernels (the host not keeping up / host synchronisations): total, and summed by the kernel that ran BEFORE the gap
end = int(sel[0]['End_Timestamp'])
gaps = collections.defaultdict(lambda: [0, 0.0])
idle = 0.0
for a, b in zip(sel, sel[1:]):
    end = max(end, int(a['End_Timestamp']))
    g = (int(b['Start_Timestamp']) - end) / 1e3
    if g > 0:
        idle += g
        key = a['Kernel_Name'][:60] + '  ->  ' + b['Kernel_Name'][:60]
        gaps[key][0] += 1; gaps[key][1] += g
print('idle between kernels: %.2f ms/step; by (kernel before -> kernel after), us/step:' % (idle / 3e3))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print('   %8.1f us/step %6.1f gaps/step  %s' % (v[1] / 3, v[0] / 3, k))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:70]:
    print('%7.2f ms/step %7.1f calls/step %8.1f us  %s' % (v[1] / 3, v[0] / 3, v[1] / v[0] * 1e3, k))
PY
cat $R/gpurun_out/train_steady.txt
