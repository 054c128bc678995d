#!/bin/bash
# Does the SORT chunk kernel (side stream) overlap the detector's kernels?  kernel trace of the default bench -> for every
# sort_streams_kernel interval: how much detector kernel time ran inside it.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ov -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-verify > /tmp/ov.json 2>/tmp/ov.log
python3 - "$(find /tmp/prof_ov -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')) for r in rows]
iv.sort()
sorts = [v for v in iv if 'sort_streams_kernel' in v[2]]
print('%d sort kernels; queues seen: %s' % (len(sorts), sorted(set((v[3]) for v in iv))[:10]))
for s0, s1, name, q, st in sorts[-4:]:
    inside = sum(min(e, s1) - max(b, s0) for b, e, n, _, _ in iv if 'sort_streams' not in n and e > s0 and b < s1)
    print('sort %.2f ms (queue %s stream %s): detector kernel time inside its interval %.2f ms' % ((s1 - s0) / 1e6, q, st, inside / 1e6))
PY
cut -c1-200 /tmp/ov.json
