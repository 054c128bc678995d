"""Per-kernel time per frame: e2e run vs detector-only run (dumps of tools/sort_overlap_trace.sh)."""
import json, sys
a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, [0, 0]); cb, tb = b.get(k, [0, 0])
    rows.append((ta - tb, k, ca, ta, cb, tb))
rows.sort(reverse=True)
print('%-100s %8s %10s %8s %10s %9s' % ('kernel', 'calls/f', 'us/f e2e', 'calls/f', 'us/f det', 'diff us/f'))
for d, k, ca, ta, cb, tb in rows[:22]:
    print('%-100s %8.1f %10.1f %8.1f %10.1f %9.1f' % (k, ca, ta, cb, tb, d))
print('... total e2e %.1f us/frame, detect %.1f us/frame, diff %.1f' % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values()), sum(r[0] for r in rows)))
