"""Summarise a rocprofv3 rocpd sqlite db: per-kernel totals (ms per frame if --frames given)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = cur.execute("select sum(end-start)/1e6 from kernels").fetchone()[0]
print('total kernel time %.2f ms  (%.2f ms/frame over %g frames)' % (tot, tot / frames, frames))
print('%10s %8s %10s  %s' % ('ms/frame', 'calls/f', 'avg_us', 'kernel'))
for name, n, ms, us in cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc limit %d" % top):
    print('%10.3f %8.1f %10.1f  %s' % (ms / frames, n / frames, us, name[:120]))
