#!/bin/bash
# Regenerate waymo_2d_tracking_amd/tuning/tunableop_gfx950.csv on an MI355X: run every bench stage with online TunableOp tuning
# and merge the per-process result files (validator header once, one line per GEMM signature, fastest entry kept).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mv waymo_2d_tracking_amd/tuning/tunableop_gfx950.csv /tmp/old_tunableop.csv 2>/dev/null
i=0
for args in "--steps 2 --warmup 2" "--stage train --steps 2 --warmup 3" "--stage detect --tta x1.5,hflip --steps 2 --warmup 2" ; do
  i=$((i+1))
  WT_GEMM_TUNING_ONLINE=1 WT_TUNABLEOP_OUT=/tmp/wt_tune_$i.csv python3 bench.py $args --no-cpu-baseline | tail -1 | cut -c1-160
done
python3 - <<'PY'
import glob
head, best = [], {}
for f in sorted(glob.glob('/tmp/wt_tune_*.csv')):
    for line in open(f):
        line = line.strip()
        if not line:
            continue
        if line.startswith('Validator'):
            if line not in head:
                head.append(line)
            continue
        op, sig, sol, t = line.split(',')
        if (op, sig) not in best or float(t) < float(best[(op, sig)][1]):
            best[(op, sig)] = (sol, t)
with open('gpurun_out/tunableop_gfx950.csv', 'w') as o:
    o.write('\n'.join(head) + '\n')
    for (op, sig), (sol, t) in sorted(best.items()):
        o.write('%s,%s,%s,%s\n' % (op, sig, sol, t))
print('merged', len(best), 'GEMM signatures')
PY
