#!/bin/bash
# Per-kernel times of the GPU JPEG decoder (rocprofv3 --kernel-trace --stats of tools/jpeg_bench.py) -> gpurun_out/jpeg/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/jpeg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_jpeg -- python3 $R/tools/jpeg_bench.py ${1:-20} > $OUT/bench.txt 2> /tmp/prof_jpeg.log
cp $(find /tmp/prof_jpeg -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find /tmp/prof_jpeg -name "*kernel_trace.csv" | head -1) /tmp/jpeg_trace.csv
python3 - <<'PY' > $OUT/per_image.txt
import csv, collections
rows = list(csv.DictReader(open('/tmp/jpeg_trace.csv')))
rows = [r for r in rows if 'jpeg' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# images = groups ending with a color kernel
img, cur = [], []
for r in rows:
    cur.append(r)
    if 'color' in r['Kernel_Name']:
        img.append(cur); cur = []
def short(n): return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
for k, idx in (('first image set (1920x1280 q90)', 5), ('second (1920x886 q90)', 30), ('third (q100 noise)', 55)):
    if idx >= len(img): continue
    g = img[idx]
    t0, t1 = int(g[0]['Start_Timestamp']), int(g[-1]['End_Timestamp'])
    print('%s: %d launches, first start -> last end %.1f us' % (k, len(g), (t1 - t0) / 1e3))
    for r in g:
        print('    %8.1f us  grid %-8s %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''), short(r['Kernel_Name'])))
PY
cat $OUT/bench.txt | grep -v amdgpu; cat $OUT/per_image.txt
