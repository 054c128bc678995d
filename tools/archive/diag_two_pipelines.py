import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline
kw = dict(n_cameras=2, frames_per_camera=2, height=256, width=384, segment_frames=8, distinct_times=4, use_graph=False, deterministic=True)
def snapshot(p):
    out = [p.category[:2].clone(), p.xywhs[:2].clone()]
    for c in range(2):
        k = int(p.chunk_counts[c, 0])
        out += [p.out_bbox[c][:k].clone(), p.out_id[c][:k].clone()]
    return out
def serial(seed):
    a = DetectTrackPipeline(seed=seed, **kw)
    for _ in range(2):
        a.step(True); torch.cuda.synchronize()
    return snapshot(a)
r1 = serial(5); r2 = serial(5)
print('serial vs serial equal:', all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(r1, r2)))
for x, y in zip(r1, r2):
    if x.shape != y.shape or not torch.equal(x, y):
        print('  diff', x.shape, y.shape, (x.double() - y.double()).abs().max() if x.shape == y.shape else None)
a2, b2 = DetectTrackPipeline(seed=5, **kw), DetectTrackPipeline(seed=6, **kw)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for _ in range(2):
    with torch.cuda.stream(sa):
        a2.step(True)
    with torch.cuda.stream(sb):
        b2.step(True)
torch.cuda.synchronize()
g = snapshot(a2)
print('concurrent vs serial equal:', all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(r1, g)))
d = (r1[1] != g[1]).nonzero()
print('xywhs entries that differ: %d of %d; by (chunk, frame slot block): %s' % (len(d), r1[1].numel(), sorted(set((int(c), int(i) // 100) for c, _, i in d.tolist()))))
for c, comp, i in d[:12].tolist():
    print('   chunk %d comp %d slot %d: serial %.6f concurrent %.6f' % (c, comp, i, float(r1[1][c, comp, i]), float(g[1][c, comp, i])))
for x, y in zip(r1, g):
    if x.shape != y.shape or not torch.equal(x, y):
        print('  diff', x.shape, y.shape, (x.double() - y.double()).abs().max() if x.shape == y.shape else None)
