"""Like diag_two_pipelines.py, with per-frame checksums of intermediates (feature maps, proposals, stage logits) to localise a mismatch."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline
kw = dict(n_cameras=2, frames_per_camera=2, height=256, width=384, segment_frames=8, distinct_times=4, use_graph=False, deterministic=True)
LEVEL = os.environ.get('DIAG_LEVEL', 'all')


def instrument(p, log):
    core = p.model.model
    orig = core.forward_padded_from

    def wrapped(feats, img_h, img_w, proposals=None, intermediates=None):
        inter = {}
        out = orig(feats, img_h, img_w, proposals, inter)
        rec = [f.double().sum() for f in feats] + [f.double().sum() for f in inter['feats']] + [inter['proposals'].double().sum()]
        for lg, dl in inter['stage_out']:
            rec += [lg.double().sum(), dl.double().sum()]
        rec += [inter['boxes'].double().sum(), inter['scores'].double().sum()]
        log.append(torch.stack(rec))
        return out
    core.forward_padded_from = wrapped


names = ['c2', 'c3', 'c4', 'c5', 'p2', 'p3', 'p4', 'p5', 'p6', 'proposals', 'logits0', 'deltas0', 'logits1', 'deltas1', 'logits2', 'deltas2', 'boxes', 'scores']


def serial(seed):
    a = DetectTrackPipeline(seed=seed, **kw)
    log = []
    instrument(a, log)
    for _ in range(2):
        a.step(True); torch.cuda.synchronize()
    return torch.stack(log).cpu()


r1 = serial(5)
a2, b2 = DetectTrackPipeline(seed=5, **kw), DetectTrackPipeline(seed=6, **kw)
la, lb = [], []
instrument(a2, la); instrument(b2, lb)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for _ in range(2):
    with torch.cuda.stream(sa):
        a2.step(True)
    with torch.cuda.stream(sb):
        b2.step(True)
torch.cuda.synchronize()
g = torch.stack(la).cpu()
print('frames x checksums equal:', bool(torch.equal(r1, g)))
for f in range(r1.shape[0]):
    bad = [names[i] for i in range(r1.shape[1]) if r1[f, i] != g[f, i]]
    if bad:
        print('  frame %d differs in: %s' % (f, bad))
