"""Two detectors on two streams at the same time vs the same two run one after the other: first intermediate tensor that differs."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
from waymo_2d_tracking_amd.detnet.nn import ops

torch.backends.cudnn.deterministic = True
ms = [Detectron2Det(seed=5).cuda().eval(), Detectron2Det(seed=6).cuda().eval()]
g = torch.Generator().manual_seed(0)
NIT = int(os.environ.get("DIAG_ITERS", "10"))
# a different image every iteration: a consumer that reads a stale buffer must not find the previous iteration's identical values there
imgs_it = [[torch.randint(0, 256, (1, 256, 384, 3), generator=g, dtype=torch.uint8).cuda() for _ in range(2)] for _ in range(NIT)]
imgs = imgs_it[0]


def run(m, img):
    inter = {}
    with torch.no_grad():
        xn, (ho, wo) = ops.preprocess(img, 1.0, False, False, True, (103.530, 116.280, 123.675), (57.375, 57.120, 58.395), 32)
        out = m.model.forward_padded(xn, ho, wo, None, inter)
    flat = {'out_boxes': out[0], 'out_scores': out[1]}
    for i, f in enumerate(inter['feats']):
        flat['feat%d' % i] = f
    flat['proposals'] = inter['proposals']
    for k, (lg, dl) in enumerate(inter['stage_out']):
        flat['logits%d' % k] = lg
        flat['deltas%d' % k] = dl
    return {k: v.clone() for k, v in flat.items()}


run(ms[0], imgs[0]); run(ms[1], imgs[1])
refs = []
for it in range(NIT):
    refs.append([run(ms[0], imgs_it[it][0]), run(ms[1], imgs_it[it][1])])
    torch.cuda.synchronize()
s = [torch.cuda.Stream(), torch.cuda.Stream()]
for it in range(NIT):
    got = [None, None]
    ref = refs[it]
    for i in range(2):
        with torch.cuda.stream(s[i]):
            got[i] = run(ms[i], imgs_it[it][i])
    torch.cuda.synchronize()
    for i in range(2):
        for k in ref[i]:
            if not torch.equal(ref[i][k], got[i][k]):
                print('iteration %d model %d: %s differs, max |d| %.3e' % (it, i, k, float((ref[i][k].double() - got[i][k].double()).abs().max())))
print('done')
