"""A few res4 identity bottleneck blocks (deformable, 1024 channels, 16 x 24 map) of two models on two streams vs serial, new data per iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
torch.backends.cudnn.deterministic = True
NIT = int(os.environ.get('DIAG_ITERS', '15'))
NB = int(os.environ.get('DIAG_BLOCKS', '6'))
ms = [Detectron2Det(seed=5).cuda().eval(), Detectron2Det(seed=6).cuda().eval()]
g = torch.Generator().manual_seed(0)
xs = [[(torch.randn(1, 1024, 16, 24, generator=g) * 0.5).cuda().contiguous(memory_format=torch.channels_last) for _ in range(2)] for _ in range(NIT)]


def run(m, x):
    outs = []
    with torch.no_grad():
        x = x.clone(memory_format=torch.preserve_format)
        for b in range(1, 1 + NB):
            x = m.model.backbone.res4[b](x)
            outs.append(x.clone())
    return outs


run(ms[0], xs[0][0]); run(ms[1], xs[0][1])
refs = []
for it in range(NIT):
    refs.append([run(ms[i], xs[it][i]) for i in range(2)])
    torch.cuda.synchronize()
s = [torch.cuda.Stream(), torch.cuda.Stream()]
first = {}
for it in range(NIT):
    got = [None, None]
    for i in range(2):
        with torch.cuda.stream(s[i]):
            got[i] = run(ms[i], xs[it][i])
    torch.cuda.synchronize()
    for i in range(2):
        for b, (x, y) in enumerate(zip(refs[it][i], got[i])):
            if not torch.equal(x, y):
                first[(it, i)] = (b, float((x - y).abs().max()))
                break
print('first differing block per (iteration, stream):', first)
