"""Two streams, each a deep chain of the deformable 3x3 path of a res4 block (offset conv GEMM + table launch + persistent kernel), new data per
iteration: equal to serial?  (No split-operand kernel involved.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
from waymo_2d_tracking_amd.detnet.nn import ops, cascade_rcnn
torch.backends.cudnn.deterministic = True
NIT = int(os.environ.get('DIAG_ITERS', '8'))
REPS = int(os.environ.get('CHAIN_REPS', '30'))
ms = [Detectron2Det(seed=5).cuda().eval(), Detectron2Det(seed=6).cuda().eval()]
g = torch.Generator().manual_seed(0)
H, W = int(os.environ.get('CHAIN_H', '16')), int(os.environ.get('CHAIN_W', '24'))
xs = [[(torch.randn(1, 1024, H, W, generator=g) * 0.5).cuda().contiguous(memory_format=torch.channels_last) for _ in range(2)] for _ in range(NIT)]


def chain(m, x):
    with torch.no_grad():
        for b in range(1, 1 + REPS):
            blk = m.model.backbone.res4[b]
            offset, table = blk.offset_conv(x, deform_table=True)
            x = ops.deform_conv3x3(x, offset, blk.packed_weight(), cascade_rcnn.GROUPS, 1, 1, blk.conv2_scale, blk.conv2_bias, relu=False, table=table) * 0.5
    return x


chain(ms[0], xs[0][0]); chain(ms[1], xs[0][1])
refs = []
for it in range(NIT):
    refs.append([chain(ms[i], xs[it][i]) for i in range(2)])
    torch.cuda.synchronize()
s = [torch.cuda.Stream(), torch.cuda.Stream()]
bad = 0
for it in range(NIT):
    got = [None, None]
    for i in range(2):
        with torch.cuda.stream(s[i]):
            got[i] = chain(ms[i], xs[it][i])
    torch.cuda.synchronize()
    for i in range(2):
        if not torch.equal(refs[it][i], got[i]):
            bad += 1
            print('iteration %d stream %d differs: max |d| %.3e' % (it, i, float((refs[it][i] - got[i]).abs().max())))
print('deform chains %dx%d mismatches: %d' % (H, W, bad))
