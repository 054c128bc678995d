"""ROIAlign backward at the training shape (512 sampled ROIs, 256 channels, 4 FPN levels of an 886 x 1280 crop), HIP-event timed (tools only).
WD_ROI_BWD=sample selects the per-sample kernel."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops

g = torch.Generator().manual_seed(0)
strides = [4, 8, 16, 32]
feats = [torch.zeros((1, 256, 896 // s, 1280 // s), device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True) for s in strides]
wh = torch.exp(torch.rand((512, 2), generator=g) * 3.0 + 2.5)              # 12 .. 245 px
xy = torch.rand((512, 2), generator=g) * torch.tensor([1280.0, 886.0])
boxes = torch.cat((xy - wh / 2, xy + wh / 2), 1).clamp(min=0)
boxes[:, 2].clamp_(max=1280); boxes[:, 3].clamp_(max=886)
rois = torch.cat((torch.zeros(512, 1), boxes), 1).cuda()
gout = torch.randn((512, 256, 7, 7), device='cuda').contiguous(memory_format=torch.channels_last)


def step():
    out = ops.RoiPoolFpnFn.apply(rois, [1.0 / s for s in strides], 7, 2, 4, 224.0, *feats)
    out.backward(gout)
    for f in feats:
        f.grad = None


def fwd():
    with torch.no_grad():
        ops.RoiPoolFpnFn.apply(rois, [1.0 / s for s in strides], 7, 2, 4, 224.0, *feats)


t = {}
for name, f in (('fwd', fwd), ('both', step)):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    t[name] = e0.elapsed_time(e1) / 10 * 1e3
print('ROIAlign 512 ROIs x 256 ch: forward %.1f us, backward (incl. zero-filling the four gradient maps) %.1f us' % (t['fwd'], t['both'] - t['fwd']))
