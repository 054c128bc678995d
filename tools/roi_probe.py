"""What bounds roi_pool_row_kernel?  (tools only)  Same 1000-ROI call with (a) the benchmark's random ROIs, (b) 1000 copies of ONE ROI
(every feature byte after the first touch is a cache hit), (c) random positions but all on one level / one size."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops

g = torch.Generator().manual_seed(0)
strides = [4, 8, 16, 32]
feats = [torch.randn(1, 256, 1280 // s, 1920 // s, device='cuda').contiguous(memory_format=torch.channels_last) for s in strides]
n = 1000


def rois_random():
    size = torch.exp(torch.empty(n).uniform_(3.0, 6.5, generator=g))
    ar = torch.exp(torch.empty(n).uniform_(-0.7, 0.7, generator=g))
    w, h = size * ar.sqrt(), size / ar.sqrt()
    cx = torch.empty(n).uniform_(0, 1920, generator=g)
    cy = torch.empty(n).uniform_(0, 1280, generator=g)
    return torch.stack([torch.zeros(n), (cx - w / 2).clamp(0, 1920), (cy - h / 2).clamp(0, 1280), (cx + w / 2).clamp(0, 1920), (cy + h / 2).clamp(0, 1280)], 1)


def bench(rois, label):
    rois = rois.cuda()
    f = lambda: ops.roi_pool_fpn(feats, rois, [1.0 / s for s in strides])
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    print('%-58s %.1f us' % (label, e0.elapsed_time(e1) / 20 * 1e3), flush=True)


r = rois_random()
bench(r, 'a) random ROIs (the benchmark set)')
bench(r[:1].repeat(n, 1), 'b) 1000 copies of one ROI (all cache hits)')
one = torch.tensor([[0.0, 400.0, 300.0, 560.0, 460.0]])
bench(one.repeat(n, 1), 'c) 1000 copies of a 160 x 160 box (level 3, 20 x 20 px footprint)')
cx = torch.empty(n).uniform_(100, 1800, generator=g)
cy = torch.empty(n).uniform_(100, 1100, generator=g)
bench(torch.stack([torch.zeros(n), cx - 80, cy - 80, cx + 80, cy + 80], 1), 'd) 160 x 160 boxes at random positions')
bench(torch.stack([torch.zeros(n), cx - 20, cy - 20, cx + 20, cy + 20], 1), 'e) 40 x 40 boxes at random positions (level 2, 10 x 10 px)')
