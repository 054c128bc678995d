"""Microbench of wd_roi_pool_fpn_f32: 1000 FPN-consistent ROIs on 1920x1280 feature pyramids (C=256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
g = torch.Generator().manual_seed(0)
strides = [4, 8, 16, 32]
feats = [torch.randn(1, 256, 1280 // s, 1920 // s, device='cuda').contiguous(memory_format=torch.channels_last) for s in strides]
n = 1000
size = torch.exp(torch.empty(n).uniform_(3.0, 6.5, generator=g))          # sqrt(area) 20 .. 665 px
ar = torch.exp(torch.empty(n).uniform_(-0.7, 0.7, generator=g))
w, h = size * ar.sqrt(), size / ar.sqrt()
cx = torch.empty(n).uniform_(0, 1920, generator=g); cy = torch.empty(n).uniform_(0, 1280, generator=g)
rois = torch.stack([torch.zeros(n), (cx - w / 2).clamp(0, 1920), (cy - h / 2).clamp(0, 1280), (cx + w / 2).clamp(0, 1920), (cy + h / 2).clamp(0, 1280)], 1).cuda()
f = lambda: ops.roi_pool_fpn(feats, rois, [1.0 / s for s in strides])
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
# algorithmic bytes: unique footprint (ceil(w_l)+1)(ceil(h_l)+1)*C*4 + 20 + 49*C*4 per ROI
lvl = torch.floor(4 + torch.log2(torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])) / 224 + 1e-8)).clamp(2, 5)
sc = 1.0 / (2.0 ** lvl)
fw = torch.ceil((rois[:, 3] - rois[:, 1]) * sc) + 1; fh = torch.ceil((rois[:, 4] - rois[:, 2]) * sc) + 1
alg = float((fw * fh * 256 * 4 + 20 + 49 * 256 * 4).sum())
print('roi_pool_fpn 1000 rois: %.1f us, algorithmic %.1f MB -> %.0f GB/s (%.1f%% of 8 TB/s)' % (us, alg / 1e6, alg / us / 1e3, alg / us / 1e3 / 80))
