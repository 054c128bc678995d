"""HBM-bound kernels (ROIAlign, box NMS mask, fused pre-processing) with a working set the 256 MiB Infinity Cache cannot hold:
every launch reads a different copy of its input (SETS copies, > 1 GiB in total), so the memory-side counters collected over this
script (tools/hbm_roofline.sh: separate rocprofv3 --pmc passes, TCC_EA0_* raw counters) are HBM traffic, not L3 hits.

Prints one JSON object: per kernel the HIP-event time per launch and the ALGORITHMIC bytes per launch
  roi_pool   : UNIQUE footprint - the union over the 1000 ROIs of the feature pixels their bilinear samples touch, per level
               (each pixel counted once however many ROIs overlap it) x C x 4 + rois 20 B + output 49 x C x 4 per ROI
  nms_mask   : 16 n (boxes) in + n^2 / 8 (bit mask) out
  preprocess : 3 H W (uint8) in + 3 Hp Wp x 4 out
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
from waymo_2d_tracking_amd.detnet.nn.cascade_rcnn import PIXEL_MEAN, PIXEL_STD

SETS = int(os.environ.get('SETS', 6))
REPS = int(os.environ.get('REPS', 30))
g = torch.Generator().manual_seed(0)
strides = [4, 8, 16, 32]
out = {}


def timed(fn, n):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# ---- ROIAlign: 1000 FPN-consistent ROIs, SETS pyramids of 209 MB ----
pyramids = [[torch.randn(1, 256, 1280 // s, 1920 // s, device='cuda').contiguous(memory_format=torch.channels_last) for s in strides]
            for _ in range(SETS)]
n = 1000
size = torch.exp(torch.empty(n).uniform_(3.0, 6.5, generator=g))          # sqrt(area) 20 .. 665 px
ar = torch.exp(torch.empty(n).uniform_(-0.7, 0.7, generator=g))
w, h = size * ar.sqrt(), size / ar.sqrt()
cx = torch.empty(n).uniform_(0, 1920, generator=g); cy = torch.empty(n).uniform_(0, 1280, generator=g)
rois = torch.stack([torch.zeros(n), (cx - w / 2).clamp(0, 1920), (cy - h / 2).clamp(0, 1280), (cx + w / 2).clamp(0, 1920),
                    (cy + h / 2).clamp(0, 1280)], 1)
scales = [1.0 / s for s in strides]
us = timed(lambda i: ops.roi_pool_fpn(pyramids[i % SETS], rois.cuda(), scales), REPS)
# unique footprint: mark every feature pixel a ROI's samples can touch (aligned ROIAlign: x1*scale - 0.5 .. x2*scale - 0.5, + 1 px)
lvl = torch.floor(4 + torch.log2(torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])) / 224 + 1e-8)).clamp(2, 5).long()
unique_px, summed_px = 0, 0
for li, s in enumerate(strides):
    H, W = 1280 // s, 1920 // s
    m = np.zeros((H, W), bool)
    for r in rois[lvl == li + 2].numpy():
        x0, y0 = int(np.floor(r[1] / s - 0.5)), int(np.floor(r[2] / s - 0.5))
        x1, y1 = int(np.ceil(r[3] / s - 0.5)) + 1, int(np.ceil(r[4] / s - 0.5)) + 1
        x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, W), min(y1, H)
        m[y0:y1, x0:x1] = True
        summed_px += max(x1 - x0, 0) * max(y1 - y0, 0)
    unique_px += int(m.sum())
alg = unique_px * 256 * 4 + n * 20 + n * 49 * 256 * 4
out['roi_pool_wg_kernel'] = dict(us=us, algorithmic_bytes=alg, unique_footprint_bytes=unique_px * 256 * 4,
                                  per_roi_footprint_sum_bytes=summed_px * 256 * 4, output_bytes=n * 49 * 256 * 4,
                                  gbs=alg / us / 1e3, frac_of_8TBs=alg / us / 1e3 / 8000)
del pyramids
torch.cuda.empty_cache()

# ---- NMS mask: n boxes (RPN size) ----
for nb in (4741,):
    sets = []
    for _ in range(8):
        c = torch.rand((nb, 2), generator=g) * torch.tensor([1920.0, 1280.0]); wh = torch.rand((nb, 2), generator=g) * 200 + 20
        sets.append((torch.cat([c - wh / 2, c + wh / 2], 1).cuda().contiguous(),
                     torch.randint(0, 5, (nb,), generator=g, dtype=torch.int32).cuda()))
    us = timed(lambda i: ops.nms_sorted(sets[i % 8][0], sets[i % 8][1], 0.7), REPS)
    alg = 16 * nb + nb * nb // 8
    out['nms(n=%d) mask+sweep' % nb] = dict(us=us, algorithmic_bytes=alg, gbs=alg / us / 1e3, frac_of_8TBs=alg / us / 1e3 / 8000,
                                            pair_tests=nb * (nb - 1) // 2)

# ---- pre-processing: 1920x1280 uint8 frames, SETS*8 different frames (7.4 MB in + 29.5 MB out each) ----
frames = torch.randint(0, 256, (SETS * 8, 1280, 1920, 3), generator=g, dtype=torch.uint8).cuda()
for scale, hf in ((1.0, False), (1.5, True)):
    res = ops.preprocess(frames[:1], scale, hf, False, True, PIXEL_MEAN, PIXEL_STD, 32)[0]
    us = timed(lambda i: ops.preprocess(frames[i % len(frames):i % len(frames) + 1], scale, hf, False, True, PIXEL_MEAN, PIXEL_STD, 32), REPS)
    alg = 1280 * 1920 * 3 + res.numel() * 4
    out['preprocess_kernel(scale %.1f%s)' % (scale, ', hflip' if hf else '')] = dict(us=us, algorithmic_bytes=alg, gbs=alg / us / 1e3,
                                                                                 frac_of_8TBs=alg / us / 1e3 / 8000)
print(json.dumps(out))
