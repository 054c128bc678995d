"""Microbench of wd_preprocess_f32 on one 1920x1280 frame: uint8 HWC / float NCHW sources, scale 1 and --tta x1.5,hflip."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
from waymo_2d_tracking_amd.detnet.nn.cascade_rcnn import PIXEL_MEAN, PIXEL_STD
g = torch.Generator().manual_seed(0)
u8 = torch.randint(0, 256, (1, 1280, 1920, 3), generator=g, dtype=torch.uint8).cuda()
f32 = u8.permute(0, 3, 1, 2).float().contiguous()
for name, src, esz in (('u8 HWC', u8, 1), ('f32 NCHW', f32, 4)):
    for scale, hf in ((1.0, False), (1.5, True)):
        f = lambda: ops.preprocess(src, scale, hf, False, True, PIXEL_MEAN, PIXEL_STD, 32)
        for _ in range(3): out, _ = f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        alg = 1280 * 1920 * 3 * esz + out.numel() * 4
        print('preprocess %-9s scale %.1f hflip %d: %7.1f us, algorithmic %.1f MB -> %.0f GB/s (%.1f%% of 8 TB/s)'
              % (name, scale, hf, us, alg / 1e6, alg / us / 1e3, alg / us / 1e3 / 80))
