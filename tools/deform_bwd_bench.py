"""Deformable-conv backward (DeformConvFn.backward: im2col -> 2 batched GEMMs -> col2im) by problem size (tools only): does the time per
column byte drop when the 9 x C x P column slabs fit the 256 MB Infinity Cache?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops

C, G = 1024, 32
w = (torch.randn(C, C // G, 3, 3, device='cuda') * 0.05).requires_grad_(True)
import os as _os
SIZES = ((56, 80),) if _os.environ.get('DBW_ONE') == '1' else ((56, 80), (28, 80), (14, 80), (112, 80))
for (H, W) in SIZES:
    x = torch.randn(1, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    off = (torch.randn(1, 18, H, W, device='cuda') * 0.5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    dy = torch.randn(1, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)

    def step():
        y = ops.DeformConvFn.apply(x, off, w, G, 1, 1)
        y.backward(dy)
        x.grad = None; off.grad = None; w.grad = None

    def fwd_only():
        with torch.no_grad():
            ops.DeformConvFn.apply(x, off, w, G, 1, 1)
    for f, name in ((fwd_only, 'fwd'), (step, 'fwd+bwd')):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        col_mb = H * W * 9 * C * 4 / 1e6
        print('%3dx%-3d %-8s %8.1f us   columns %6.1f MB   %.2f us per column MB' % (H, W, name, us, col_mb, us / col_mb), flush=True)
