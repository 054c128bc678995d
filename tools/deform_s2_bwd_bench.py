"""Backward of the two stride-2 DeformConvs of res3 / res4 at the training shapes (tools only): fused kernels vs the column-slab form
(WD_FUSED_DEFORM_S2=0), HIP-event timed over DeformConvFn forward + backward minus forward."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops

SC = float(os.environ.get('OFF_SCALE', '0.5'))
for (C, G, H, W) in ((512, 32, 224, 320), (1024, 32, 112, 160)):
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = torch.randn(1, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    off = (torch.randn(1, 18, Ho, Wo, device='cuda') * SC).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(C, C // G, 3, 3, device='cuda') * 0.05).requires_grad_(True)
    dy = torch.randn(1, C, Ho, Wo, device='cuda').contiguous(memory_format=torch.channels_last)

    def fwd():
        with torch.no_grad():
            ops.DeformConvFn.apply(x, off, w, G, 2, 1)

    def both():
        y = ops.DeformConvFn.apply(x, off, w, G, 2, 1)
        y.backward(dy)
        x.grad = None; off.grad = None; w.grad = None
    t = {}
    for name, f in (('fwd', fwd), ('both', both)):
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
    print('C=%d in %dx%d stride 2: forward %.1f us, backward %.1f us' % (C, H, W, t['fwd'], t['both'] - t['fwd']), flush=True)
