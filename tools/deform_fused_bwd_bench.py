"""Fused deformable backward kernels alone (tools only): wd_deform_dw_f32 (+ wd_deform_dxoff_f32 when present) at the res4 / res3 training
shapes, HIP-event timed.  WT_LIB_PATH selects a variant library (tools/build_variant.sh)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops

SC = float(os.environ.get('OFF_SCALE', '0.5'))
for (C, G, H, W) in ((1024, 32, 56, 80), (512, 32, 112, 160)):
    x = torch.randn(1, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    off = (torch.randn(1, 18, H, W, device='cuda') * SC).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(1, H, W, C, device='cuda')
    w = torch.randn(C, C // G, 3, 3, device='cuda') * 0.05
    fns = [('dw', lambda: ops.deform_dw(x, off, dy, G))]
    if hasattr(ops, 'deform_dxoff'):
        fns.append(('dx+doffset', lambda: ops.deform_dxoff(x, off, dy, w, G)))
    for name, f in fns:
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        print('C=%d %dx%d %-11s %8.1f us' % (C, H, W, name, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
