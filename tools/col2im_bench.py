"""Microbench of wd_deform_col2im_f32 (deformable conv backward: dx + doffset) on the training crop shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
def bench(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
std = float(os.environ.get('OFF_STD', '0.5'))
for name, C, H, W in (('res3', 512, 111, 160), ('res4', 1024, 56, 80), ('res5', 2048, 28, 40)):
    x = torch.randn(1, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    off = (torch.randn(1, 18, H, W, device='cuda') * std).contiguous(memory_format=torch.channels_last)
    dcol = torch.randn(32, H * W, 9, C // 32, device='cuda')
    t = bench(lambda: ops.deform_col2im(dcol, x, off, 1, 1, 32))
    print('%s C=%d %dx%d: col2im %.1f us (dcol %.0f MB)' % (name, C, H, W, t, dcol.numel() * 4 / 1e6), flush=True)
