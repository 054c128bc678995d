"""Offset conv (C -> 18, 3x3): MIOpen direct conv vs GEMM + tap shift-add, on the res3 / res4 / res5 shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from waymo_2d_tracking_amd.detnet.nn import ops
torch.backends.cudnn.benchmark = True
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for C, H, W in ((512, 160, 240), (1024, 80, 120), (2048, 40, 60)):
    x = torch.randn(1, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = (torch.randn(18, C, 3, 3, device='cuda') * 0.01).contiguous(memory_format=torch.channels_last)
    b = torch.randn(18, device='cuda')
    t0 = timeit(lambda: F.conv2d(x, w, b, 1, 1))
    for align in (2, 16, 32, 64):
        w2 = ops.tap_gemm_weight(w, align)
        a = x.permute(0, 2, 3, 1).reshape(H * W, C)
        tg = timeit(lambda: torch.mm(a, w2.t()))
        t1 = timeit(lambda: ops.conv3x3_few(x, w2, b, 18, 1))
        print('C=%d %dx%d: conv2d %.1f us | ld=%d gemm %.1f us, gemm+shift-add %.1f us' % (C, H, W, t0, w2.shape[0], tg, t1))
