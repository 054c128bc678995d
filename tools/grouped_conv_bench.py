"""res2 grouped 3x3 conv (256 ch, groups 32, 320x480): hand-written kernel vs MIOpen grouped convolution."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from waymo_2d_tracking_amd.detnet.nn import ops
torch.backends.cudnn.benchmark = True
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for C, H, W in ((256, 320, 480),):
    x = torch.randn(1, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, C // 32, 3, 3, device='cuda')
    wcl = w.contiguous(memory_format=torch.channels_last)
    pw = ops.deform_pack_weight(w, 32)
    sc = torch.ones(C, device='cuda'); bi = torch.zeros(C, device='cuda')
    t0 = bench(lambda: ops.deform_conv3x3(x, None, pw, 32, 1, 1, sc, bi, True))
    t1 = bench(lambda: F.conv2d(x, wcl, None, 1, 1, 1, 32))
    print('C=%d %dx%d groups 32: HIP kernel (+BN+ReLU fused) %.1f us, MIOpen grouped conv (conv only) %.1f us' % (C, H, W, t0, t1))
