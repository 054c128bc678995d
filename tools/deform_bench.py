"""Microbench of wd_deform_conv3x3_f32 on the detector's layer shapes (1920x1280 input)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops

def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for name, C, H, W, stride, deform in [('res2', 256, 320, 480, 1, False), ('res3', 512, 160, 240, 1, True), ('res3s2', 512, 320, 480, 2, True),
                                      ('res4', 1024, 80, 120, 1, True), ('res5', 2048, 40, 60, 1, True)]:
    x = torch.randn(1, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    off = (torch.randn(1, 18, Ho, Wo, device="cuda") * float(os.environ.get("OFF_STD", "1.5"))).contiguous(memory_format=torch.channels_last) if deform else None
    w = torch.randn(C, C // 32, 3, 3, device='cuda')
    pw = ops.deform_pack_weight(w, 32)
    sc = torch.ones(C, device='cuda'); bi = torch.zeros(C, device='cuda')
    t = bench(lambda: ops.deform_conv3x3(x, off, pw, 32, stride, 1, sc, bi, True))
    gf = 2.0 * C * (C // 32) * 9 * Ho * Wo / 1e9
    print('%-7s C=%4d %3dx%3d s%d  %6.2f GF  %7.1f us  %6.1f TF' % (name, C, Ho, Wo, stride, gf, t, gf / t * 1e3), flush=True)
