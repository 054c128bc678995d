import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from waymo_2d_tracking_amd.detnet.nn import ops
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (h, w) in ((40, 60), (80, 120), (160, 240)):
    x = torch.randn((1, 256, h, w), device='cuda').contiguous(memory_format=torch.channels_last)
    a = bench(lambda: F.interpolate(x, scale_factor=2.0, mode='nearest').contiguous(memory_format=torch.channels_last))
    b = bench(lambda: ops.upsample2x_nearest(x))
    mb = x.numel() * 4 * 5 / 1e6
    print('%dx%d -> x2: F.interpolate %.1f us, own %.1f us (%.2f TB/s)' % (h, w, a, b, mb / b))
