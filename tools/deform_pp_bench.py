"""res4 deformable conv: default kernel vs the ping-pong kernel at several offset spreads (tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops


def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [('res4', 1024, 80, 120)] + ([('res4tta', 1024, 120, 180)] if os.environ.get('TTA') else [])
for name, C, H, W in shapes:
    x = torch.randn(1, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, 32, 3, 3, device='cuda')
    pw = ops.deform_pack_weight(w, C // 32)
    sc = torch.ones(C, device='cuda'); bi = torch.zeros(C, device='cuda')
    gf = 2.0 * C * 32 * 9 * H * W / 1e9
    for std in [float(v) for v in os.environ.get('STDS', '0,0.5,2,4').split(',')]:
        off = (torch.randn(1, 18, H, W, device='cuda') * std).contiguous(memory_format=torch.channels_last)
        res = {}
        for mode in os.environ.get('MODES', 'lds,pp').split(','):
            os.environ['WD_DEFORM_PATCH'] = mode
            res[mode] = bench(lambda: ops.deform_conv3x3(x, off, pw, C // 32, 1, 1, sc, bi, True))
        os.environ['WD_DEFORM_PATCH'] = 'lds'
        a = ops.deform_conv3x3(x, off, pw, C // 32, 1, 1, sc, bi, True)
        os.environ['WD_DEFORM_PATCH'] = 'pp'
        b = ops.deform_conv3x3(x, off, pw, C // 32, 1, 1, sc, bi, True)
        err = (a - b).abs().max().item()
        print('%s std %.1f: ' % (name, std) + '  '.join('%s %.1f us (%.1f TF, %.2f of peak)' % (m, t, gf / t * 1e3, gf / t * 1e3 / 157.3)
                                                          for m, t in res.items()) + '  max|lds-pp| %.2e' % err, flush=True)
