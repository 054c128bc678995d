"""Stride-2 deformable layers (first blocks of res3 / res4 / res5) through wd_deform_conv3x3_f32: time, fraction of the f32 MFMA peak,
agreement with the round-1 gather kernel.   WD_PP_S2=narrow python tools/deform_s2_bench.py  = the round-3 routing (14 x 14 patches)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops


def bench(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = {'res3s2': (512, 320, 480), 'res4s2': (1024, 160, 240), 'res5s2': (2048, 80, 120)}
for name in os.environ.get('SHAPES', 'res3s2,res4s2,res5s2').split(','):
    Cc, H, W = shapes[name]
    Ho, Wo = H // 2, W // 2
    torch.manual_seed(0)
    x = torch.randn(1, Cc, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cc, Cc // 32, 3, 3, device='cuda')
    pw = ops.deform_pack_weight(w, 32)
    sc = torch.ones(Cc, device='cuda')
    bi = torch.zeros(Cc, device='cuda')
    gf = 2.0 * Cc * (Cc // 32) * 9 * Ho * Wo / 1e9
    for std in [float(v) for v in os.environ.get('STDS', '0.2,1,2').split(',')]:
        off = (torch.randn(1, 18, Ho, Wo, device='cuda') * std).contiguous(memory_format=torch.channels_last)
        os.environ['WD_DEFORM_PATCH'] = 'lds'
        ref = ops.deform_conv3x3(x, off, pw, 32, 2, 1, sc, bi, True)
        t_ref = bench(lambda: ops.deform_conv3x3(x, off, pw, 32, 2, 1, sc, bi, True))
        os.environ['WD_DEFORM_PATCH'] = 'pp'
        got = ops.deform_conv3x3(x, off, pw, 32, 2, 1, sc, bi, True)
        t_pp = bench(lambda: ops.deform_conv3x3(x, off, pw, 32, 2, 1, sc, bi, True))
        print('%s std %.1f: r1 %.1f us (%.2f)  default %.1f us (%.2f)  max|diff| %.2e' % (name, std, t_ref, gf / t_ref * 1e3 / 157.3, t_pp,
                                                                                       gf / t_pp * 1e3 / 157.3, (got - ref).abs().max().item()), flush=True)
