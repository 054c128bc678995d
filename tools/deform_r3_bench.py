"""Round-3 deformable-conv bench (tools only): res4 / res3 / res5 shapes, persistent kernel with the in-kernel table and with the
per-layer pre-pass table, offset spreads from STDS, correctness against the round-1 LDS kernel.
    WT_LIB_PATH=.../variants/lib_x.so python tools/deform_r3_bench.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd import _lib
from waymo_2d_tracking_amd.detnet.nn import ops


def bench(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def table_for(off):
    """offsets (1,18,H,W) -> (offsets NHWC, table) through wd_deform_offsets_table_f32: the offsets are handed over as the centre-tap
    partial sums of the offset conv's GEMM (all other taps zero)."""
    n, _, h, w = off.shape
    partial = torch.zeros((n * h * w, 176), device='cuda')
    partial[:, 4 * 18:4 * 18 + 18] = off.permute(0, 2, 3, 1).reshape(-1, 18)
    lib = _lib.lib()
    out = torch.empty((n, 18, h, w), device='cuda').contiguous(memory_format=torch.channels_last)
    lib.wd_deform_table_bytes.restype = C.c_size_t
    table = torch.empty(int(lib.wd_deform_table_bytes(C.c_int(n), C.c_int(h), C.c_int(w))), dtype=torch.uint8, device='cuda')
    _lib.check(lib.wd_deform_offsets_table_f32(C.c_void_p(partial.data_ptr()), C.c_int(176), None, C.c_int(n), C.c_int(h), C.c_int(w),
                                               C.c_void_p(out.data_ptr()), C.c_void_p(table.data_ptr()),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'table')
    return out, table


shapes = {'res4': (1024, 80, 120), 'res3': (512, 160, 240), 'res5': (2048, 40, 60)}
for name in os.environ.get('SHAPES', 'res4').split(','):
    Cc, H, W = shapes[name]
    torch.manual_seed(0)
    x = torch.randn(1, Cc, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cc, Cc // 32, 3, 3, device='cuda')
    pw = ops.deform_pack_weight(w, 32)
    sc = torch.ones(Cc, device='cuda')
    bi = torch.zeros(Cc, device='cuda')
    gf = 2.0 * Cc * (Cc // 32) * 9 * H * W / 1e9
    for std in [float(v) for v in os.environ.get('STDS', '0,0.2,0.5,1,2').split(',')]:
        off = (torch.randn(1, 18, H, W, device='cuda') * std).contiguous(memory_format=torch.channels_last)
        os.environ['WD_DEFORM_PATCH'] = 'lds'
        ref = ops.deform_conv3x3(x, off, pw, 32, 1, 1, sc, bi, True)
        t_ref = bench(lambda: ops.deform_conv3x3(x, off, pw, 32, 1, 1, sc, bi, True))
        os.environ['WD_DEFORM_PATCH'] = 'pp'
        got = ops.deform_conv3x3(x, off, pw, 32, 1, 1, sc, bi, True)
        t_pp = bench(lambda: ops.deform_conv3x3(x, off, pw, 32, 1, 1, sc, bi, True))
        line = '%s std %.1f: r1 %.1f us  pp %.1f us (%.2f)' % (name, std, t_ref, t_pp, gf / t_pp * 1e3 / 157.3)
        err = (got - ref).abs().max().item()
        if Cc // 32 in (16, 32):
            o2, tab = table_for(off)
            assert torch.equal(o2, off)
            got2 = ops.deform_conv3x3(x, off, pw, 32, 1, 1, sc, bi, True, table=tab)
            t_tab = bench(lambda: ops.deform_conv3x3(x, off, pw, 32, 1, 1, sc, bi, True, table=tab))
            line += '  pp+table %.1f us (%.2f)  table==inline %s' % (t_tab, gf / t_tab * 1e3 / 157.3, bool(torch.equal(got, got2)))
        print(line + '  max|r1-pp| %.2e' % err, flush=True)
