"""Split-operand GEMM (csrc/det_gemm_split.hip) against the library f32 GEMM / convolution on the detector's shapes: time and error vs float64.

    python tools/gemm_split_bench.py [--iters 20] [--json out.json] [--quick]

For every shape: microseconds of wd_gemm_split_f32 / wd_conv_split_f32, of torch's f32 path (hipBLASLt addmm or MIOpen conv2d) and - GEMMs only -
of the hand-written exact-f32 MFMA kernel (wd_gemm_nt_f32); max / rms error of each against a float64 reference on the same inputs.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.nn.functional as F

from waymo_2d_tracking_amd.detnet.nn import ops


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def err(y, ref):
    d = (y.double() - ref)
    return float(d.abs().max()), float(d.pow(2).mean().sqrt())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--json', default=None)
    ap.add_argument('--quick', action='store_true')
    args = ap.parse_args()
    torch.manual_seed(0)
    dev = 'cuda'
    rows = []
    gemms = [('res4 1x1 9600x1024x1024', 9600, 1024, 1024), ('res3 1x1 38400x512x512', 38400, 512, 512),
             ('res2 1x1 153600x256x256', 153600, 256, 256), ('res5 1x1 2400x2048x2048', 2400, 2048, 2048),
             ('res5 conv1 2400x2048x1024', 2400, 2048, 1024), ('fpn lateral p4 9600x256x1024', 9600, 256, 1024)]
    if args.quick:
        gemms = gemms[:1]
    for name, m, n, k in gemms:
        a = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev) / k ** 0.5
        bias = torch.randn(n, device=dev)
        res = torch.randn(m, n, device=dev)
        pw = ops.split_pack_weight(w)
        ref = torch.relu(a.double() @ w.double().t() + bias.double() + res.double())
        y = ops.gemm_split(a, pw, n, bias, res, True)
        y32 = torch.relu(torch.addmm(bias, a, w.t()) + res)
        yown = ops.gemm_nt(a, w, bias, res, True)
        e_s, e_l, e_o = err(y, ref), err(y32, ref), err(yown, ref)
        out = torch.empty_like(y)
        t_s = timeit(lambda: ops.gemm_split(a, pw, n, bias, res, True, out=out), args.iters)
        t_p = timeit(lambda: ops.gemm_split(a, pw, n, None, None, False, out=out), args.iters)
        t_l = timeit(lambda: torch.addmm(bias, a, w.t(), out=out), args.iters)
        t_lt = timeit(lambda: ops.gemm_lt(a, w, bias, res, True, out=out), args.iters)
        flops = 2.0 * m * n * k
        rows.append(dict(shape=name, split_us=t_s, split_plain_us=t_p, lib_addmm_us=t_l, lib_lt_residual_us=t_lt,
                         split_tflops_f32_equiv=flops / t_s * 1e-6, split_tflops_bf16=6 * flops / t_s * 1e-6, lib_tflops=flops / t_lt * 1e-6,
                         err_split=e_s, err_lib_f32=e_l, err_own_f32=e_o, rms_ratio_vs_lib=e_s[1] / e_l[1], rms_ratio_vs_own=e_s[1] / e_o[1]))
        print(json.dumps(rows[-1]), flush=True)
        del a, w, res, ref, y, y32, yown, out
    convs = [('box head 3x3 1000x256x7x7', 1000, 256, 7, 7, 256), ('fpn out p3 3x3 256x160x240', 1, 256, 160, 240, 256),
             ('fpn out p2 3x3 256x320x480', 1, 256, 320, 480, 256)]
    if args.quick:
        convs = convs[:1]
    for name, b, c, h, w_, n in convs:
        x = torch.randn(b, c, h, w_, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(n, c, 3, 3, device=dev) / (9 * c) ** 0.5).contiguous(memory_format=torch.channels_last)
        bias = torch.randn(n, device=dev)
        pw = ops.split_pack_weight(w)
        ref = F.conv2d(x.double(), w.double(), bias.double(), 1, 1)
        y = ops.conv_split(x, pw, n, 3, 1, 1, bias)
        y32 = F.conv2d(x, w, bias, 1, 1)
        e_s, e_l = err(y, ref), err(y32, ref)
        torch.backends.cudnn.benchmark = True
        t_s = timeit(lambda: ops.conv_split(x, pw, n, 3, 1, 1, bias, None, True), args.iters)
        t_l = timeit(lambda: F.conv2d(x, w, None, 1, 1), args.iters)
        flops = 2.0 * b * h * w_ * n * 9 * c
        rows.append(dict(shape=name, split_us=t_s, lib_conv_us=t_l, split_tflops_f32_equiv=flops / t_s * 1e-6, split_tflops_bf16=6 * flops / t_s * 1e-6,
                         lib_tflops=flops / t_l * 1e-6, err_split=e_s, err_lib_f32=e_l, rms_ratio_vs_lib=e_s[1] / e_l[1]))
        print(json.dumps(rows[-1]), flush=True)
        del x, w, ref, y, y32
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()
