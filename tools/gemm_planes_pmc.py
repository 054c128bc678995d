"""One variant of the res4-shaped split-operand GEMM, a few launches (the program behind tools/pmc_planes.sh):
    python tools/gemm_planes_pmc.py VARIANT M N K [launches]      VARIANT = f32 | planes_f32out | planes_all | f32_both"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops

var = sys.argv[1]
m, n, k = (int(v) for v in sys.argv[2:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
torch.manual_seed(0)
a = torch.randn(m, k, device='cuda')
w = torch.randn(n, k, device='cuda') / k ** 0.5
bias = torch.randn(n, device='cuda')
res = torch.randn(m, n, device='cuda')
pw = ops.split_pack_weight(w)
ap = ops.split_planes_pack(a)
rp = ops.split_planes_pack(res)
out = torch.empty(m, n, device='cuda')
op = ops.split_planes_empty(m, n, 'cuda')
for _ in range(reps):
    if var == 'f32':
        ops.gemm_split(a, pw, n, bias, res, True, out=out)
    elif var == 'planes_f32out':
        ops.gemm_split_io(m, n, k, pw, a_planes=ap, bias=bias, residual=res, relu=True, out=out)
    elif var == 'planes_all':
        ops.gemm_split_io(m, n, k, pw, a_planes=ap, bias=bias, residual_planes=rp, relu=True, want_out=False, out_planes=op)
    elif var == 'f32_both':
        ops.gemm_split_io(m, n, k, pw, a=a, bias=bias, residual=res, relu=True, out=out, out_planes=op)
torch.cuda.synchronize()
