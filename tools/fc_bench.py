"""Box-head FC (1024 x 12544 -> 1024, bias + ReLU): own split-K MFMA kernel vs hipBLASLt with epilogue (tools only)."""
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


for M, N, K in ((1024, 1024, 12544), (1000, 1024, 12544)):
    a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    gf = 2.0 * M * N * K / 1e9
    t1 = bench(lambda: ops.gemm_nt(a, w, b, None, True))
    t2 = bench(lambda: ops.gemm_lt(a, w, b, None, True))
    t3 = bench(lambda: torch._addmm_activation(b, a, w.t(), use_gelu=False))
    err = (ops.gemm_nt(a, w, b, None, True) - ops.gemm_lt(a, w, b, None, True)).abs().max().item()
    print('%dx%dx%d: gemm_nt %.1f us (%.0f TF)  gemm_lt %.1f us (%.0f TF)  torch addmm_activation %.1f us (%.0f TF)  max diff %.2e'
          % (M, N, K, t1, gf / t1 * 1e3, t2, gf / t2 * 1e3, t3, gf / t3 * 1e3, err))
