"""Microbench: wd_gemm_nt_f32 vs torch (hipBLASLt) on the detector's GEMM shapes (fp32)."""
import sys, os, time
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

shapes = [(153600, 256, 256), (153600, 256, 64), (38400, 512, 512), (38400, 512, 256), (9600, 1024, 1024), (9600, 1024, 512),
          (2400, 2048, 2048), (2400, 2048, 1024), (1000, 1024, 12544), (153600, 256, 256)]
torch.backends.cuda.matmul.allow_tf32 = False
for M, N, K in shapes:
    a = torch.randn(M, K, device='cuda'); bt = torch.randn(N, K, device='cuda') / K ** 0.5
    bias = torch.randn(N, device='cuda'); res = torch.randn(M, N, device='cuda')
    t_mine = bench(lambda: ops.gemm_nt(a, bt, bias, res if K < 8192 else None, True))
    t_lin = bench(lambda: torch.nn.functional.linear(a, bt, bias))
    t_full = bench(lambda: torch.relu_(torch.nn.functional.linear(a, bt, bias).add_(res)))
    gf = 2.0 * M * N * K / 1e9
    print('M=%6d N=%5d K=%5d  %6.1f GF | mine %7.1f us %6.1f TF | torch linear %7.1f us %6.1f TF | linear+add+relu %7.1f us %6.1f TF'
          % (M, N, K, gf, t_mine, gf / t_mine * 1e3, t_lin, gf / t_lin * 1e3, t_full, gf / t_full * 1e3), flush=True)
