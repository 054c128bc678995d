"""Does a split-operand launch write outside its output?  The output is the middle third of a sentinel-filled buffer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
torch.manual_seed(0)
for m, n, k in [(6144, 256, 256), (6144, 256, 64), (384, 1024, 1024), (96, 2048, 2048), (1536, 512, 512), (6000, 256, 256), (100, 512, 128)]:
    a = torch.randn(m, k, device='cuda'); w = ops.split_pack_weight(torch.randn(n, k, device='cuda') / k ** 0.5)
    big = torch.full((3 * m * n,), 12345.0, device='cuda')
    out = big[m * n: 2 * m * n].view(m, n)
    for _ in range(5):
        ops.gemm_split(a, w, n, None, None, True, out=out)
    torch.cuda.synchronize()
    lo, hi = big[:m * n], big[2 * m * n:]
    print(m, n, k, 'guard below intact:', bool((lo == 12345.0).all()), ' above intact:', bool((hi == 12345.0).all()), ' output finite:', bool(torch.isfinite(out).all()))
