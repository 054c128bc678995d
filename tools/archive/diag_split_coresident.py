"""Single stream, one launch with more workgroups than CUs at a tile height whose LDS lets two workgroups share a CU (MT = 2: 64 KiB each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
torch.manual_seed(0)
for m, n, k in [(65536, 256, 256), (32768, 512, 512), (16384, 1024, 64)]:
    a = torch.randn(m, k, device='cuda'); w = torch.randn(n, k, device='cuda') / k ** 0.5
    ref = a.double() @ w.double().t()
    pw = ops.split_pack_weight(w)
    errs = []
    for _ in range(5):
        y = ops.gemm_split(a, pw, n)
        errs.append(float((y.double() - ref).abs().max()))
    print(m, n, k, 'max err over 5 launches:', ['%.2e' % e for e in errs])
