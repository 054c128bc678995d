"""Are the remaining library GEMMs (offset-conv mm, RPN head addmm) bit-reproducible with other kernels in flight on a second stream?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
torch.manual_seed(0)
torch.backends.cudnn.deterministic = True
shapes = [(6144, 176, 256, 'offset res2?'), (1536, 176, 512, 'offset res3'), (384, 176, 1024, 'offset res4'), (96, 176, 2048, 'offset res5'),
          (6144, 3, 256, 'rpn obj p2'), (6144, 12, 256, 'rpn deltas p2'), (1536, 12, 256, 'rpn p3'), (384, 12, 256, 'rpn p4'), (1000, 5, 1024, 'cls'), (1000, 4, 1024, 'box')]
cases = []
for m, n, k, name in shapes:
    a = torch.randn(m, k, device='cuda'); w = torch.randn(n, k, device='cuda'); b = torch.randn(n, device='cuda')
    cases.append((a, w, b, name))
big_a = torch.randn(4096, 1024, device='cuda'); big_w = torch.randn(1024, 1024, device='cuda'); pw = ops.split_pack_weight(big_w)


def run():
    out = []
    for a, w, b, name in cases:
        out.append(torch.mm(a, w.t()))
        out.append(torch.addmm(b, a, w.t()))
    return out


ref = run(); torch.cuda.synchronize()
print('serial repeat equal:', all(torch.equal(x, y) for x, y in zip(ref, run())))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = {}
for it in range(20):
    with torch.cuda.stream(s2):
        for _ in range(6):
            ops.gemm_split(big_a, pw, 1024)
    with torch.cuda.stream(s1):
        got = run()
    torch.cuda.synchronize()
    for i, (x, y) in enumerate(zip(ref, got)):
        if not torch.equal(x, y):
            key = cases[i // 2][3] + (' mm' if i % 2 == 0 else ' addmm')
            bad[key] = bad.get(key, 0) + 1
print('mismatching launches under concurrency (of 20):', bad)
