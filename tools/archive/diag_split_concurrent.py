"""Two streams issuing split-operand GEMMs / convolutions at the same time must give the results of serial launches, bit for bit."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops

torch.manual_seed(0)
shapes = [(384, 1024, 1024), (1536, 512, 512), (6144, 256, 256), (96, 2048, 2048), (24, 256, 2048), (1000, 1024, 12544)]
cases = []
for m, n, k in shapes:
    a = torch.randn(m, k, device='cuda'); w = torch.randn(n, k, device='cuda') / k ** 0.5
    b = torch.randn(n, device='cuda'); r = torch.randn(m, n, device='cuda')
    cases.append((a, ops.split_pack_weight(w), n, b, r))
convs = []
for bsz, c, h, wd, n in [(1, 256, 16, 24, 256), (1, 256, 4, 6, 256), (64, 256, 7, 7, 256)]:
    x = torch.randn(bsz, c, h, wd, device='cuda').contiguous(memory_format=torch.channels_last)
    wt = torch.randn(n, c, 3, 3, device='cuda') / 48
    convs.append((x, ops.split_pack_weight(wt), n))


def run_all():
    outs = []
    for a, pw, n, b, r in cases:
        outs.append(ops.gemm_split(a, pw, n, b, r, True))
    for x, pw, n in convs:
        outs.append(ops.conv_split(x, pw, n, 3, 1, 1))
    return outs


ref = run_all()
torch.cuda.synchronize()
again = run_all()
torch.cuda.synchronize()
print('serial repeat equal:', all(torch.equal(x, y) for x, y in zip(ref, again)))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = 0
for it in range(10):
    with torch.cuda.stream(s1):
        o1 = run_all()
    with torch.cuda.stream(s2):
        o2 = run_all()
    torch.cuda.synchronize()
    for i, (x, y, z) in enumerate(zip(ref, o1, o2)):
        if not torch.equal(x, y) or not torch.equal(x, z):
            bad += 1
            print('iteration %d case %d differs: %.3e %.3e' % (it, i, float((x - y).abs().max()), float((x - z).abs().max())))
print('concurrent mismatches:', bad)
