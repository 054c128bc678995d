"""Guaranteed overlap: two streams, each a dependent chain of split-operand GEMMs enqueued deep enough (REPS launches per chain) that both
queues stay full.  CHAIN_M rows (6144 -> MT = 2 tiles whose 64 KiB of LDS let two workgroups share a CU; 38400 -> MT = 5, one per CU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
torch.manual_seed(0)
NIT = int(os.environ.get('DIAG_ITERS', '6'))
M = int(os.environ.get('CHAIN_M', '6144'))
C = int(os.environ.get('CHAIN_C', '256'))
REPS = int(os.environ.get('CHAIN_REPS', '40'))
w1 = [ops.split_pack_weight(torch.randn(C, C, device='cuda') / C ** 0.5) for _ in range(2)]
w3 = [ops.split_pack_weight(torch.randn(C, C, device='cuda') / C ** 0.5 * 0.25) for _ in range(2)]
xs = [[torch.randn(M, C, device='cuda') for _ in range(2)] for _ in range(NIT)]


def chain(i, x):
    x = x.clone()
    for _ in range(REPS):
        y = ops.gemm_split(x, w1[i], C, None, None, True)
        x = ops.gemm_split(y, w3[i], C, None, x, True, out=x)
    return x


refs = []
for it in range(NIT):
    refs.append([chain(i, xs[it][i]) for i in range(2)])
    torch.cuda.synchronize()
s = [torch.cuda.Stream(), torch.cuda.Stream()]
bad = 0
for it in range(NIT):
    got = [None, None]
    for i in range(2):
        with torch.cuda.stream(s[i]):
            got[i] = chain(i, xs[it][i])
    torch.cuda.synchronize()
    for i in range(2):
        if not torch.equal(refs[it][i], got[i]):
            bad += 1
            print('iteration %d stream %d differs: max |d| %.3e' % (it, i, float((refs[it][i] - got[i]).abs().max())))
print('M=%d C=%d mismatches: %d' % (M, C, bad))
