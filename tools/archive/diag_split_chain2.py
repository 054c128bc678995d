"""res2-block-shaped chain (6144 rows: 64 -> 256 shortcut, 64 -> 256 conv1, 256 -> 256 conv3 + residual) on two streams, new data per iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
torch.manual_seed(0)
NIT = int(os.environ.get('DIAG_ITERS', '20'))
M = int(os.environ.get('CHAIN_M', '6144'))
ws = [ops.split_pack_weight(torch.randn(256, 64, device='cuda') / 8) for _ in range(2)]
w1 = [ops.split_pack_weight(torch.randn(256, 64, device='cuda') / 8) for _ in range(2)]
w3 = [ops.split_pack_weight(torch.randn(256, 256, device='cuda') / 16 * 0.25) for _ in range(2)]
w1b = [ops.split_pack_weight(torch.randn(256, 256, device='cuda') / 16) for _ in range(2)]
xs = [[torch.randn(M, 64, device='cuda') for _ in range(2)] for _ in range(NIT)]


def chain(i, x):
    sc = ops.gemm_split(x, ws[i], 256)
    y = ops.gemm_split(x, w1[i], 256, None, None, True)
    out = ops.gemm_split(y, w3[i], 256, None, sc, True, out=sc)
    outs = [out.clone()]
    for _ in range(2):
        y = ops.gemm_split(out, w1b[i], 256, None, None, True)
        out = ops.gemm_split(y, w3[i], 256, None, out, True, out=out)
        outs.append(out.clone())
    return outs


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
        for k in range(3):
            if not torch.equal(refs[it][i][k], got[i][k]):
                bad += 1
                print('iteration %d stream %d block %d differs: max |d| %.3e' % (it, i, k, float((refs[it][i][k] - got[i][k]).abs().max())))
                break
print('mismatches:', bad)
