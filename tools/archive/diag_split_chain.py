"""Chains of dependent split-operand launches (conv1 -> conv3 + residual, conv 3x3 behind) on two streams at once, new data every iteration:
every result must equal the serial run of the same chain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
torch.manual_seed(0)
NIT = int(os.environ.get('DIAG_ITERS', '20'))
INPLACE = os.environ.get('CHAIN_INPLACE', '1') == '1'
C = 1024
w1 = [ops.split_pack_weight(torch.randn(C, C, device='cuda') / C ** 0.5) for _ in range(2)]
w3 = [ops.split_pack_weight(torch.randn(C, C, device='cuda') / C ** 0.5 * 0.25) for _ in range(2)]
wc = [ops.split_pack_weight(torch.randn(256, C // 4, 3, 3, device='cuda') / 48) for _ in range(2)]
xs = [[torch.randn(384, C, device='cuda') for _ in range(2)] for _ in range(NIT)]


def chain(i, x):
    x = x.clone()
    for _ in range(4):                                    # four "bottleneck blocks": conv1, conv3 + residual (in place on the block input)
        y = ops.gemm_split(x, w1[i], C, None, None, True)
        x = ops.gemm_split(y, w3[i], C, None, x, True, out=x if INPLACE else None)
    img = x[:, :256].reshape(1, 16, 24, 256).permute(0, 3, 1, 2)       # channels_last view
    z = ops.conv_split(img, wc[i], 256, 3, 1, 1)
    return x.clone(), z.clone()


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
        for k in range(2):
            if not torch.equal(refs[it][i][k], got[i][k]):
                bad += 1
                print('iteration %d stream %d output %d differs: max |d| %.3e' % (it, i, k, float((refs[it][i][k] - got[i][k]).abs().max())))
print('mismatches:', bad)
