"""res2 block chain incl. the grouped 3x3 kernel: conv1 (split GEMM) -> grouped conv (own kernel, LDS-DMA patches) -> conv3 (split GEMM + residual),
two streams, new data per iteration.  CHAIN_GEMM=lib runs the two 1x1 convs through torch instead."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
torch.manual_seed(0)
NIT = int(os.environ.get('DIAG_ITERS', '20'))
LIB = os.environ.get('CHAIN_GEMM', 'split') == 'lib'
H, W = 64, 96
W1 = [torch.randn(256, 256, device='cuda') / 16 for _ in range(2)]
W3 = [torch.randn(256, 256, device='cuda') / 16 * 0.25 for _ in range(2)]
w1 = [ops.split_pack_weight(w) for w in W1]
w3 = [ops.split_pack_weight(w) for w in W3]
wg = [ops.deform_pack_weight(torch.randn(256, 8, 3, 3, device='cuda') / 8.5, 32) for _ in range(2)]
xs = [[torch.randn(1, 256, H, W, device='cuda').contiguous(memory_format=torch.channels_last) for _ in range(2)] for _ in range(NIT)]


def chain(i, x):
    outs = []
    for _ in range(3):
        a = x.permute(0, 2, 3, 1).reshape(H * W, 256)
        if LIB:
            y = torch.relu(a @ W1[i].t())
        else:
            y = ops.gemm_split(a, w1[i], 256, None, None, True)
        y4 = y.view(1, H, W, 256).permute(0, 3, 1, 2)
        z4 = ops.deform_conv3x3(y4, None, wg[i], 32, 1, 1, None, None, True)
        z = z4.permute(0, 2, 3, 1).reshape(H * W, 256)
        if LIB:
            o = torch.relu(z @ W3[i].t() + a)
        else:
            o = ops.gemm_split(z, w3[i], 256, None, a, True, out=a)
        x = o.view(1, H, W, 256).permute(0, 3, 1, 2)
        outs.append(x.clone())
    return outs


refs = []
for it in range(NIT):
    refs.append([chain(i, xs[it][i].clone()) for i in range(2)])
    torch.cuda.synchronize()
s = [torch.cuda.Stream(), torch.cuda.Stream()]
bad = 0
for it in range(NIT):
    got = [None, None]
    for i in range(2):
        with torch.cuda.stream(s[i]):
            got[i] = chain(i, xs[it][i].clone())
    torch.cuda.synchronize()
    for i in range(2):
        for k in range(3):
            if not torch.equal(refs[it][i][k], got[i][k]):
                bad += 1
                print('iteration %d stream %d block %d differs: max |d| %.3e' % (it, i, k, float((refs[it][i][k] - got[i][k]).abs().max())))
                break
print('mismatches:', bad)
