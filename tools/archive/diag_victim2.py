"""Roles swapped: stream 0 repeats a small-tile split-operand GEMM (MT = 2) and checks every output; stream 1 loops the res5 stride-2 deformable conv
(deform_conv3x3_kernel<64, true>) or another kernel.  AGGRESSOR = deform64 | gconv | none"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
torch.manual_seed(0)
AGG = os.environ.get('AGGRESSOR', 'deform64')
o1 = (torch.randn(1, 2048, 16, 24, device='cuda') * 0.5).contiguous(memory_format=torch.channels_last)
off = (torch.randn(1, 18, 8, 12, device='cuda') * 0.3).contiguous(memory_format=torch.channels_last)
wgt = ops.deform_pack_weight(torch.randn(2048, 64, 3, 3, device='cuda') / 24, 32)
a2 = torch.randn(6144, 256, device='cuda'); w2 = ops.split_pack_weight(torch.randn(256, 256, device='cuda') / 16)
ref = ops.gemm_split(a2, w2, 256, None, None, True).clone()
xg = torch.randn(1, 256, 64, 96, device='cuda').contiguous(memory_format=torch.channels_last)
wg = ops.deform_pack_weight(torch.randn(256, 8, 3, 3, device='cuda') / 8.5, 32)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
outs = []
for rep in range(200):
    with torch.cuda.stream(s1):
        for _ in range(3):
            if AGG == 'deform64':
                ops.deform_conv3x3(o1, off, wgt, 32, 2, 1, None, None, True)
            elif AGG == 'gconv':
                ops.deform_conv3x3(xg, None, wg, 32, 1, 1, None, None, True)
    with torch.cuda.stream(s0):
        outs.append(ops.gemm_split(a2, w2, 256, None, None, True))
torch.cuda.synchronize()
bad = sum(0 if torch.equal(y, ref) else 1 for y in outs)
print('aggressor %s: %d of %d split-GEMM launches differ' % (AGG, bad, len(outs)))
