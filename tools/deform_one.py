"""Run only the res4 deformable conv shape a few times (for PMC collection)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
C, H, W = int(os.environ.get('DC', 1024)), int(os.environ.get('DH', 80)), int(os.environ.get('DW', 120))
x = torch.randn(1, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
off = (torch.randn(1, 18, H, W, device="cuda") * float(os.environ.get("OFF_STD", "0.7"))).contiguous(memory_format=torch.channels_last)
pw = ops.deform_pack_weight(torch.randn(C, C // 32, 3, 3, device='cuda'), 32)
sc = torch.ones(C, device='cuda'); bi = torch.zeros(C, device='cuda')
for _ in range(5):
    ops.deform_conv3x3(x, off, pw, 32, 1, 1, sc, bi, True)
torch.cuda.synchronize()
