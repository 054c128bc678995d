"""One dense 3x3 convolution shape on the split-operand kernel, graph-replay timing (planner knobs: WD_SPLIT_MT, WD_SPLIT_SPLITK).
    python tools/conv_split_one.py B C H W N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
b, c, h, w, n = (int(v) for v in sys.argv[1:6])
x = torch.randn(b, c, h, w, device='cuda').contiguous(memory_format=torch.channels_last)
pw = ops.split_pack_weight(torch.randn(n, c, 3, 3, device='cuda') / (9 * c) ** 0.5)
for _ in range(3):
    ops.conv_split(x, pw, n, 3, 1, 1)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
with torch.cuda.stream(st):
    ops.conv_split(x, pw, n, 3, 1, 1)
    with torch.cuda.graph(g, stream=st):
        for _ in range(10):
            ops.conv_split(x, pw, n, 3, 1, 1)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print('conv3x3 %dx%dx%dx%d -> %d  MT=%s SPLITK=%s: %.1f us' % (b, c, h, w, n, os.environ.get('WD_SPLIT_MT', 'auto'), os.environ.get('WD_SPLIT_SPLITK', 'auto'), e0.elapsed_time(e1) / 10 * 1e3))
