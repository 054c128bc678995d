import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
g = torch.Generator().manual_seed(0)
for n in (1000, 4741):
    c = torch.rand((n, 2), generator=g) * torch.tensor([1920.0, 1280.0]); wh = torch.rand((n, 2), generator=g) * 200 + 20
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1).cuda(); scores = torch.rand(n, generator=g).cuda()
    idx = torch.randint(0, 5, (n,), generator=g, dtype=torch.int32).cuda()
    order = torch.argsort(scores, descending=True); b = boxes[order].contiguous(); i = idx[order].contiguous()
    f = lambda: ops.nms_sorted(b, i, 0.7)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print('nms n=%d: %.1f us (kept %d)' % (n, e0.elapsed_time(e1) / 20 * 1e3, int(f().sum())))
