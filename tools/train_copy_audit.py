"""Which tensors does one training step copy (aten::copy_ / contiguous / clone) and from where?  torch.profiler over two steps of the bench's
training loop, device time per (op, shape, python caller).  Tools only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
from torch.profiler import profile, ProfilerActivity
from waymo_2d_tracking_amd.detnet.nn import training
from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
from waymo_2d_tracking_amd.tuning import enable_gemm_tuning

torch.backends.cudnn.benchmark = True
enable_gemm_tuning()
dev = torch.device('cuda')
det = Detectron2Det(seed=0).to(dev).train()
params = training.set_trainable(det.model)
opt = torch.optim.SGD(params, lr=0.002, momentum=0.9, weight_decay=1e-4, fused=True)
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 256, (1, 3, 886, 1280), generator=g).float().to(dev)
wh = torch.rand((30, 2), generator=g) * 280 + 20
xy = torch.rand((30, 2), generator=g) * torch.tensor([1280 - 300.0, 886 - 300.0])
boxes = torch.cat((xy, xy + wh), 1).to(dev)
classes = torch.randint(0, 4, (30,), generator=g).to(dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss = sum(training.losses(det.model, img, boxes, classes).values())
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 35.0)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::add_', 'aten::add', 'aten::fill_', 'aten::zero_', 'aten::mul') and ev.device_time_total > 0:
        where = ''
        for fr in (ev.stack or []):
            if 'waymo_2d_tracking_amd' in fr or 'autograd' in fr:
                where = fr.split('/')[-1][:70]
                break
        key = (ev.name, str(ev.input_shapes)[:70], where)
        acc[key][0] += 1
        acc[key][1] += ev.device_time_total
for key, (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%8.1f us %4d x  %-14s %-70s %s' % (us, n, key[0], key[1], key[2]))
