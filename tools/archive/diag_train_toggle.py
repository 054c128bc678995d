"""Which tensor of the training graph toggles between two states from run to run (same inputs)?  Forward outputs and output gradients of every res4 / res5
bottleneck (block output, conv1 output, offsets) are captured over REPEATS runs and compared with run 0.

    REPEATS=8 python tools/diag_train_toggle.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
from waymo_2d_tracking_amd.detnet.nn import training, cascade_rcnn

torch.backends.cudnn.benchmark = os.environ.get('BENCHMARK', '0') == '1'
m = Detectron2Det(seed=4).cuda().train()
training.set_trainable(m.model)
g = torch.Generator().manual_seed(11)
img = torch.randint(0, 256, (1, 3, 160, 224), generator=g).float().cuda()
gt = torch.tensor([[20., 30., 120., 150.], [100., 40., 215., 155.], [5., 5., 60., 60.], [130., 8., 200., 70.]]).cuda()
cls = torch.tensor([0, 1, 3, 0]).cuda()
cfg = dict(pre_nms=300, post_nms=200, rpn_batch=64, rpn_pos=0.5, roi_batch=128, roi_pos=0.25)
fwd, bwd = {}, {}


def watch(name, mod):
    def hook(_m, _i, out):
        fwd[name] = out.detach().clone()
        if out.requires_grad:
            out.register_hook(lambda gr, n=name: bwd.__setitem__(n, gr.detach().clone()))
    mod.register_forward_hook(hook)


for name, mod in m.model.named_modules():
    if isinstance(mod, cascade_rcnn.Bottleneck) and ('res4' in name or 'res5' in name):
        watch(name, mod)
        watch(name + '.conv1', mod.conv1)
        if mod.deform:
            watch(name + '.conv2_offset', mod.conv2_offset)

proposals = None
runs = []
for r in range(int(os.environ.get('REPEATS', '8'))):
    fwd.clear(); bwd.clear()
    for p in m.model.parameters():
        p.grad = None
    inter = {}
    got = training.losses(m.model, img, gt, cls, choose=training.first_choice, config=cfg, proposals=proposals)
    sum(got.values()).backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in m.model.named_parameters() if p.grad is not None}
    runs.append((dict(fwd), dict(bwd), grads))
f0, b0, g0 = runs[0]
for r, (f, b, gr) in enumerate(runs[1:], 1):
    df = [(float((f[n] - f0[n]).abs().max() / f0[n].abs().max().clamp_min(1e-30)), n) for n in f0]
    db = [(float((b[n] - b0[n]).abs().max() / b0[n].abs().max().clamp_min(1e-30)), n) for n in b0 if n in b]
    dg = [(float((gr[n] - g0[n]).abs().max() / g0[n].abs().max().clamp_min(1e-30)), n) for n in g0]
    df.sort(reverse=True); db.sort(reverse=True); dg.sort(reverse=True)
    print('run %d vs run 0' % r)
    print('   forward outputs that differ: %d of %d; worst: %s' % (sum(1 for e, _ in df if e > 0), len(df), ', '.join('%.1e %s' % x for x in df[:3])))
    print('   output gradients, worst:  %s' % ', '.join('%.1e %s' % (e, n.replace('backbone.', '')) for e, n in db[:6]))
    print('   weight gradients, worst:  %s' % ', '.join('%.1e %s' % (e, n.replace('backbone.', '')) for e, n in dg[:5]))
    # the block whose OUTPUT gradient agrees (<1e-4) while its conv1-output / offset gradient does not: the toggle is inside that block's backward
    big = {n for e, n in db if e > 2e-3}
    if big:
        print('   gradients off by > 2e-3: %s' % sorted(big))
