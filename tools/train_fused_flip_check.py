import sys, os
sys.path.insert(0, '/root/repo')
os.chdir('/root/repo')
import torch, re
from oracle import detector_ref as R
from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
from waymo_2d_tracking_amd.detnet.nn import training, cascade_rcnn
m = Detectron2Det(seed=4).cuda().train()
training.set_trainable(m.model)
g = torch.Generator().manual_seed(12)
img = torch.randint(0, 256, (1, 3, 160, 224), generator=g).float().cuda()
gt = torch.tensor([[20., 30., 120., 150.], [100., 40., 215., 155.], [5., 5., 60., 60.]]).cuda()
cls = torch.tensor([0, 1, 3]).cuda()
cfg = dict(pre_nms=300, post_nms=200, rpn_batch=64, rpn_pos=0.5, roi_batch=128, roi_pos=0.25)
names = [n for n, _ in m.model.backbone.named_modules() if re.fullmatch(r'res\d\.\d+', n)]
for rep in range(4):
    out = {}
    for fused in (False, True):
        cascade_rcnn.FUSED_TRAINING_EPILOGUES = fused
        for p in m.model.parameters():
            p.grad = None
        cascade_rcnn.DECISION_LOG = log = []
        losses = training.losses(m.model, img, gt, cls, choose=training.first_choice, config=cfg)
        cascade_rcnn.DECISION_LOG = None
        sum(losses.values()).backward()
        out[fused] = ({n: p.grad.clone() for n, p in m.model.named_parameters() if p.requires_grad and p.grad is not None}, [R.block_decisions(*t) for t in log])
    (g0, d0), (g1, d1) = out[False], out[True]
    rel = {n: float((g0[n] - g1[n]).abs().max() / (g0[n].abs().max() + 1e-30)) for n in g0}
    fl = []
    for name, a, b in zip(names, d0, d1):
        nr = [int((a[k] != b[k]).sum()) for k in ('relu1', 'relu2', 'relu3')]
        nc = 0 if a['cells'] is None else int((a['cells'] != b['cells']).sum())
        worst = max([v for n, v in rel.items() if n.startswith('backbone.' + name + '.')] or [0.0])
        if sum(nr) or nc or worst > 2e-3:
            fl.append('%s relu %s cells %d worst %.2e' % (name, nr, nc, worst))
    other = max(v for n, v in rel.items() if not n.startswith('backbone.res'))
    print('run %d: worst %.2e median %.2e non-bottleneck worst %.2e' % (rep, max(rel.values()), sorted(rel.values())[len(rel)//2], other))
    for f in fl: print('   ', f)
