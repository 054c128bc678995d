"""Round 6: which DISCRETE decisions differ between the float32 HIP training graph and the float64 restatement, block by block, and how
the per-tensor gradient errors line up with them.  REPEATS runs of the HIP side against one float64 reference.
    python tools/train_flip_check.py [repeats]"""
import copy
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import detector_ref as R
from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
from waymo_2d_tracking_amd.detnet.nn import training, cascade_rcnn

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = Detectron2Det(seed=4).cuda().train()
training.set_trainable(m.model)
cpu = copy.deepcopy(m.model).cpu()
g = torch.Generator().manual_seed(11)
img = torch.randint(0, 256, (1, 3, 160, 224), generator=g).float()
gt = torch.tensor([[20., 30., 120., 150.], [100., 40., 215., 155.], [5., 5., 60., 60.], [130., 8., 200., 70.]])
cls = torch.tensor([0, 1, 3, 0])
cfg = dict(pre_nms=300, post_nms=200, rpn_batch=64, rpn_pos=0.5, roi_batch=128, roi_pos=0.25)
ref, inter = R.losses(cpu, img, gt, cls, torch.float64, cfg['rpn_batch'], cfg['rpn_pos'], cfg['pre_nms'], cfg['post_nms'], cfg['roi_batch'], cfg['roi_pos'],
                      return_intermediates=True)
sum(ref.values()).backward()
rp = dict(cpu.named_parameters())
names = [n for n, _ in m.model.backbone.named_modules() if re.fullmatch(r'res\d\.\d+', n)]
assert len(names) == len(inter['blocks']), (len(names), len(inter['blocks']))
for rep in range(reps):
    for p in m.model.parameters():
        p.grad = None
    cascade_rcnn.DECISION_LOG = log = []
    got = training.losses(m.model, img.cuda(), gt.cuda(), cls.cuda(), choose=training.first_choice, config=cfg, proposals=inter['proposals'].float().cuda())
    cascade_rcnn.DECISION_LOG = None
    sum(got.values()).backward()
    assert len(log) == len(names)
    rel = {}
    for n, p in m.model.named_parameters():
        if p.requires_grad and p.grad is not None and float(rp[n].grad.abs().max()) > 0:
            rel[n] = float((p.grad.double().cpu() - rp[n].grad.double()).abs().max()) / float(rp[n].grad.abs().max())
    print('== run %d: worst tensor %.2e, median %.2e, tensors above 4e-3: %d' % (rep, max(rel.values()), sorted(rel.values())[len(rel) // 2],
                                                                                sum(1 for v in rel.values() if v > 4e-3)))
    for i, name in enumerate(names):
        d32 = R.block_decisions(*log[i])
        d64 = inter['blocks'][i]
        nr = [int((d32[k] != d64[k]).sum()) for k in ('relu1', 'relu2', 'relu3')]
        nc = 0 if d64['cells'] is None else int((d32['cells'] != d64['cells']).sum())
        worst = max([v for n, v in rel.items() if n.startswith('backbone.' + name + '.')] or [0.0])
        if sum(nr) or nc or worst > 2e-3:
            print('  %-8s relu flips conv1 / conv2 / out: %3d %3d %3d   cell flips: %3d   worst tensor of the block %.2e' % (name, nr[0], nr[1], nr[2], nc, worst))
