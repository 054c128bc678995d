"""Gradient error of the HIP training graph against the float64 CPU restatement, every trainable tensor: max |diff| / max |ref| (the quantity
tests/test_gpu_detector.py::test_training_losses_and_gradients_vs_f64_restatement bounds by 2e-3).  A/B: WD_SPLIT_TRAIN=0|1 python tools/train_grad_check.py"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import detector_ref as R
from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
from waymo_2d_tracking_amd.detnet.nn import training

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
got = training.losses(m.model, img.cuda(), gt.cuda(), cls.cuda(), choose=training.first_choice, config=cfg, proposals=inter['proposals'].float().cuda())
sum(got.values()).backward()
print('WD_SPLIT_TRAIN=%s  losses rel err: %s' % (os.environ.get('WD_SPLIT_TRAIN', '1'),
                                                 {k: '%.1e' % (abs(float(got[k]) - float(ref[k])) / max(1.0, abs(float(ref[k])))) for k in sorted(ref)}))
gp, rp = dict(m.model.named_parameters()), dict(cpu.named_parameters())
rows = []
for n, p in gp.items():
    if p.requires_grad and p.grad is not None and rp[n].grad is not None:
        a, b = p.grad.double().cpu(), rp[n].grad.double()
        s = float(b.abs().max())
        if s > 0:
            rows.append((float((a - b).abs().max()) / s, n))
rows.sort(reverse=True)
print('worst 12 of %d tensors (max |diff| / max |ref|):' % len(rows))
for e, n in rows[:12]:
    print('   %.2e  %s' % (e, n))
import statistics
print('median %.2e' % statistics.median(e for e, _ in rows))

# REPEATS=n: the GPU side again (same inputs): worst tensors of every run against float64 AND against the first GPU run (run-to-run noise of the float
# atomics; a tensor that moves by more than the atomics explain points at a discrete event: a bilinear sample crossing a pixel boundary, or a bug)
reps = int(os.environ.get('REPEATS', '0'))
first = {n: p.grad.detach().clone() for n, p in gp.items() if p.requires_grad and p.grad is not None}
for r in range(reps):
    for p in gp.values():
        p.grad = None
    got = training.losses(m.model, img.cuda(), gt.cuda(), cls.cuda(), choose=training.first_choice, config=cfg, proposals=inter['proposals'].float().cuda())
    sum(got.values()).backward()
    vs64, vs0, l2 = [], [], []
    for n, p in gp.items():
        if n in first and rp[n].grad is not None and float(rp[n].grad.abs().max()) > 0:
            s = float(rp[n].grad.abs().max())
            d = p.grad.double().cpu() - rp[n].grad.double()
            vs64.append((float(d.abs().max()) / s, n))
            vs0.append((float((p.grad - first[n]).abs().max()) / s, n))
            l2.append((float(d.norm() / rp[n].grad.double().norm()), n))
    vs64.sort(reverse=True)
    vs0.sort(reverse=True)
    l2.sort(reverse=True)
    print('run %d  relative L2 vs f64, worst: %s; median %.1e; tensors with max-norm error > 4e-3: %d' % (
        r + 1, ', '.join('%.1e %s' % (e, n.replace('backbone.', '')) for e, n in l2[:4]), l2[len(l2) // 2][0], sum(1 for e, _ in vs64 if e > 4e-3)))
    print('run %d  vs f64: %s' % (r + 1, ', '.join('%.1e %s' % (e, n.replace('backbone.', '')) for e, n in vs64[:4])))
    print('        vs run 0: %s' % ', '.join('%.1e %s' % (e, n.replace('backbone.', '')) for e, n in vs0[:4]))
