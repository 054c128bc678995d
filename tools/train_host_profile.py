"""Host side of one config-5 training step (bench.py --stage train): is the step CPU-bound, and where does the Python time go?

    python tools/train_host_profile.py [--steps 3]

Prints per step: wall time with a device synchronisation at the end, the time at which step() RETURNED (all launches enqueued) and the time the
host spent blocked in its synchronisation points, then cProfile's top functions by own time over the profiled steps."""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.nn as nn

from waymo_2d_tracking_amd.detnet.nn import training
from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
from waymo_2d_tracking_amd.tuning import enable_gemm_tuning


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--top', type=int, default=45)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    enable_gemm_tuning()
    dev = torch.device('cuda')
    det = Detectron2Det(seed=0).to(dev).train()
    params = training.set_trainable(det.model)
    opt = torch.optim.SGD(params, lr=0.002, momentum=0.9, weight_decay=1e-4, fused=True)
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (1, 3, 886, 1280), generator=g).float().to(dev)
    n = 30
    wh = torch.rand((n, 2), generator=g) * 280 + 20
    xy = torch.rand((n, 2), generator=g) * torch.tensor([1280 - 300.0, 886 - 300.0])
    boxes = torch.cat((xy, xy + wh), 1).to(dev)
    classes = torch.randint(0, 4, (n,), generator=g).to(dev)
    marks = {}

    def step():
        opt.zero_grad(set_to_none=True)
        loss = sum(training.losses(det.model, img, boxes, classes).values())
        marks['fwd'] = time.perf_counter()
        loss.backward()
        marks['bwd'] = time.perf_counter()
        torch.nn.utils.clip_grad_norm_(params, 35.0)
        opt.step()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    for i in range(args.steps):
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('step %d: wall %.1f ms; forward enqueued at %.1f, backward at %.1f, step() returned at %.1f ms (the device then ran %.1f ms more)'
              % (i, (t2 - t0) * 1e3, (marks['fwd'] - t0) * 1e3, (marks['bwd'] - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for i in range(args.steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(args.top)
    print('cProfile over %d steps (own time; the profiler itself slows Python ~2x):' % args.steps)
    print('\n'.join(l[:200] for l in s.getvalue().splitlines()[:args.top + 12]))


if __name__ == '__main__':
    main()
