"""End-to-end detect -> SORT step used by bench.py (BASELINE.json metric) and by smoke().

One step = one 5-camera chunk: `frames_per_step` synthetic 1920x1280x3 uint8 frames already resident in HBM go
through the Cascade R-CNN X152-FPN detector one by one (batch 1, like the reference's --batch-size=1); their <= 100
detections per frame are converted on the device to the detection-JSON wire values (int box, 5-decimal score,
category) and written into the frame-slotted SoA layout of wt_track_streams_dev; then every (camera, class) tracker
runs as one wavefront of the persistent SORT kernel.  Nothing leaves the GPU inside the timed region.
"""
import ctypes as C
import os
import time

import numpy as np
import torch

from . import _lib
from .detnet.nn.detectron2_det import Detectron2Det, detections_to_wire
from .tracking.utils import make_params
from .tuning import enable_gemm_tuning

SLOTS = 100          # detectron2 TEST.DETECTIONS_PER_IMAGE (top-100, detectron2_det via fast_rcnn_inference)


class DetectTrackPipeline(object):
    def __init__(self, n_cameras=5, frames_per_camera=2, height=1280, width=1920, seed=0, device='cuda',
                 iou_threshold=(0.01, 0.01, 1.0, 0.0), score_threshold=(0.0, 0.0, 0.0, 0.0), max_age=2, min_hits=0, tta=''):
        self.dev = torch.device(device)
        # --tta x1.5,hflip (nn/tta.py:228-267): one pass on the enlarged, flipped image, folded into the pre-processing kernel
        self.tta_scale, self.tta_hflip = 1.0, False
        for aug in [a for a in tta.split(',') if a]:
            if aug.startswith('x'):
                self.tta_scale *= float(aug[1:])
            elif aug == 'hflip':
                self.tta_hflip = True
            elif aug != 'orig':
                raise ValueError('bench supports --tta orig / xS / hflip, got %r' % aug)
        torch.backends.cudnn.benchmark = True      # the reference's --cudnn-benchmark: let MIOpen pick its fastest conv
        enable_gemm_tuning()                       # ... and TunableOp its fastest library GEMM per 1x1-conv shape
        self.nc, self.fpc, self.h, self.w = n_cameras, frames_per_camera, height, width
        self.model = Detectron2Det(seed=seed).to(self.dev).eval()
        self.n_frames = n_cameras * frames_per_camera
        g = torch.Generator(device='cpu').manual_seed(seed)
        self.frames = torch.randint(0, 256, (self.n_frames, height, width, 3), generator=g, dtype=torch.uint8).to(self.dev)
        n = self.n_frames * SLOTS
        # two sets of detection slots: SORT over chunk s runs on its own stream while the detector fills the slots of
        # chunk s + 1 (the tracker kernel is one wave per (camera, class) - it would leave the other CUs idle)
        self._slots = []
        for _ in range(2):
            x = torch.zeros(n, dtype=torch.float64, device=self.dev)
            self._slots.append(dict(x=x, y=torch.zeros_like(x), wd=torch.zeros_like(x), ht=torch.zeros_like(x),
                                    score=torch.zeros_like(x), category=torch.zeros(n, dtype=torch.int32, device=self.dev)))
        self._cur = 0
        self._bind(0)
        self.track_stream = torch.cuda.Stream(device=self.dev)
        self._track_done = None
        self.frame_off = (torch.arange(self.n_frames + 1, dtype=torch.int64) * SLOTS).to(self.dev)
        self.stream_off = (torch.arange(n_cameras + 1, dtype=torch.int64) * frames_per_camera).to(self.dev)
        self.clip_w = torch.full((n_cameras,), float(width), dtype=torch.float64, device=self.dev)
        self.clip_h = torch.full((n_cameras,), float(height), dtype=torch.float64, device=self.dev)
        self.params, self._keep = make_params(max_age, min_hits, list(score_threshold), list(iou_threshold))
        self.lib = _lib.lib()
        ws = int(self.lib.wt_track_streams_workspace(C.c_int64(n), C.c_int64(self.n_frames), C.c_int32(n_cameras),
                                                     C.c_int64(SLOTS), C.byref(self.params)))
        self.ws = torch.empty(ws, dtype=torch.uint8, device=self.dev)
        self.out_frame = torch.empty(n + 1, dtype=torch.int64, device=self.dev)
        self.out_cat = torch.empty(n + 1, dtype=torch.int32, device=self.dev)
        self.out_bbox = torch.empty((n + 1, 4), dtype=torch.float64, device=self.dev)
        self.out_score = torch.empty(n + 1, dtype=torch.float64, device=self.dev)
        self.out_id = torch.empty(n + 1, dtype=torch.int64, device=self.dev)
        self.counts = torch.zeros(2, dtype=torch.int64, device=self.dev)
        self.n_dets_last = 0

    def _bind(self, i):
        self._cur = i
        for k, v in self._slots[i].items():
            setattr(self, k, v)

    def detect_frame(self, f):
        """Frame f (camera-major order) -> wire-format detections written into slots [f*100, f*100+100)."""
        # decoded uint8 HWC RGB frame -> fused pre-processing kernel (ToTensor(scaling=False) + BGR + normalise + pad)
        (boxes, scores, classes), = self.model.predict_device(self.frames[f:f + 1], self.tta_scale, self.tta_hflip)
        ho, wo = self.model.last_input_size
        if self.tta_hflip:                                                      # HFlipTTA.post_process: cx <- 1 - cx
            boxes = torch.stack((wo - boxes[:, 2], boxes[:, 1], wo - boxes[:, 0], boxes[:, 3]), dim=1)
        xywh, score, cat = detections_to_wire(boxes, scores, classes, wo, ho, self.w, self.h)
        k = xywh.shape[0]
        a = f * SLOTS
        self.category[a:a + SLOTS] = 0                                          # unused slots: category 0 = ignored
        self.x[a:a + k] = xywh[:, 0]; self.y[a:a + k] = xywh[:, 1]
        self.wd[a:a + k] = xywh[:, 2]; self.ht[a:a + k] = xywh[:, 3]
        self.score[a:a + k] = score
        self.category[a:a + k] = cat
        return k

    def track(self):
        p = lambda t: C.c_void_p(t.data_ptr())
        n = self.n_frames * SLOTS
        rc = self.lib.wt_track_streams_dev(
            C.c_int64(n), p(self.x), p(self.y), p(self.wd), p(self.ht), p(self.score), p(self.category),
            C.c_int64(self.n_frames), p(self.frame_off), C.c_int32(self.nc), p(self.stream_off), p(self.clip_w),
            p(self.clip_h), C.c_int64(SLOTS), C.byref(self.params), C.c_int64(0), p(self.out_frame), p(self.out_cat),
            p(self.out_bbox), p(self.out_score), p(self.out_id), p(self.counts), C.c_void_p(self.counts.data_ptr() + 8),
            p(self.ws), C.c_size_t(self.ws.numel()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, 'wt_track_streams_dev')

    def step(self, with_tracking=True):
        total = 0
        for f in range(self.n_frames):
            total += self.detect_frame(f)
        if with_tracking:
            main = torch.cuda.current_stream()
            filled = torch.cuda.Event()
            filled.record(main)
            with torch.cuda.stream(self.track_stream):
                self.track_stream.wait_event(filled)             # slots of this chunk are complete
                self.track()                                     # reads the slot set bound during detection
                self._track_done = torch.cuda.Event()
                self._track_done.record(self.track_stream)
            self._bind(1 - self._cur)                            # the next chunk's detections go to the other set
            # the set being re-bound was consumed by the track() of the previous step on the same (in-order) stream,
            # i.e. before the track() just queued; the detector may only overwrite it after that one has finished
            main.wait_event(self._prev_done) if getattr(self, '_prev_done', None) is not None else None
            self._prev_done = self._track_done
        self.n_dets_last = total
        return total


def _pmc_traffic(tag):
    """HBM bytes per launch of the roofline kernel from the committed PMC pass (profiles/r01_e2e_pmc_traffic.json,
    collected with tools/pmc_traffic.sh - counters cannot be read from inside the timed run); None if not recorded."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r01_e2e_pmc_traffic.json')
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    for name, v in rec.items():
        if tag.startswith(name):
            return v.get('traffic_bytes_per_launch')
    return None


def run(args, world, rank, timed_steps):
    from .detnet.nn import ops
    fps = max(1, args.frames_per_step // 5)
    pipe = DetectTrackPipeline(5, fps, seed=rank, tta=getattr(args, 'tta', '') or '')
    steps = args.steps or 3
    warmup = args.warmup if args.warmup is not None else 1
    track = args.stage == 'e2e'
    state = {'n': 0}

    def step():
        # HIP-event instrumentation of the dominant hand-written kernel only inside the timed region
        if state['n'] == warmup:
            ops.EVENT_LOG = []
        state['n'] += 1
        pipe.step(track)

    dt, ev_ms = timed_steps(world, step, steps, warmup)
    log, ops.EVENT_LOG = ops.EVENT_LOG or [], None
    n_out, births = [int(v) for v in pipe.counts.cpu().tolist()]
    frames = pipe.n_frames
    # roofline of the dominant hand-written kernel: the deformable-conv implicit GEMM (47 launches per frame);
    # achieved = algorithmic flops per launch / mean launch duration of the most frequent shape (res4, 36 of 47)
    by_shape = {}
    for tag, flops, e0, e1 in log:
        d = by_shape.setdefault(tag, [0, 0.0, flops])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
    roofline = None
    if by_shape:
        deform = {k: v for k, v in by_shape.items() if 'no offsets' not in k}
        tag = max(deform, key=lambda k: deform[k][1])
        cnt, ms, flops = deform[tag]
        ach = flops / (ms / cnt * 1e-3) / 1e12
        roofline = dict(bound='mfma', kernel=tag, achieved=ach, peak=157.3,
                        unit='TFLOP/s', frac=ach / 157.3, traffic=_pmc_traffic(tag), launches=cnt, avg_us=ms / cnt * 1e3,
                        flops_per_launch=flops,
                        all_shapes={k: dict(launches=v[0], avg_us=v[1] / v[0] * 1e3, tflops=v[2] / (v[1] / v[0] * 1e-3) / 1e12)
                                    for k, v in by_shape.items()})
    res = dict(value=frames * world * steps / dt, unit='frames/s', ms_per_step=1e3 * dt / steps, dtype='f32',
               workload='Cascade R-CNN X152-32x8d-FPN dconv (random-init, fp32, batch 1%s) on synthetic 1920x1280x3 frames'
                        ' -> top-100 detections/frame -> %s; %d cameras x %d frames per step per GPU'
                        % (', --tta ' + args.tta if getattr(args, 'tta', '') else '',
                           'SORT (max_age 2, min_hits 0, all boxes tracked)' if track else 'no tracking', 5, fps),
               roofline=roofline,
               extra=dict(frames_per_step=frames, dets_per_frame=pipe.n_dets_last / frames, track_rows=n_out, births=births))
    res['pipeline'] = pipe         # bench.py times the CPU port (oracle) against the same parameters
    return res, steps, warmup


def run_train(args, world, rank, timed_steps):
    """Config 5: Cascade R-CNN X152 dconv training fwd+bwd+SGD step, 886x1280 crop (train.py:37-47), batch 1 per
    GPU, DDP over RCCL when world > 1 (bucketed gradient all-reduce overlapped with backward)."""
    import torch.nn as nn
    from .detnet.nn import training
    from .detnet.nn.detectron2_det import Detectron2Det
    torch.backends.cudnn.benchmark = True
    enable_gemm_tuning()
    dev = torch.device('cuda')
    det = Detectron2Det(seed=0).to(dev).train()
    params = training.set_trainable(det.model)

    class Wrapper(nn.Module):
        def __init__(self, model):
            super().__init__()
            self.model = model

        def forward(self, image_bgr, boxes, classes):
            return sum(training.losses(self.model, image_bgr, boxes, classes).values())

    net = Wrapper(det.model)
    if world > 1:
        net = nn.parallel.DistributedDataParallel(net, device_ids=[torch.cuda.current_device()])
    opt = torch.optim.SGD(params, lr=0.002, momentum=0.9, weight_decay=1e-4)       # detnet/configs/detectron2.cfg
    g = torch.Generator().manual_seed(rank)
    img = torch.randint(0, 256, (1, 3, 886, 1280), generator=g).float().to(dev)
    n = 30
    wh = torch.rand((n, 2), generator=g) * 280 + 20
    xy = torch.rand((n, 2), generator=g) * torch.tensor([1280 - 300.0, 886 - 300.0])
    boxes = torch.cat((xy, xy + wh), 1).to(dev)
    classes = torch.randint(0, 4, (n,), generator=g).to(dev)
    last = {}

    def step():
        opt.zero_grad(set_to_none=True)
        loss = net(img, boxes, classes)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 35.0)
        opt.step()
        last['loss'] = loss.detach()

    steps = args.steps or 3
    warmup = args.warmup if args.warmup is not None else 1
    dt, ev_ms = timed_steps(world, step, steps, warmup)
    res = dict(value=world * steps / dt, unit='images/s', ms_per_step=1e3 * dt / steps, dtype='f32',
               workload='Cascade R-CNN X152-32x8d-FPN dconv training step (fwd+bwd+SGD), 886x1280 crop, batch 1/GPU, '
                        '30 synthetic gt boxes, FREEZE_AT 2, FrozenBN, %s' % ('DDP x%d over RCCL' % world if world > 1 else 'single GPU'),
               extra=dict(loss=float(last['loss']), trainable_params=sum(p.numel() for p in params)))
    return res, steps, warmup
