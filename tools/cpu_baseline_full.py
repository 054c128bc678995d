"""The CPU port of the headline path timed ONCE at the real size (verdict r03, weak 8): one synthetic 1920x1280 frame through
oracle/detector_ref.py (PyTorch CPU fp32, the bench model's parameters) + oracle SORT on its detections, on this box's host cores.
Also the 640x448 frame bench.py times in its default run, so that the extrapolation it prints can be compared with a measurement.
    python tools/cpu_baseline_full.py > profiles/r04_cpu_baseline_full_size.json      (a few minutes of CPU time)"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import detector_ref as R
from oracle import oracle as O
from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det, detections_to_wire

O.build()
m = Detectron2Det(seed=0).eval()
cpu = m.model.cpu()
out = {'threads': torch.get_num_threads(), 'cpu_count': os.cpu_count()}
for (h, w) in ((448, 640), (1280, 1920)):
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (1, 3, h, w), generator=g).float()[:, [2, 1, 0]]
    t0 = time.perf_counter()
    with torch.no_grad():
        boxes, scores, classes = R.forward(cpu, img)
    t_det = time.perf_counter() - t0
    xywh, score, cat = detections_to_wire(boxes, scores, classes, w, h)
    n = xywh.shape[0]
    packed = dict(x=xywh[:, 0].numpy().copy(), y=xywh[:, 1].numpy().copy(), w=xywh[:, 2].numpy().copy(), h=xywh[:, 3].numpy().copy(),
                  score=score.numpy().copy(), category=cat.numpy().astype(np.int32), frame_det_offsets=np.array([0, n], np.int64),
                  stream_frame_offsets=np.array([0, 1], np.int64), clip_w=np.array([float(w)]), clip_h=np.array([float(h)]))
    t0 = time.perf_counter()
    O.track_streams(packed, 2, 0, [0.0] * 4, [0.01, 0.01, 1.0, 0.0])
    t_sort = time.perf_counter() - t0
    out['%dx%d' % (w, h)] = dict(detector_s=t_det, sort_s=t_sort, frames_per_s=1.0 / (t_det + t_sort), detections=int(n))
small, full = out['640x448'], out['1920x1280']
out['measured_time_ratio_full_over_small'] = full['detector_s'] / small['detector_s']
out['pixel_ratio'] = 1920 * 1280 / (640 * 448)
out['note'] = ('bench.py scales the 640x448 detector time by the pixel ratio (full_res_equivalent); the measured ratio says how far '
               'that extrapolation is from one real full-size frame on the same cores')
print(json.dumps(out, indent=1))
