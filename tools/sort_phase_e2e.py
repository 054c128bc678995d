"""Per-phase cycle split of the SORT kernel on the detections of the end-to-end pipeline (tools only).  Build with
    WD_HIPCC_FLAGS=-DWT_PHASE_TIMING python -m waymo_2d_tracking_amd.build --force   (or tools/build_variant.sh for det_deform_pp only)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from waymo_2d_tracking_amd import _lib
from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline

pipe = DetectTrackPipeline(5, 2, seed=0)
for _ in range(5):
    pipe.step(True)
torch.cuda.synchronize()
cat = pipe.category[:5].cpu().numpy().reshape(5, 5, 2, 100)
print('detections per (camera, frame) by class 1..4:', [int((cat[4] == c).sum() / 10) for c in (1, 2, 3, 4)])
counts = pipe.chunk_counts[:5].cpu().numpy()
print('rows / births per chunk:', counts.tolist())
out = (C.c_ulonglong * 12)()
if not hasattr(_lib.lib(), 'wt_debug_phase_cycles'):
    sys.exit('library was not built with -DWT_PHASE_TIMING')
_lib.lib().wt_debug_phase_cycles(out, 0)
v = np.array(list(out), dtype=np.float64)
names = ['predict', 'iou matrix', 'munkres', 'match filter', 'kalman update', 'births + emit + reap']
tot = v[:6].sum()
print('total cycles (lane 0 of every tracker wave, 5 chunks): %.3g = %.1f ms per tracker and chunk at 2.2 GHz' % (tot, tot / 2.2e9 * 1e3 / (20 * 5)))
for n, x in zip(names, v[:6]):
    print('%-22s %5.1f %%' % (n, 100 * x / tot))
for n, i in (('  munkres: step 1 + greedy stars', 6), ('  munkres: steps 3-5 (cover / prime / augment)', 8), ('  munkres: step 6 (adjust)', 7)):
    print('%-46s %5.1f %%' % (n, 100 * v[i] / tot))
