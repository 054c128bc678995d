"""Where do the 4.8 % between the detector alone and detect + SORT go?  (tools only)
 a) detector alone   b) e2e as benchmarked   c) e2e with every detection filtered out (the whole tracking call chain - memsets,
 9 launches, events, stream waits - but an empty SORT kernel)   d) e2e with SORT on the MAIN stream (no overlap)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline


def fps(pipe, track, steps=4):
    for _ in range(2):
        pipe.step(track)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.step(track)
    torch.cuda.synchronize()
    return pipe.n_frames * steps / (time.perf_counter() - t0)


pipe = DetectTrackPipeline(5, 2, seed=0)
print('a) detector alone                          %.2f frames/s' % fps(pipe, False), flush=True)
print('b) detect + SORT (bench configuration)     %.2f frames/s' % fps(pipe, True), flush=True)
empty = DetectTrackPipeline(5, 2, seed=0, model=pipe.model, score_threshold=(2.0, 2.0, 2.0, 2.0))
print('c) detect + SORT call chain, no detections %.2f frames/s' % fps(empty, True), flush=True)
same = DetectTrackPipeline(5, 2, seed=0, model=pipe.model)
same.track_stream = torch.cuda.current_stream()
print('d) detect + SORT on the main stream        %.2f frames/s' % fps(same, True), flush=True)
