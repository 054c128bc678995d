"""The detect -> SORT pipeline at the Waymo side-camera size 1920x886 (static-shape kernels / graph capture at another size); tools only."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline
pipe = DetectTrackPipeline(n_cameras=2, frames_per_camera=2, height=886, width=1920, seed=1, segment_frames=8, distinct_times=4)
import time
for _ in range(3):
    pipe.step(True)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(2):
    pipe.step(True)
torch.cuda.synchronize()
print('1920x886: %.1f frames/s' % (8 / (time.time() - t0)))
n_out, births = [int(v) for v in pipe.counts.cpu().tolist()]
print('last chunk: %d track rows, %d births' % (n_out, births))
