"""Offset statistics of the random-init bench model's deformable layers on a synthetic frame (tools only): std of the learned
offsets and the share of samples beyond the persistent kernel's 2-px halo, per layer; which kernel each layer calibrated to."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline
from waymo_2d_tracking_amd.detnet.nn import ops, cascade_rcnn

stats = []
orig = ops.deform_conv3x3


def spy(x, offset, *a, **kw):
    if offset is not None:
        stats.append((tuple(x.shape), float(offset.std().item()), ops.far_offset_share(offset), bool(kw.get('far_offsets', False))))
    return orig(x, offset, *a, **kw)


pipe = DetectTrackPipeline(5, 2, use_graph=False)
ops.deform_conv3x3 = spy
cascade_rcnn.ops.deform_conv3x3 = spy
with torch.no_grad():
    pipe.detect_frame(0, 0, 0)
torch.cuda.synchronize()
for i, (shape, std, far, hint) in enumerate(stats):
    print('layer %2d  x %-22s offset std %.3f px  far share %.4f  far-kernel hint %s' % (i, shape, std, far, hint))
