"""Throughput of the drop-in inference CLI (waymo_2d_tracking_amd/detnet/inference.py: image folder -> detection JSON) on synthetic 1920x1280 / 1920x886 JPEG
frames with a random-init model file, next to bench.py --stage detect.    python tools/cli_throughput.py [n_images] [extra CLI flags ...]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
from waymo_2d_tracking_amd.detnet import nn as detnn, inference as I

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
extra = sys.argv[2:]
tmp = tempfile.mkdtemp(prefix='wt_cli_')
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:1280, 0:1920]
base = np.clip(128 + 100 * np.sin(xx[..., None] / 7.0 + np.arange(3)) * np.cos(yy[..., None] / 9.0) + rng.normal(0, 12, (1280, 1920, 3)), 0, 255).astype(np.uint8)
for i in range(n):
    cam = ('FRONT', 'FRONT_LEFT', 'FRONT_RIGHT', 'SIDE_LEFT', 'SIDE_RIGHT')[i % 5]
    d = os.path.join(tmp, 'images', 'seg%02d' % (i // 50), str(1000 + (i % 50) // 5))
    os.makedirs(d, exist_ok=True)
    h = 1280 if cam.startswith('FRONT') else 886
    Image.fromarray(np.roll(base, (3 * i, 5 * i), (0, 1))[:h]).save(os.path.join(d, cam + '.jpg'), quality=90)
net = detnn.create('detectron2:Misc/cascade_mask_rcnn_X_152_32x8d_FPN_IN5k_gn_dconv.yaml', ['vehicle', 'pedestrian', 'sign', 'cyclist'], pretrained=None,
                   freeze_pretrained=2, frozen_bn=True, seed=0)
model = os.path.join(tmp, 'random.model')
net.save(model)
del net
argv = ['-m', model, '-i', os.path.join(tmp, 'images'), '--export', os.path.join(tmp, 'sub.json'), '--batch-size=1'] + extra
I.main(argv)                                   # warm-up run (library kernel selection, captures)
torch.cuda.synchronize()
t0 = time.time()
rows = I.main(argv)
torch.cuda.synchronize()
dt = time.time() - t0
print('inference CLI %s: %d images in %.2f s = %.1f images/s (whole run incl. model load, decode, export)' % (' '.join(extra), n, dt, n / dt))
