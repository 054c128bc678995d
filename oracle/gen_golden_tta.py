"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/tta_g7.npz by importing the REFERENCE's TTA operators
(/root/reference/detnet/nn/tta.py: ResizeTTA :179-190, HFlipTTA :147-156, VFlipTTA :158-167, SequentialTTA :107-117,
TTA :228-267) in this container (python3.10, torch CPU float32) - SURVEY fixture G7, row a21.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_tta.py

Only input/output DATA is written (tiny image tensors, transformed tensors, box lists); no reference source is copied.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden_ensemble import GOLDEN, import_reference  # noqa: E402  (same stubbing harness)


class FakeDetector(object):
    """Stands in for Detectron2Det: predict returns fixed per-class (n,5) [score,cx,cy,w,h] arrays and records its input."""

    def __init__(self, dets):
        self.dets = dets
        self.seen = []

    def predict(self, x):
        self.seen.append(x.clone())
        return [[d.copy() for d in self.dets] for _ in range(x.shape[0])]


def main():
    import torch
    torch.set_num_threads(1)
    import_reference()
    from detnet.nn import tta as T
    rng = np.random.default_rng(777)
    out = {}
    cases = [('x1.5,hflip', 13, 18), ('x1.5,hflip', 32, 48), ('x2,vflip', 9, 7), ('hflip', 8, 12), ('x0.5', 16, 20),
             ('x1.5', 11, 10), ('orig', 6, 5), ('x1.5,hflip,vflip', 10, 14)]
    for ci, (spec, h, w) in enumerate(cases):
        x = torch.from_numpy(rng.integers(0, 256, (2, 3, h, w)).astype(np.float32))
        dets = [np.concatenate((rng.uniform(0.05, 1, (n, 1)), rng.uniform(0.1, 0.9, (n, 2)), rng.uniform(0.02, 0.2, (n, 2))), 1)
                for n in (3, 0, 2, 1)]
        det = FakeDetector(dets)
        model = T.TTA(det, spec.split(','))
        y = model.predict(x)                                   # -> [image][class] (n,5)
        pre = det.seen[0]
        out['c%d_spec' % ci] = np.array(spec)
        out['c%d_x' % ci] = x.numpy()
        out['c%d_pre' % ci] = pre.numpy()
        for k in range(4):
            out['c%d_det%d' % (ci, k)] = dets[k]
            for b in range(2):
                out['c%d_post_b%d_k%d' % (ci, b, k)] = np.asarray(y[b][k])
    path = os.path.join(GOLDEN, 'tta_g7.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
