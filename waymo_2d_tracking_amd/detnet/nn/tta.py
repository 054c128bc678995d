"""Detection merging + test-time augmentation - mirrors /root/reference/detnet/nn/tta.py.

``nms_detections`` / ``merge_detections`` keep the reference signatures (tta.py:8, :22) on lists of
``(n_i, 5) [score, cx, cy, w, h]`` arrays; the arithmetic runs in the batched ensemble kernel
(``wt_ensemble_groups_host``, one wavefront per call here, thousands per call from detnet.ensemble).
Test-time augmentation (tta.py:69-267) is a parsed plan of image operations around ``detector.predict``; on the Waymo setting the
plan folds into the detector's fused pre-processing kernel.
"""
import ctypes as C

import numpy as np
import torch
from torch import nn

from ... import _lib


def _run_group(detections, method, thr, cut):
    """Centre-form group -> kernel (which expects left/top rows and applies ensemble.py:19-28 itself)."""
    dl = [np.asarray(d, dtype=np.float64).reshape(-1, 5) for d in detections]
    d = np.vstack(dl) if dl else np.zeros((0, 5))
    # the kernel computes cx = x + w/2 first; feed x = cx - w/2 would not round-trip bit-exactly, so the
    # centre-form entry uses the dedicated flag below (method + 16: rows are already centre-form)
    sizes = np.asarray([len(x) for x in dl], dtype=np.int32).reshape(1, -1)
    off = np.asarray([0, len(d)], dtype=np.int64)
    out = np.zeros((len(d) + 1, 5))
    cnt = np.zeros(2, np.int64)
    d = np.ascontiguousarray(d)
    _lib.check(_lib.lib().wt_ensemble_groups_host(_lib.ptr(d), _lib.ptr(off), _lib.ptr(sizes), C.c_int64(1),
                                                  C.c_int(len(dl)), C.c_int(method + 16), C.c_double(thr), C.c_double(cut),
                                                  _lib.ptr(out), _lib.ptr(cnt)), 'wt_ensemble_groups_host')
    return out[:int(cnt[0])].copy()


def nms_detections(detections, iou_thresh=0.5, soft=False, soft_nms_cut=1):
    """tta.py:8-19"""
    return _run_group(detections, 2 if soft else 1, iou_thresh, soft_nms_cut)


def merge_detections(detections, nms_thresh=0.5):
    """tta.py:22-66 (score-weighted box fusion).  Unlike the reference the inputs are NOT mutated."""
    return _run_group(detections, 0, nms_thresh, 1.0)


def parse_tta(data_aug):
    """`--tta` tokens -> the ordered list of image operations the reference's TTA.__init__ builds (tta.py:228-258):
    every `xS` resize in the given order, then hflip, then vflip ('orig' is the identity).  The brute / dflip / batch modes
    of the reference are not on the Waymo path (`--tta x1.5,hflip`, README.md:37)."""
    for unsupported in ('brute', 'dflip', 'batch'):
        if unsupported in data_aug:
            raise NotImplementedError('TTA mode %r is not part of the Waymo hot path' % unsupported)
    plan = [('resize', float(tok[1:])) for tok in data_aug if tok.startswith('x')]
    plan += [(flip,) for flip in ('hflip', 'vflip') if flip in data_aug]
    known = {'orig', 'hflip', 'vflip'}
    bad = [tok for tok in data_aug if tok not in known and not tok.startswith('x')]
    if bad:
        raise ValueError('unknown TTA token(s) %s' % bad)
    return plan


def apply_plan(x, plan):
    """Image side of the plan on a (B, 3, H, W) tensor: bilinear resize (align_corners False) / flips (tta.py:147-190)."""
    for op in plan:
        if op[0] == 'resize':
            x = torch.nn.functional.interpolate(x, scale_factor=op[1], mode='bilinear', align_corners=False)
        else:
            x = torch.flip(x, [3] if op[0] == 'hflip' else [2])
    return x


def undo_plan(detections, plan):
    """Box side: detections = [per image [per class ndarray (n, 5) [score, cx, cy, w, h] normalised]].  Boxes are normalised,
    so a resize needs no undo; a horizontal / vertical flip mirrors cx / cy (tta.py:150-155,167-172), in reverse plan order."""
    for op in reversed(plan):
        col = {'hflip': 1, 'vflip': 2}.get(op[0])
        if col is None:
            continue
        for per_class in detections:
            for boxes in per_class:
                boxes[..., col] = 1 - boxes[..., col]
    return detections


class TTA(nn.Module):
    """tta.py:228-267: detector.predict on the augmented image, detections mapped back."""

    def __init__(self, detector, data_aug):
        super().__init__()
        self.detector = detector
        self.plan = parse_tta(list(data_aug))

    def _fused_pre(self):
        """(scale, hflip, vflip) when the plan is at most one resize followed by flips (the Waymo setting --tta x1.5,hflip), which
        the detector folds into its pre-processing kernel; None otherwise."""
        if sum(op[0] == 'resize' for op in self.plan) > 1:
            return None
        scale = next((op[1] for op in self.plan if op[0] == 'resize'), 1.0)
        return float(scale), ('hflip',) in self.plan, ('vflip',) in self.plan

    def graph_lanes(self, lanes=2):
        """The streaming form of predict() (round 6): a GraphLanePredictor for the detector with this plan folded into its pre-processing, or None when
        the plan cannot be folded (more than one resize) / the detector has no static-shape pass; undo the plan on its results with undo_plan()."""
        fused = self._fused_pre() if hasattr(self.detector, 'predict_padded') and not getattr(self.detector, 'tta_min_sizes', None) else None
        if fused is None:
            return None
        from .detectron2_det import GraphLanePredictor
        return GraphLanePredictor(self.detector, *fused, lanes=lanes)

    def predict(self, x):
        fused = self._fused_pre() if hasattr(self.detector, 'predict_device') and torch.is_tensor(x) else None
        if fused is not None:
            # one HIP kernel: resize + flip + BGR + normalise + pad (tta.py:147-190 folded into the detector input)
            y = self.detector.predict(x, *fused)
        else:
            y = self.detector.predict(apply_plan(x, self.plan))
        return undo_plan(y, self.plan)
