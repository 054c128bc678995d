"""Detection merging + test-time augmentation - mirrors /root/reference/detnet/nn/tta.py.

``nms_detections`` / ``merge_detections`` keep the reference signatures (tta.py:8, :22) on lists of
``(n_i, 5) [score, cx, cy, w, h]`` arrays; the arithmetic runs in the batched ensemble kernel
(``wt_ensemble_groups_host``, one wavefront per call here, thousands per call from detnet.ensemble).
The TTA operator algebra (tta.py:69-267) is host-side tensor plumbing around ``detector.predict``.
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


class OpTTA(object):
    def pre_process(self, x):
        raise NotImplementedError

    def post_process(self, y):
        raise NotImplementedError


class Compose(OpTTA, list):
    pass


class SequentialTTA(Compose):
    """tta.py:107-116"""

    def pre_process(self, x):
        for tta in self:
            x = tta.pre_process(x)
        return x

    def post_process(self, y):
        for tta in self[::-1]:
            y = tta.post_process(y)
        return y


class ParallelTTA(Compose):
    """tta.py:119-136"""

    def __init__(self, ttas, merge_func=merge_detections):
        super().__init__(ttas)
        self.merge_func = merge_func

    def pre_process(self, x):
        return sum([tta.pre_process(x) for tta in self], [])

    def post_process(self, y):
        l = len(y) // len(self)
        Y = [tta.post_process(y[i * l:i * l + l]) for i, tta in enumerate(self)]
        return [[[self.merge_func(b) for b in zip(*ci)] for ci in zip(*yi)] for yi in zip(*Y)]


class OrigTTA(OpTTA):
    def pre_process(self, x):
        return x

    def post_process(self, y):
        return y


class HFlipTTA(OpTTA):
    """tta.py:147-156"""

    def pre_process(self, x):
        return [torch.flip(xi, [3]) for xi in x]

    def post_process(self, y):
        for yi in y:
            for d in yi:
                for c in d:
                    c[..., 1] = 1 - c[..., 1]
        return y


class VFlipTTA(OpTTA):
    def pre_process(self, x):
        return [torch.flip(xi, [2]) for xi in x]

    def post_process(self, y):
        for yi in y:
            for d in yi:
                for c in d:
                    c[..., 2] = 1 - c[..., 2]
        return y


class ResizeTTA(OpTTA):
    """tta.py:179-190"""

    def __init__(self, scale_factor):
        self.scale_factor = scale_factor

    def __repr__(self):
        return self.__class__.__name__ + f'(scale_factor={self.scale_factor})'

    def pre_process(self, x):
        return [torch.nn.functional.interpolate(xi, scale_factor=self.scale_factor, mode='bilinear', align_corners=False)
                for xi in x]

    def post_process(self, y):
        return y


class TTA(nn.Module):
    """tta.py:228-267 (orig / xS / hflip / vflip; the unused brute / dflip / batch modes are not provided)."""

    def __init__(self, detector, data_aug):
        super().__init__()
        self.detector = detector
        ttas = []
        if 'orig' in data_aug:
            ttas.append(OrigTTA())
        for aug in data_aug:
            if aug.startswith('x'):
                ttas.append(ResizeTTA(float(aug[1:])))
        for unsupported in ('brute', 'dflip', 'batch'):
            if unsupported in data_aug:
                raise NotImplementedError('TTA mode %r is not part of the Waymo hot path' % unsupported)
        if 'hflip' in data_aug:
            ttas.append(HFlipTTA())
        if 'vflip' in data_aug:
            ttas.append(VFlipTTA())
        self.tta = SequentialTTA(ttas)

    def _fused_pre(self):
        """(scale, hflip, vflip) when the sequence is resize / flips only (the Waymo setting --tta x1.5,hflip) and the
        detector can fold them into its pre-processing kernel; None otherwise."""
        scale, hflip, vflip = 1.0, False, False
        for t in self.tta:
            if isinstance(t, OrigTTA):
                continue
            if isinstance(t, ResizeTTA) and not (hflip or vflip):     # resize must come before the flips to commute
                scale *= float(t.scale_factor)
            elif isinstance(t, HFlipTTA):
                hflip = not hflip
            elif isinstance(t, VFlipTTA):
                vflip = not vflip
            else:
                return None
        n_resize = sum(isinstance(t, ResizeTTA) for t in self.tta)
        return (scale, hflip, vflip) if n_resize <= 1 else None

    def predict(self, x):
        fused = self._fused_pre() if hasattr(self.detector, 'predict_device') and torch.is_tensor(x) else None
        if fused is not None:
            # one HIP kernel: resize + flip + BGR + normalise + pad (tta.py:147-190 folded into the detector input)
            Y = [self.detector.predict(x, *fused)]
        else:
            X = self.tta.pre_process([x])
            Y = [self.detector.predict(xi) for xi in X]
        y = self.tta.post_process(Y)
        return y[0]
