# -*- coding: utf-8 -*-
"""Box utilities of the hot path - mirrors /root/reference/detnet/utils/box_utils.py (bbox branch only).

``nms`` keeps the reference signature (box_utils.py:307) and return convention; the greedy loop runs in the
HIP kernels behind ``wt_softnms_f64_host`` / ``wt_hardnms_f64_host`` (include/waymotrack.h).  The rotated-box
(shapely) branch and the SSD encode/decode helpers belong to other model families and are not provided.
"""
import ctypes as C

import numpy as np
import torch

from ... import _lib


def point_form(boxes):
    """box_utils.py:32-35: (cx, cy, w, h) -> (xmin, ymin, xmax, ymax)."""
    center = boxes[..., :2]
    half_wh = boxes[..., 2:4] * 0.5
    return torch.cat((center - half_wh, center + half_wh), -1)


def center_size(boxes):
    """box_utils.py:57-69: (xmin, ymin, xmax, ymax) -> (cx, cy, w, h)."""
    xy0 = boxes[:, :2]
    xy1 = boxes[:, 2:4]
    return torch.cat(((xy0 + xy1) * 0.5, xy1 - xy0), dim=1)


def nms(boxes, scores, overlap=0.5, top_k=0, soft=False, conf_thresh=0, soft_nms_cut=1):
    """box_utils.py:307-395.  boxes (n,4) xyxy tensor, scores (n) tensor.

    Returns (keep, scores): soft -> (list of int, float tensor of decayed scores); hard -> (LongTensor, scores[keep]),
    both in descending original-score order, like the reference."""
    if boxes.size(-1) == 8:
        raise NotImplementedError('rotated boxes (nms_rboxes) are outside the Cascade R-CNN hot path')
    b = np.ascontiguousarray(boxes.detach().cpu().numpy(), dtype=np.float64).reshape(-1, 4)
    s = np.ascontiguousarray(scores.detach().cpu().numpy(), dtype=np.float64).reshape(-1)
    n = len(s)
    keep = np.zeros(n + 1, np.int64)
    out = np.zeros(n + 1, np.float64)
    k = C.c_int(0)
    if soft:
        _lib.check(_lib.lib().wt_softnms_f64_host(_lib.ptr(b), _lib.ptr(s), C.c_int(n), C.c_double(overlap),
                                                  C.c_double(soft_nms_cut), C.c_double(conf_thresh), C.c_int(top_k),
                                                  _lib.ptr(keep), _lib.ptr(out), C.byref(k)), 'wt_softnms_f64_host')
        return keep[:k.value].tolist(), scores.new_tensor(out[:k.value])
    _lib.check(_lib.lib().wt_hardnms_f64_host(_lib.ptr(b), _lib.ptr(s), C.c_int(n), C.c_double(overlap), C.c_int(top_k),
                                              _lib.ptr(keep), _lib.ptr(out), C.byref(k)), 'wt_hardnms_f64_host')
    keep_t = torch.from_numpy(keep[:k.value].copy())
    return keep_t, scores[keep_t]
