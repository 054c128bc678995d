"""SORT on the GPU - mirrors the public names of /root/reference/tracking/sort/sort.py.

``Sort(max_age, min_hits).update(dets, iou_threshold)`` keeps the reference signature (sort.py:234,244) and
return value; the tracker state (Kalman x/P per track, list order, counters) is resident in device memory and
every call runs the single-wavefront HIP kernel behind ``wt_sort_update_host`` (include/waymotrack.h).
``KalmanBoxTracker.count`` is the process-global ID counter of sort.py:86.
"""
import ctypes as C

import numpy as np

from ... import _lib


class _Counter(object):
    """KalmanBoxTracker.count twin backed by wt_idctr (shared by every Sort of the process)."""

    def __init__(self):
        self._h = None

    def handle(self):
        if self._h is None:
            self._h = C.c_void_p(_lib.lib().wt_idctr_create(C.c_int64(0)))
        return self._h

    @property
    def value(self):
        return int(_lib.lib().wt_idctr_get(self.handle()))

    @value.setter
    def value(self, v):
        _lib.lib().wt_idctr_set(self.handle(), C.c_int64(int(v)))


class _CountDescriptor(object):
    def __get__(self, obj, owner):
        return owner._counter.value

    def __set__(self, obj, v):        # instance assignment (class assignment is handled by the metaclass)
        type(obj)._counter.value = v


class _KBTMeta(type):
    def __setattr__(cls, name, value):
        if name == 'count':
            cls._counter.value = value
        else:
            super().__setattr__(name, value)


class KalmanBoxTracker(object, metaclass=_KBTMeta):
    """Only the global ``count`` attribute survives on the host (sort.py:86); tracks live on the GPU."""
    _counter = _Counter()
    count = _CountDescriptor()


def linear_assignment(X):
    """sklearn.utils.linear_assignment_.linear_assignment (0.22.2) on a float32 cost matrix (sort.py:26,206)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    if X.ndim != 2:
        X = np.atleast_2d(X)
    n, m = X.shape
    pairs = np.zeros((min(n, m) + 1, 2), dtype=np.int32)
    k = C.c_int(0)
    _lib.check(_lib.lib().wt_linear_assignment_f32_host(_lib.ptr(X), C.c_int(n), C.c_int(m), _lib.ptr(pairs), C.byref(k)),
               'wt_linear_assignment_f32_host')
    return pairs[:k.value].astype(np.int64)


def _associate(dets, trks, iou_threshold):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    dets = dets.reshape(-1, 5) if dets.size and dets.shape[-1] == 5 else np.ascontiguousarray(
        np.concatenate([dets.reshape(-1, 4), np.zeros((dets.size // 4, 1), np.float32)], axis=1) if dets.size else
        np.zeros((0, 5), np.float32))
    trks = np.ascontiguousarray(np.asarray(trks, dtype=np.float64).reshape(len(trks), -1)[:, :4]) if len(trks) else \
        np.zeros((0, 4), np.float64)
    n, t = len(dets), len(trks)
    matches = np.zeros((min(n, t) + 1, 2), np.int32)
    ud = np.zeros(n + 1, np.int32)
    ut = np.zeros(t + 1, np.int32)
    nm, nud, nut = C.c_int(0), C.c_int(0), C.c_int(0)
    _lib.check(_lib.lib().wt_associate_host(_lib.ptr(dets), C.c_int(n), _lib.ptr(trks), C.c_int(t),
                                            C.c_double(iou_threshold), _lib.ptr(matches), C.byref(nm), _lib.ptr(ud),
                                            C.byref(nud), _lib.ptr(ut), C.byref(nut)), 'wt_associate_host')
    return matches[:nm.value].astype(np.int64), ud[:nud.value].astype(np.int64), ut[:nut.value].astype(np.int64)


def associate_detections_to_trackers(detections, trackers, iou_threshold=0.3):
    """sort.py:193-230: returns (matches (K,2), unmatched_detections, unmatched_trackers)."""
    return _associate(detections, trackers, iou_threshold)


class Sort(object):
    def __init__(self, max_age=1, min_hits=3):
        """sort.py:234-242"""
        self.max_age = max_age
        self.min_hits = min_hits
        self.frame_count = 0
        self.confidence_factor = 0.1
        h = C.c_void_p()
        _lib.check(_lib.lib().wt_sort_create(C.c_int(max_age), C.c_int(min_hits), KalmanBoxTracker._counter.handle(),
                                             C.byref(h)), 'wt_sort_create')
        self._h = h
        self._n_tracks = 0
        self._borrowed = False

    @classmethod
    def _borrow(cls, handle, max_age, min_hits):
        """View of a Sort owned by a wt_mct (MultiClassTrackerSort): same methods, never destroys the handle."""
        self = cls.__new__(cls)
        self.max_age, self.min_hits, self.frame_count, self.confidence_factor = max_age, min_hits, 0, 0.1
        self._h = C.c_void_p(handle)
        self._n_tracks = 0
        self._borrowed = True
        return self

    def update(self, dets, iou_threshold):
        """sort.py:244-296: dets (N,5) [x1,y1,x2,y2,score] (or empty) -> (K,6) [x1,y1,x2,y2,id+1,confidence]."""
        dets = np.ascontiguousarray(dets, dtype=np.float32).reshape(-1, 5)
        self.frame_count += 1
        cap = self._n_tracks + len(dets) + 8
        out = np.zeros((cap, 6), dtype=np.float64)
        k = C.c_int(0)
        _lib.check(_lib.lib().wt_sort_update_host(self._h, _lib.ptr(dets), C.c_int(len(dets)), C.c_double(iou_threshold),
                                                  _lib.ptr(out), C.c_int(cap), C.byref(k)), 'wt_sort_update_host')
        self._n_tracks = int(_lib.lib().wt_sort_num_tracks(self._h))
        if k.value > 0:
            return out[:k.value].copy()
        return np.empty((0, 6))

    def state(self):
        """Debug hook: (ids, x (n,7), P (n,49)) of the live tracks in list order."""
        cap = self._n_tracks + 4096
        ids = np.zeros(cap, np.int64)
        x = np.zeros((cap, 7))
        P = np.zeros((cap, 49))
        n = C.c_int(0)
        _lib.check(_lib.lib().wt_sort_state_host(self._h, C.c_int(cap), _lib.ptr(ids), _lib.ptr(x), _lib.ptr(P), C.byref(n)),
                   'wt_sort_state_host')
        return ids[:n.value].copy(), x[:n.value].copy(), P[:n.value].copy()

    def __del__(self):
        h = getattr(self, '_h', None)
        if h and not getattr(self, '_borrowed', False):
            try:
                _lib.lib().wt_sort_destroy(h)
            except Exception:
                pass
            self._h = None
