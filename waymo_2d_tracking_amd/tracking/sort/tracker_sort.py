#
# Multi class SORT tracker - mirrors /root/reference/tracking/sort/tracker_sort.py
#
import ctypes as C

import numpy as np

from ... import _lib
from .sort import KalmanBoxTracker, Sort


class MultiClassTrackerSort(object):

    def __init__(self, max_age=1, min_hits=0):
        """tracker_sort.py:12-20: one GPU-resident Sort per class, created on first sight (wt_mct_* of
        include/waymotrack.h holds them; ``self.trackers`` exposes them in first-seen order like the reference's dict)."""
        self.max_age = max_age
        self.min_hits = min_hits
        self.trackers = {}
        h = C.c_void_p()
        _lib.check(_lib.lib().wt_mct_create(C.c_int(max_age), C.c_int(min_hits), KalmanBoxTracker._counter.handle(),
                                            C.byref(h)), 'wt_mct_create')
        self._h = h
        self._rows = 0

    def track(self, detected_objects, iou_thresholds):
        """tracker_sort.py:22-51
        :param detected_objects: [[x1, y1, x2, y2, confidence, class_name], ...]
        :return: {class_name: ndarray (K,6) [x1, y1, x2, y2, object_id, confidence]}
        """
        lib = _lib.lib()
        dets = np.ascontiguousarray(np.asarray(detected_objects, dtype=np.float64).reshape(-1, 6))
        thr = np.ascontiguousarray(iou_thresholds, dtype=np.float64)
        n_cls_cap = len(self.trackers) + len(dets) + 1
        cap = self._rows + len(dets) + 8
        out = np.zeros((cap, 6), dtype=np.float64)
        classes = np.zeros(n_cls_cap, np.int32)
        counts = np.zeros(n_cls_cap, np.int32)
        k = C.c_int(0)
        _lib.check(lib.wt_mct_track_host(self._h, _lib.ptr(dets), C.c_int(len(dets)), _lib.ptr(thr), C.c_int(len(thr)),
                                         _lib.ptr(out), C.c_int(cap), _lib.ptr(classes), _lib.ptr(counts), C.c_int(n_cls_cap),
                                         C.byref(k)), 'wt_mct_track_host')
        all_tracked_objects = {}
        a = 0
        self._rows = 0
        for i in range(k.value):
            cls = int(classes[i])
            if cls not in self.trackers:
                self.trackers[cls] = Sort._borrow(lib.wt_mct_tracker(self._h, C.c_int(cls)), self.max_age, self.min_hits)
            trk = self.trackers[cls]
            trk.frame_count += 1
            trk._n_tracks = int(lib.wt_sort_num_tracks(trk._h))
            self._rows += trk._n_tracks
            n = int(counts[i])
            all_tracked_objects[cls] = out[a:a + n].copy() if n else np.empty((0, 6))
            a += n
        return all_tracked_objects

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            try:
                _lib.lib().wt_mct_destroy(h)
            except Exception:
                pass
            self._h = None
