"""Device-resident front-ends of the C ABI (the *_dev entry points of include/waymotrack.h).

torch is used only as plumbing: device memory (tensors), the current HIP stream and, for multi-GPU runs,
torch.distributed.  Everything timed in bench.py goes through these classes; inputs and outputs stay in HBM.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .tracking.utils import make_params


def _dp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class DeviceTracker(object):
    """wt_track_streams_dev on tensors already in HBM (layout: include/waymotrack.h, wt_track_streams_host)."""

    def __init__(self, packed, iou_threshold, max_age, min_hits, score_threshold=None, device='cuda'):
        self.lib = _lib.lib()
        dev = torch.device(device)
        f64 = lambda k: torch.from_numpy(np.ascontiguousarray(packed[k], dtype=np.float64)).to(dev)
        self.x, self.y, self.w, self.h, self.score = f64('x'), f64('y'), f64('w'), f64('h'), f64('score')
        self.category = torch.from_numpy(np.ascontiguousarray(packed['category'], dtype=np.int32)).to(dev)
        self.frame_off = torch.from_numpy(np.ascontiguousarray(packed['frame_det_offsets'], dtype=np.int64)).to(dev)
        self.stream_off = torch.from_numpy(np.ascontiguousarray(packed['stream_frame_offsets'], dtype=np.int64)).to(dev)
        self.clip_w, self.clip_h = f64('clip_w'), f64('clip_h')
        self.n_dets = int(packed['x'].size)
        self.n_frames = int(packed['frame_det_offsets'].size - 1)
        self.n_streams = int(packed['stream_frame_offsets'].size - 1)
        fo = np.asarray(packed['frame_det_offsets'])
        self.max_frame = int(np.diff(fo).max()) if self.n_frames else 0
        self.params, self._keep = make_params(max_age, min_hits, score_threshold, iou_threshold)
        ws = self.lib.wt_track_streams_workspace(C.c_int64(self.n_dets), C.c_int64(self.n_frames),
                                                 C.c_int32(self.n_streams), C.c_int64(self.max_frame),
                                                 C.byref(self.params))
        if ws == 0:
            raise _lib.WaymoTrackError('wt_track_streams_workspace: ' + self.lib.wt_last_error().decode())
        self.ws_bytes = int(ws)
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        n = self.n_dets + 1
        self.out_frame = torch.empty(n, dtype=torch.int64, device=dev)
        self.out_category = torch.empty(n, dtype=torch.int32, device=dev)
        self.out_bbox = torch.empty((n, 4), dtype=torch.float64, device=dev)
        self.out_score = torch.empty(n, dtype=torch.float64, device=dev)
        self.out_id = torch.empty(n, dtype=torch.int64, device=dev)
        self.counts = torch.zeros(2, dtype=torch.int64, device=dev)

    def run(self, id_base=0):
        """Enqueue one tracking pass on the current stream (no host synchronisation)."""
        rc = self.lib.wt_track_streams_dev(
            C.c_int64(self.n_dets), _dp(self.x), _dp(self.y), _dp(self.w), _dp(self.h), _dp(self.score),
            _dp(self.category), C.c_int64(self.n_frames), _dp(self.frame_off), C.c_int32(self.n_streams),
            _dp(self.stream_off), _dp(self.clip_w), _dp(self.clip_h), C.c_int64(self.max_frame), C.byref(self.params),
            C.c_int64(id_base), _dp(self.out_frame), _dp(self.out_category), _dp(self.out_bbox), _dp(self.out_score),
            _dp(self.out_id), _dp(self.counts), C.c_void_p(self.counts.data_ptr() + 8), _dp(self.workspace),
            C.c_size_t(self.ws_bytes), _stream())
        _lib.check(rc, 'wt_track_streams_dev')

    def results(self):
        """Synchronise and fetch (dict of numpy arrays, n_births)."""
        n_out, births = [int(v) for v in self.counts.cpu().tolist()]
        if n_out < 0:
            raise _lib.WaymoTrackError('SORT kernel status %d' % -n_out)
        return dict(frame=self.out_frame[:n_out].cpu().numpy(), category=self.out_category[:n_out].cpu().numpy(),
                    bbox=self.out_bbox[:n_out].cpu().numpy(), score=self.out_score[:n_out].cpu().numpy(),
                    object_id=self.out_id[:n_out].cpu().numpy()), births


class DeviceEnsemble(object):
    """wt_ensemble_groups_dev on tensors already in HBM."""

    def __init__(self, dets5, group_offsets, input_sizes, k_inputs, method, iou_thresh, cut, device='cuda'):
        self.lib = _lib.lib()
        dev = torch.device(device)
        self.dets5 = torch.from_numpy(np.ascontiguousarray(dets5, dtype=np.float64)).to(dev)
        self.off = torch.from_numpy(np.ascontiguousarray(group_offsets, dtype=np.int64)).to(dev)
        self.sizes = torch.from_numpy(np.ascontiguousarray(input_sizes, dtype=np.int32)).to(dev)
        self.n_rows = int(len(dets5))
        self.n_groups = int(len(group_offsets) - 1)
        self.max_rows = int(np.diff(np.asarray(group_offsets)).max()) if self.n_groups else 0
        self.k, self.method, self.thr, self.cut = int(k_inputs), int(method), float(iou_thresh), float(cut)
        ws = int(self.lib.wt_ensemble_groups_workspace(C.c_int64(self.n_rows), C.c_int64(self.n_groups),
                                                       C.c_int64(self.max_rows)))
        self.ws_bytes = ws
        self.workspace = torch.empty(max(ws, 16), dtype=torch.uint8, device=dev)
        self.out5 = torch.empty((self.n_rows + 1, 5), dtype=torch.float64, device=dev)
        self.counts = torch.zeros(self.n_groups + 1, dtype=torch.int64, device=dev)

    def run(self):
        rc = self.lib.wt_ensemble_groups_dev(_dp(self.dets5), _dp(self.off), _dp(self.sizes), C.c_int64(self.n_rows),
                                             C.c_int64(self.n_groups), C.c_int64(self.max_rows), C.c_int(self.k),
                                             C.c_int(self.method), C.c_double(self.thr), C.c_double(self.cut),
                                             _dp(self.out5), _dp(self.counts), _dp(self.workspace),
                                             C.c_size_t(self.ws_bytes), _stream())
        _lib.check(rc, 'wt_ensemble_groups_dev')
