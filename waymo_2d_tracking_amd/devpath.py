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


class StreamingTracker(object):
    """wt_track_state_* / wt_track_chunk_dev: trackers that stay resident in HBM while the frames of their streams arrive
    chunk by chunk (online detect -> track; tracking/utils.py:29-36 keeps one MultiClassTrackerSort per stream alive).

    All chunks of one tracker share `max_frame_dets`; a chunk holds at most `max_chunk_frames` frames / `max_chunk_dets`
    detection slots.  Nothing here synchronises with the host."""

    def __init__(self, n_streams, max_frame_dets, max_chunk_frames, max_chunk_dets, iou_threshold, max_age, min_hits,
                 score_threshold=None, device='cuda'):
        self.lib = _lib.lib()
        self.dev = torch.device(device)
        self.n_streams, self.max_frame = int(n_streams), int(max_frame_dets)
        self.params, self._keep = make_params(max_age, min_hits, score_threshold, iou_threshold)
        sb = int(self.lib.wt_track_state_bytes(C.c_int32(self.n_streams), C.c_int64(self.max_frame), C.byref(self.params)))
        wb = int(self.lib.wt_track_chunk_workspace(C.c_int64(max_chunk_dets), C.c_int64(max_chunk_frames),
                                                   C.c_int32(self.n_streams), C.c_int64(self.max_frame),
                                                   C.byref(self.params)))
        if sb == 0 or wb == 0:
            raise _lib.WaymoTrackError('streaming tracker sizes: ' + (self.lib.wt_last_error() or b'').decode())
        self.state = torch.empty(sb, dtype=torch.uint8, device=self.dev)
        self.workspace = torch.empty(wb, dtype=torch.uint8, device=self.dev)
        self.prefix = torch.zeros(self.n_streams + 1, dtype=torch.int64, device=self.dev)
        self.max_chunk_frames, self.max_chunk_dets = int(max_chunk_frames), int(max_chunk_dets)
        self.reset()

    def reset(self):
        """Fresh trackers for every stream (a new segment starts)."""
        _lib.check(self.lib.wt_track_state_init_dev(_dp(self.state), C.c_size_t(self.state.numel()),
                                                    C.c_int32(self.n_streams), C.c_int64(self.max_frame),
                                                    C.byref(self.params), _stream()), 'wt_track_state_init_dev')

    def feed(self, x, y, w, h, score, category, frame_off, stream_off, clip_w, clip_h, out_frame, out_category, out_bbox,
             out_score, out_local_id, counts):
        """Enqueue one chunk on the current stream.  counts: int64[2] device tensor (rows, births of this chunk)."""
        n = int(x.numel())
        nf = int(frame_off.numel()) - 1
        if n > self.max_chunk_dets or nf > self.max_chunk_frames:
            raise ValueError('chunk exceeds the sizes this tracker was created for')
        rc = self.lib.wt_track_chunk_dev(
            _dp(self.state), C.c_size_t(self.state.numel()), C.c_int64(n), _dp(x), _dp(y), _dp(w), _dp(h), _dp(score),
            _dp(category), C.c_int64(nf), _dp(frame_off), C.c_int32(self.n_streams), _dp(stream_off), _dp(clip_w),
            _dp(clip_h), C.c_int64(self.max_frame), C.byref(self.params), _dp(out_frame), _dp(out_category), _dp(out_bbox),
            _dp(out_score), _dp(out_local_id), C.c_void_p(counts.data_ptr()), C.c_void_p(counts.data_ptr() + 8),
            _dp(self.workspace), C.c_size_t(self.workspace.numel()), _stream())
        _lib.check(rc, 'wt_track_chunk_dev')

    def global_ids(self, row_stream, local_id, id_base=0):
        """Reference object ids (sort.py:86,140-141 order) of collected rows, once their streams are complete."""
        row_stream = row_stream.to(torch.int32).contiguous()
        local_id = local_id.to(torch.int64).contiguous()
        out = torch.empty_like(local_id)
        rc = self.lib.wt_track_global_ids_dev(_dp(self.state), C.c_size_t(self.state.numel()), C.c_int32(self.n_streams),
                                              C.c_int64(self.max_frame), C.byref(self.params), C.c_int64(local_id.numel()),
                                              _dp(row_stream), _dp(local_id), C.c_int64(id_base), _dp(out), _dp(self.prefix),
                                              _stream())
        _lib.check(rc, 'wt_track_global_ids_dev')
        return out
