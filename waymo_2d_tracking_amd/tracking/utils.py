"""Host glue of the tracking stage - mirrors /root/reference/tracking/utils.py.

``read_data_file`` and ``track_sort`` keep the reference's names, arguments and return values
(utils.py:63-96, :25-60).  ``track_all`` is the batched form of the loop in tracking/track.py:43-47: all
(segment, camera) streams of a file go to the GPU in ONE call of ``wt_track_streams_host``
(include/waymotrack.h), where every (stream, class) tracker is one wavefront of the persistent SORT kernel.
No arithmetic of the tracker runs on the host.
"""
import ctypes as C
import json

import numpy as np

from .. import _lib

# utils.py:11-17
IMAGE_SIZES = {
    'FRONT': [1920, 1280],
    'FRONT_LEFT': [1920, 1280],
    'FRONT_RIGHT': [1920, 1280],
    'SIDE_LEFT': [1920, 886],
    'SIDE_RIGHT': [1920, 886]
}


def clip_xy(camera_id, x, y):
    """utils.py:20-22 (kept for API parity; the batched path clips inside the kernel)."""
    w, h = IMAGE_SIZES[camera_id]
    return np.clip(x, a_min=0, a_max=w), np.clip(y, a_min=0, a_max=h)


def read_data_file(file_name, score_threshold):
    """utils.py:63-96: detections JSON -> {segment: {camera: {frame(int): [entry, ...]}}}.

    The frame key is created before the w/h < 1 and score filters, so a frame whose detections are all
    dropped still exists (and still ticks the trackers)."""
    entries = {}
    with open(file_name) as fp:
        raw_entries = json.load(fp)
    if 'annotations' in raw_entries:
        raw_entries = raw_entries['annotations']
    for entry in raw_entries:
        segment_id, frame_id, camera_id = entry['image_id'].split('/')
        frames = entries.setdefault(segment_id, {}).setdefault(camera_id, {})
        bucket = frames.setdefault(int(frame_id), [])
        bbox = entry['bbox']
        if bbox[2] < 1 or bbox[3] < 1:
            continue
        category_id = entry['category_id']
        score = entry['score'] if 'score' in entry else 1.0
        if score < score_threshold[category_id - 1]:
            continue
        new_entry = {'bbox': bbox, 'score': score, 'category_id': category_id}
        if 'object_id' in entry:
            new_entry['object_id'] = entry['object_id']
        bucket.append(new_entry)
    return entries


def pack_streams(predictions, stream_keys=None):
    """Nested dict -> the SoA/CSR layout of wt_track_streams (include/waymotrack.h).

    stream_keys: list of (segment_id, camera_id); default = insertion order of the dict, i.e. the loop order
    of tracking/track.py:43-47.  Frames of a stream ascend (utils.py:31)."""
    if stream_keys is None:
        stream_keys = [(s, c) for s in predictions for c in predictions[s]]
    xs, ys, ws, hs, ss, cs = [], [], [], [], [], []
    frame_off = [0]
    stream_off = [0]
    frame_ids = []
    clip_w, clip_h = [], []
    n = 0
    for segment_id, camera_id in stream_keys:
        cam = predictions[segment_id][camera_id]
        for frame_id in sorted(cam.keys()):
            for e in cam[frame_id]:
                b = e['bbox']
                xs.append(b[0]); ys.append(b[1]); ws.append(b[2]); hs.append(b[3])
                ss.append(e['score']); cs.append(e['category_id'])
            n += len(cam[frame_id])
            frame_off.append(n)
            frame_ids.append(frame_id)
        stream_off.append(len(frame_ids))
        wh = IMAGE_SIZES.get(camera_id)
        clip_w.append(wh[0] if wh else 0.0)
        clip_h.append(wh[1] if wh else 0.0)
    return dict(
        x=np.asarray(xs, dtype=np.float64), y=np.asarray(ys, dtype=np.float64),
        w=np.asarray(ws, dtype=np.float64), h=np.asarray(hs, dtype=np.float64),
        score=np.asarray(ss, dtype=np.float64), category=np.asarray(cs, dtype=np.int32),
        frame_det_offsets=np.asarray(frame_off, dtype=np.int64),
        stream_frame_offsets=np.asarray(stream_off, dtype=np.int64),
        frame_ids=np.asarray(frame_ids, dtype=np.int64),
        clip_w=np.asarray(clip_w, dtype=np.float64), clip_h=np.asarray(clip_h, dtype=np.float64),
        stream_keys=list(stream_keys))


def slice_streams(packed, lo, hi):
    """Streams [lo, hi) of a packed set as a packed set of their own (CSR offsets rebased) - the shard of one rank."""
    so = np.asarray(packed['stream_frame_offsets'])
    fo = np.asarray(packed['frame_det_offsets'])
    f0, f1 = int(so[lo]), int(so[hi])
    d0, d1 = int(fo[f0]), int(fo[f1])
    out = {k: np.ascontiguousarray(packed[k][d0:d1]) for k in ('x', 'y', 'w', 'h', 'score', 'category')}
    out['frame_det_offsets'] = np.ascontiguousarray(fo[f0:f1 + 1] - d0)
    out['stream_frame_offsets'] = np.ascontiguousarray(so[lo:hi + 1] - f0)
    out['frame_ids'] = np.ascontiguousarray(packed['frame_ids'][f0:f1]) if 'frame_ids' in packed else None
    out['clip_w'] = np.ascontiguousarray(packed['clip_w'][lo:hi])
    out['clip_h'] = np.ascontiguousarray(packed['clip_h'][lo:hi])
    if 'stream_keys' in packed:
        out['stream_keys'] = list(packed['stream_keys'][lo:hi])
    return out


def make_params(max_age, min_hits, score_threshold, iou_threshold):
    """wt_track_params + the arrays that must stay alive while it is used."""
    n_classes = len(iou_threshold)
    st = np.full(n_classes, -np.inf) if score_threshold is None else _lib.as_f64(score_threshold)
    it = _lib.as_f64(iou_threshold)
    if st.size != n_classes:
        raise ValueError('score_threshold and iou_threshold need one entry per class')
    p = _lib.TrackParams(int(max_age), int(min_hits), int(n_classes), 0, st.ctypes.data, it.ctypes.data)
    return p, (st, it)


def track_packed(packed, iou_thresholds, max_age, min_hits, score_threshold=None, id_base=0):
    """Run wt_track_streams_host on a packed set of streams.  Returns (dict of output arrays, n_births)."""
    lib = _lib.lib()
    n = int(packed['x'].size)
    params, keep = make_params(max_age, min_hits, score_threshold, iou_thresholds)
    if n and (packed['category'].min() < 1 or packed['category'].max() > params.n_classes):
        raise IndexError('category_id outside 1..%d (thresholds are indexed by category_id-1)' % params.n_classes)
    out_frame = np.zeros(n + 1, np.int64)
    out_cat = np.zeros(n + 1, np.int32)
    out_bbox = np.zeros((n + 1, 4), np.float64)
    out_score = np.zeros(n + 1, np.float64)
    out_id = np.zeros(n + 1, np.int64)
    n_out = C.c_int64(0)
    n_births = C.c_int64(0)
    rc = lib.wt_track_streams_host(
        C.c_int64(n), _lib.ptr(packed['x']), _lib.ptr(packed['y']), _lib.ptr(packed['w']), _lib.ptr(packed['h']),
        _lib.ptr(packed['score']), _lib.ptr(packed['category']),
        C.c_int64(packed['frame_det_offsets'].size - 1), _lib.ptr(packed['frame_det_offsets']),
        C.c_int32(packed['stream_frame_offsets'].size - 1), _lib.ptr(packed['stream_frame_offsets']),
        _lib.ptr(packed['clip_w']), _lib.ptr(packed['clip_h']), C.byref(params), C.c_int64(id_base),
        _lib.ptr(out_frame), _lib.ptr(out_cat), _lib.ptr(out_bbox), _lib.ptr(out_score), _lib.ptr(out_id),
        C.byref(n_out), C.byref(n_births))
    _lib.check(rc, 'wt_track_streams_host')
    k = n_out.value
    return dict(frame=out_frame[:k], category=out_cat[:k], bbox=out_bbox[:k], score=out_score[:k],
                object_id=out_id[:k]), n_births.value


def format_tracks(packed, out):
    """Output arrays -> the list of dicts of utils.py:52-58 (tracking JSON rows)."""
    frame_ids = packed['frame_ids']
    stream_of_frame = np.searchsorted(packed['stream_frame_offsets'], out['frame'], side='right') - 1
    keys = packed['stream_keys']
    bbox = out['bbox'].tolist()
    score = out['score'].tolist()
    cat = out['category'].tolist()
    oid = out['object_id'].tolist()
    fr = frame_ids[out['frame']].tolist() if len(out['frame']) else []
    so = stream_of_frame.tolist()
    rows = []
    for i in range(len(fr)):
        segment_id, camera_id = keys[so[i]]
        rows.append({'image_id': '%s/%i/%s' % (segment_id, fr[i], camera_id),
                     'bbox': bbox[i], 'score': score[i], 'category_id': cat[i], 'object_id': '%i' % oid[i]})
    return rows


# process-global ID counter used by track_sort(), the twin of KalmanBoxTracker.count (sort/sort.py:86)
_GLOBAL_IDS = {'next': 0}


def reset_global_ids(value=0):
    _GLOBAL_IDS['next'] = int(value)


def track_sort(predictions, segment_id, camera_id, iou_thresholds, max_age, min_hits):
    """utils.py:25-60, same signature and return value; the stream is tracked on the GPU.

    Track IDs continue from the process-global counter exactly like successive calls in the reference."""
    packed = pack_streams(predictions, [(segment_id, camera_id)])
    out, births = track_packed(packed, iou_thresholds, max_age, min_hits, None, _GLOBAL_IDS['next'])
    _GLOBAL_IDS['next'] += births
    return format_tracks(packed, out)


def track_all(predictions, iou_thresholds, max_age, min_hits, segment_ids=None):
    """Batched equivalent of the loop tracking/track.py:43-47 (one GPU call for every stream)."""
    keys = [(s, c) for s in predictions if (segment_ids is None or s in segment_ids) for c in predictions[s]]
    packed = pack_streams(predictions, keys)
    out, births = track_packed(packed, iou_thresholds, max_age, min_hits, None, _GLOBAL_IDS['next'])
    _GLOBAL_IDS['next'] += births
    return format_tracks(packed, out)


# ---------------------------------------------------------------------------------------------------------------
# native JSON I/O (include/waymotrack.h, wt_detfile_* / wt_tracks_write_json): the same result as
# read_data_file + pack_streams / json.dump(format_tracks(...)) without building Python objects per row
class NativeDetFile(object):
    def __init__(self, path, score_threshold):
        lib = _lib.lib()
        st = _lib.as_f64(score_threshold)
        self._keep = st
        h = C.c_void_p()
        _lib.check(lib.wt_detfile_read(str(path).encode(), _lib.ptr(st), C.c_int(len(st)), C.byref(h)), 'wt_detfile_read')
        self._h = h
        self.lib = lib

    def packed(self):
        """The dict pack_streams() would return (numpy views copied out of the handle)."""
        lib, h = self.lib, self._h
        for name, rt in (('wt_detfile_num_dets', C.c_int64), ('wt_detfile_num_frames', C.c_int64),
                         ('wt_detfile_num_streams', C.c_int32), ('wt_detfile_segment', C.c_char_p),
                         ('wt_detfile_camera', C.c_char_p)):
            getattr(lib, name).restype = rt
        n, nf, ns = lib.wt_detfile_num_dets(h), lib.wt_detfile_num_frames(h), lib.wt_detfile_num_streams(h)

        def arr(fn, count, ctype, dtype):
            f = getattr(lib, fn)
            f.restype = C.POINTER(ctype)
            if count == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(f(h), shape=(count,)).astype(dtype, copy=True)
        keys = [(lib.wt_detfile_segment(h, C.c_int32(s)).decode(), lib.wt_detfile_camera(h, C.c_int32(s)).decode())
                for s in range(ns)]
        clip = [IMAGE_SIZES.get(c) or [0.0, 0.0] for _, c in keys]
        return dict(
            x=arr('wt_detfile_x', n, C.c_double, np.float64), y=arr('wt_detfile_y', n, C.c_double, np.float64),
            w=arr('wt_detfile_w', n, C.c_double, np.float64), h=arr('wt_detfile_h', n, C.c_double, np.float64),
            score=arr('wt_detfile_score', n, C.c_double, np.float64),
            category=arr('wt_detfile_category', n, C.c_int32, np.int32),
            frame_det_offsets=arr('wt_detfile_frame_det_offsets', nf + 1, C.c_int64, np.int64),
            stream_frame_offsets=arr('wt_detfile_stream_frame_offsets', ns + 1, C.c_int64, np.int64),
            frame_ids=arr('wt_detfile_frame_ids', nf, C.c_int64, np.int64),
            clip_w=np.asarray([c[0] for c in clip], dtype=np.float64), clip_h=np.asarray([c[1] for c in clip], dtype=np.float64),
            stream_keys=keys)

    def write_tracks(self, path, out):
        n = len(out['frame'])
        fr = np.ascontiguousarray(out['frame'], dtype=np.int64)
        cat = np.ascontiguousarray(out['category'], dtype=np.int32)
        bb = np.ascontiguousarray(out['bbox'], dtype=np.float64)
        sc = np.ascontiguousarray(out['score'], dtype=np.float64)
        oid = np.ascontiguousarray(out['object_id'], dtype=np.int64)
        _lib.check(self.lib.wt_tracks_write_json(str(path).encode(), self._h, C.c_int64(n), _lib.ptr(fr), _lib.ptr(cat),
                                                 _lib.ptr(bb), _lib.ptr(sc), _lib.ptr(oid)), 'wt_tracks_write_json')

    def close(self):
        if getattr(self, '_h', None):
            self.lib.wt_detfile_free(self._h)
            self._h = None

    __del__ = close
