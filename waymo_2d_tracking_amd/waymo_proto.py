"""Waymo Open Dataset protobuf emit from detection / tracking JSON rows: the host side of
wt_waymo_objects_write (csrc/waymo_proto.hip).  The rows are turned into columns once; the message bytes are produced
natively - no per-object Python protobuf messages (the reference builds ~10^6-10^7 of them, coco_to_waymo.py:52-60)."""
import ctypes as C

import numpy as np

from . import _lib

# dataset.proto CameraName.Name / label.proto Label.Type, Label.DifficultyLevel / submission.proto Submission.Task, SensorType
CAMERA_NAMES = {'UNKNOWN': 0, 'FRONT': 1, 'FRONT_LEFT': 2, 'FRONT_RIGHT': 3, 'SIDE_LEFT': 4, 'SIDE_RIGHT': 5}
TYPE_VEHICLE, TYPE_PEDESTRIAN, TYPE_SIGN, TYPE_CYCLIST = 1, 2, 3, 4
DETECTION_2D, TRACKING_2D = 1, 3
CAMERA_ALL = 3


def _blob(strings):
    data = [s.encode('utf-8') for s in strings]
    off = np.zeros(len(data) + 1, dtype=np.int64)
    np.cumsum([len(d) for d in data], out=off[1:])
    return b''.join(data), off


def entries_to_columns(entries, type_of_category=None, levels=None):
    """JSON rows ({'image_id': 'segment/timestamp/CAMERA', 'bbox', 'category_id'[, 'score', 'object_id', ..._difficulty_level]})
    -> the columns wt_waymo_objects_write takes.  KeyError / ValueError on malformed rows, like the reference's loops."""
    n = len(entries)
    ctx, ts, cam = [], np.zeros(n, np.int64), np.zeros(n, np.int32)
    bbox = np.zeros((n, 4), np.float64)
    typ = np.zeros(n, np.int32)
    has_score = n > 0 and all('score' in e for e in entries)
    if not has_score and any('score' in e for e in entries):
        raise ValueError('some rows carry a score and some do not')
    score = np.zeros(n, np.float64) if has_score else None
    ids, has_id = [], np.zeros(n, np.uint8)
    det = np.zeros(n, np.int32)
    trk = np.zeros(n, np.int32)
    for i, e in enumerate(entries):
        segment, stamp, camera = e['image_id'].split('/')
        ctx.append(segment)
        ts[i] = int(stamp)
        cam[i] = CAMERA_NAMES[camera]
        bbox[i] = e['bbox']
        c = e['category_id']
        typ[i] = c if type_of_category is None else type_of_category[c]
        if typ[i] == 0:
            raise AssertionError('TYPE_UNKNOWN')                     # coco_to_waymo.py:49
        if has_score:
            score[i] = e['score']
        if 'object_id' in e:
            has_id[i] = 1
            ids.append(str(e['object_id']))
        else:
            ids.append('')
        if levels is not None:
            if 'detection_difficulty_level' in e:
                det[i] = levels[e['detection_difficulty_level']]
            if 'tracking_difficulty_level' in e:
                trk[i] = levels[e['tracking_difficulty_level']]
    return dict(n=n, context=_blob(ctx), timestamp=ts, camera=cam, bbox=bbox, score=score, type=typ,
                ids=_blob(ids) if has_id.any() else None, has_id=has_id, det_level=det, trk_level=trk)


def write(path, cols, metrics_mode=False, submission=None):
    """submission: None (bare metrics.Objects) or dict(task, account_name, unique_method_name, authors, affiliation,
    description, sensor_type).  Returns the number of bytes written."""
    ctx_blob, ctx_off = cols['context']
    id_blob, id_off = cols['ids'] if cols['ids'] is not None else (None, None)
    out = C.c_int64(0)
    sub = submission or {}
    authors = b''.join(a.encode('utf-8') + b'\0' for a in sub.get('authors', []))
    enc = lambda v: None if v is None else v.encode('utf-8')
    _lib.check(_lib.lib().wt_waymo_objects_write(
        str(path).encode(), C.c_int64(cols['n']), ctx_blob, _lib.ptr(ctx_off), _lib.ptr(cols['timestamp']), _lib.ptr(cols['camera']),
        _lib.ptr(np.ascontiguousarray(cols['bbox'])), _lib.ptr(cols['score']), _lib.ptr(cols['type']), id_blob, _lib.ptr(id_off),
        _lib.ptr(cols['has_id']) if id_off is not None else None, _lib.ptr(cols['det_level']), _lib.ptr(cols['trk_level']),
        C.c_int(1 if metrics_mode else 0), C.c_int(1 if submission else 0), C.c_int(sub.get('task', 0)),
        enc(sub.get('account_name')), enc(sub.get('unique_method_name')), authors or None, C.c_int(len(sub.get('authors', []))),
        enc(sub.get('affiliation')), enc(sub.get('description')), C.c_int(sub.get('sensor_type', 0)), C.byref(out)),
        'wt_waymo_objects_write')
    return out.value
