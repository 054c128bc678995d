"""Export of detection results - the JSON leg of /root/reference/detnet/export.py:159-176 (`export(..., format='json')` ->
`export_json` -> `dataset.load_prediction`, detnet/data/coco.py:229-252) on columns, written by the native JSON writer
(wt_detections_write_json: byte-identical to json.dump of the reference's row dicts)."""
import ctypes as C
from pathlib import Path

import numpy as np

from .. import _lib


def detection_rows(predictions, image_sizes, category_ids=None):
    """COCODetection.load_prediction (coco.py:229-252) on the column store: for every image of `image_sizes`
    ({image_id: (width, height)}, the reference iterates coco.imgs) and class: [score, cx, cy, w, h] normalised ->
    bbox = [int(x_left), int(y_top), int(w), int(h)] in pixels (truncation toward zero), score rounded to 5 decimals.
    Returns columns (image index into predictions.image_ids, category, bbox int64 (n,4), score float64) in the reference's row
    order: images in `image_sizes` order, classes ascending, detections in stored order."""
    cols, _ = predictions.shard_columns()
    n_cls = len(predictions.classnames)
    cat_of = np.asarray(category_ids if category_ids else list(range(1, n_cls + 1)), dtype=np.int32)
    idx = {k: i for i, k in enumerate(predictions.image_ids)}
    wh = np.zeros((len(predictions.image_ids), 2), np.float64)
    rank = np.full(len(predictions.image_ids), -1, np.int64)
    for r, (image_id, (width, height)) in enumerate(image_sizes.items()):
        i = idx.get(str(image_id))
        if i is None or not predictions.tested[i]:
            raise KeyError('no prediction for image %r' % image_id)       # the reference fails on predictions[...] = None too
        wh[i] = (width, height)
        rank[i] = r
    img = cols['image']
    keep = rank[img] >= 0
    order = np.lexsort((np.arange(len(img))[keep], cols['cls'][keep], rank[img[keep]]))
    sel = np.nonzero(keep)[0][order]
    im = img[sel]
    # coco.py:245-246 in the reference's arithmetic: float32 detections * int sizes -> float64 products
    scale = wh[im]
    cx = cols['cx'][sel].astype(np.float64) * scale[:, 0]; cy = cols['cy'][sel].astype(np.float64) * scale[:, 1]
    bw = cols['w'][sel].astype(np.float64) * scale[:, 0]; bh = cols['h'][sel].astype(np.float64) * scale[:, 1]
    bbox = np.trunc(np.stack((cx - bw / 2, cy - bh / 2, bw, bh), axis=1)).astype(np.int64)
    score = np.asarray([round(float(s), 5) for s in cols['score'][sel].tolist()], dtype=np.float64)
    return dict(image=im.astype(np.int32), category=cat_of[cols['cls'][sel]], bbox=bbox, score=score)


def write_detections_json(path, image_ids, rows):
    """json.dump([{image_id, category_id, bbox, score}, ...]) of column rows through libwaymotrack."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    blobs = [str(s).encode('utf-8') for s in image_ids]
    offsets = np.zeros(len(blobs) + 1, np.int64)
    np.cumsum([len(b) for b in blobs], out=offsets[1:])
    blob = b''.join(blobs) or b'\0'
    img = np.ascontiguousarray(rows['image'], dtype=np.int32)
    cat = np.ascontiguousarray(rows['category'], dtype=np.int32)
    bbox = np.ascontiguousarray(rows['bbox'], dtype=np.int64).reshape(-1, 4)
    score = np.ascontiguousarray(rows['score'], dtype=np.float64)
    rc = _lib.lib().wt_detections_write_json(str(path).encode(), C.c_int64(len(img)), _lib.ptr(img), C.c_int32(len(blobs)),
                                             C.c_char_p(blob), _lib.ptr(offsets), _lib.ptr(cat), _lib.ptr(bbox), _lib.ptr(score))
    _lib.check(rc, 'wt_detections_write_json')


def export_json(predictions, output_filename, image_sizes, category_ids=None):
    """export.py:159-165"""
    output_path = Path(output_filename).with_suffix('.json')
    write_detections_json(output_path, predictions.image_ids, detection_rows(predictions, image_sizes, category_ids))
    return output_path


def export(predictions, output_filename, image_sizes, format='json', threshold=0.0, category_ids=None):
    """export.py:168-176 for the one format on the Waymo path."""
    if format != 'json':
        raise NotImplementedError("only --export-format json is on the detection -> tracking path")
    return export_json(predictions, output_filename, image_sizes, category_ids)
