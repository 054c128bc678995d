"""ensemble submission together - mirrors /root/reference/detnet/ensemble.py (CLI flags, JSON formats).

``python -m waymo_2d_tracking_amd.detnet.ensemble A.json B.json -o OUT.json -m soft_nms --min-score=0.01
--soft-nms-cut=0.9`` is the drop-in for ``python -m detnet.ensemble ...`` (ensemble.py:87-160).  Every
(image, category) group of the whole submission set is merged in ONE call of ``wt_ensemble_groups_host``
(include/waymotrack.h): one wavefront per group on the GPU instead of the reference's per-image
ProcessPoolExecutor (ensemble.py:152-157).  Host code only parses / formats JSON.
"""
import argparse
import json
import numbers
from collections import defaultdict
from pathlib import Path

import numpy as np

METHODS = {'weighted_fusion': 0, 'nms': 1, 'soft_nms': 2}


def convert_submission(det_list, weight, min_score=0):
    """ensemble.py:31-47 -> {image_id: {category_id: [[score*weight, x, y, w, h], ...]}}"""
    detections = defaultdict(lambda: defaultdict(list))
    for det in det_list:
        bbox = det['bbox']
        if bbox[2] > 0 and bbox[3] > 0:
            row = [det['score'] * weight] + list(bbox)
            if row[0] >= min_score:
                detections[det['image_id']][det['category_id']].append(row)
    return detections


def load_yml_input_and_weight(input_files_with_weights, prefix=''):
    """ensemble.py:67-75: nested {dir: {file: weight}} -> [(path, weight), ...]"""
    results = []
    for k, v in input_files_with_weights.items():
        p = prefix + '/' + k if prefix else k
        if isinstance(v, numbers.Number):
            results.append((p, v))
        else:
            results += load_yml_input_and_weight(v, p)
    return results


def load_input_submissions(input_files, input_weights, min_score=0):
    """ensemble.py:78-84.  image ids are returned in first-appearance order (the reference iterates a set,
    so its output order depends on PYTHONHASHSEED - SURVEY App. D-4); category ids sorted."""
    input_submissions = []
    for f in input_files:
        with Path(f).open() as fp:
            input_submissions.append(json.load(fp))
    category_ids = sorted(set(d['category_id'] for det in input_submissions for d in det))
    input_detections = [convert_submission(d, w, min_score) for d, w in zip(input_submissions, input_weights)]
    image_ids = list(dict.fromkeys(k for det in input_detections for k in det.keys()))
    return image_ids, category_ids, input_detections


def pack_groups(image_ids, category_ids, input_detections):
    """Flatten every (image, category) group into the CSR layout of wt_ensemble_groups."""
    k = len(input_detections)
    rows, offsets, sizes, keys = [], [0], [], []
    n = 0
    for image_id in image_ids:
        per_input = [d.get(image_id, {}) if not isinstance(d, defaultdict) else (d[image_id] if image_id in d else {})
                     for d in input_detections]
        for category_id in category_ids:
            for det in per_input:
                r = det.get(category_id, ()) if category_id in det else ()
                rows.extend(r)
                sizes.append(len(r))
                n += len(r)
            offsets.append(n)
            keys.append((image_id, category_id))
    dets5 = np.asarray(rows, dtype=np.float64).reshape(-1, 5)
    return dict(dets5=np.ascontiguousarray(dets5), group_offsets=np.asarray(offsets, dtype=np.int64),
                input_sizes=np.asarray(sizes, dtype=np.int32).reshape(-1, k), keys=keys)


def format_groups(packed, out5, counts, min_score):
    """ensemble.py:59-63: keep score > min_score, bbox.astype(int) (truncation), round(score, 5)."""
    output_json = []
    off = packed['group_offsets']
    boxes = np.trunc(out5[:, 1:5]).astype(np.int64) if len(out5) else np.zeros((0, 4), np.int64)
    for g, (image_id, category_id) in enumerate(packed['keys']):
        for i in range(int(off[g]), int(off[g]) + int(counts[g])):
            s = float(out5[i, 0])
            if s > min_score:
                output_json.append({'image_id': image_id, 'category_id': category_id,
                                    'bbox': boxes[i].tolist(), 'score': round(s, 5)})
    return output_json


def ensemble_all(image_ids, category_ids, input_detections, method='weighted_fusion', iou_thresh=0.5,
                 soft_nms_cut=1.0, min_score=0.0):
    """All groups of all images on the GPU; returns the output JSON rows (ensemble.py:144-157)."""
    import ctypes as C
    from .. import _lib
    packed = pack_groups(image_ids, category_ids, input_detections)
    d = packed['dets5']
    G = len(packed['keys'])
    out5 = np.zeros((len(d) + 1, 5), dtype=np.float64)
    counts = np.zeros(G + 1, dtype=np.int64)
    rc = _lib.lib().wt_ensemble_groups_host(
        _lib.ptr(d), _lib.ptr(packed['group_offsets']), _lib.ptr(packed['input_sizes']), C.c_int64(G),
        C.c_int(len(input_detections)), C.c_int(METHODS[method]), C.c_double(iou_thresh), C.c_double(soft_nms_cut),
        _lib.ptr(out5), _lib.ptr(counts))
    _lib.check(rc, 'wt_ensemble_groups_host')
    return format_groups(packed, out5[:len(d)], counts[:G], min_score)


def main(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                     fromfile_prefix_chars='@')
    parser.add_argument('inputs', type=str, nargs='+', help='input json files')
    parser.add_argument('-o', '--output', type=str, help='output json file')
    parser.add_argument('-m', '--method', choices=("weighted_fusion", "nms", "soft_nms"), default="weighted_fusion",
                        help='method to merge bbox detections')
    parser.add_argument('--iou-thresh', type=float, default=0.5, help='IOU threshold for merging bboxes')
    parser.add_argument('--soft-nms-cut', type=float, default=1.0, help='cutout IoU threshold for soft nms')
    parser.add_argument('--min-score', type=float, default=0, help='minimal score to keep')
    parser.add_argument('-j', '--jobs', type=int, default=1,
                        help='accepted for compatibility (groups are merged in parallel on the GPU)')
    args = parser.parse_args(argv)

    input_files = []
    for f in args.inputs:
        f = Path(f)
        if f.is_file():
            input_files.append(f)
        elif f.is_dir():
            input_files += sorted(f.glob("**/*.json"))
        else:
            print(f"{f} is neither file nor dir?!")

    input_weights = None
    if len(input_files) == 1 and input_files[0].suffix == '.yml':
        import yaml
        with input_files[0].open() as fp:
            input_files_with_weights = load_yml_input_and_weight(yaml.safe_load(fp))
        input_files, input_weights = zip(*input_files_with_weights)
        print(input_files, input_weights)

    assert len(input_files) > 1
    print('input files:', input_files)
    if not input_weights:
        input_weights = [1] * len(input_files)     # the reference crashes here; intended value
    input_weights_weight = max(input_weights)
    input_weights = [w / input_weights_weight for w in input_weights]
    print('weights', input_weights)

    output_file = Path(args.output)
    output_file.parent.mkdir(parents=True, exist_ok=True)
    if output_file.exists():
        raise RuntimeError(f"output file {output_file} exists!")

    image_ids, category_ids, input_detections = load_input_submissions(input_files, input_weights, args.min_score)
    print('No. Images:', len(image_ids))
    print('No. categories:', len(category_ids))
    output_json = ensemble_all(image_ids, category_ids, input_detections, args.method, args.iou_thresh,
                               args.soft_nms_cut, args.min_score)
    with output_file.open('wt') as fp:
        json.dump(output_json, fp)


if __name__ == '__main__':
    main()
