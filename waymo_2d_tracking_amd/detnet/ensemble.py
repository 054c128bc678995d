"""ensemble submission together - drop-in for /root/reference/detnet/ensemble.py (CLI flags, JSON formats).

``python -m waymo_2d_tracking_amd.detnet.ensemble A.json B.json -o OUT.json -m soft_nms --min-score=0.01
--soft-nms-cut=0.9`` replaces ``python -m detnet.ensemble ...`` (ensemble.py:87-160).

Data path: the input files are parsed by the native reader (wt_detjson_read) straight into columns; the filters of
``convert_submission`` (ensemble.py:31-47), the weights and the (image, category) grouping (ensemble.py:50-55) are column
operations; every group of the whole submission set is merged in ONE call of ``wt_ensemble_groups_host`` (one workgroup per
group on the GPU instead of the reference's per-image ProcessPoolExecutor, ensemble.py:152-157); the surviving rows
(score > min_score, integer boxes, 5-decimal scores, ensemble.py:59-63) are written by the native JSON writer.
Under ``torchrun`` (one process per GPU) the images are split into contiguous blocks, each rank merges its groups and the
result columns reach rank 0 in one tensor gather (RCCL over xGMI) - no data-path collective.
"""
import argparse
import ctypes as C
import numbers
import os
from pathlib import Path

import numpy as np

from .. import _lib

METHODS = {'weighted_fusion': 0, 'nms': 1, 'soft_nms': 2}


# ---------------------------------------------------------------------------------------------------------------------
# reference-shaped helpers (dict API kept for callers of the reference's functions)
def convert_submission(det_list, weight, min_score=0):
    """ensemble.py:31-47: JSON rows -> {image_id: {category_id: [[score * weight, x, y, w, h], ...]}}; rows with a
    non-positive width / height or a weighted score below min_score are dropped."""
    grouped = {}
    for entry in det_list:
        x, y, w, h = entry['bbox']
        score = entry['score'] * weight
        if not (w > 0 and h > 0) or score < min_score:
            continue
        grouped.setdefault(entry['image_id'], {}).setdefault(entry['category_id'], []).append([score, x, y, w, h])
    return grouped


def load_yml_input_and_weight(tree, prefix=''):
    """ensemble.py:67-75: the nested {directory: {file: weight}} mapping of a weights .yml flattened to [(path, weight)] in
    written order."""
    def walk(node, base):
        for name, value in node.items():
            path = f'{base}/{name}' if base else name
            if isinstance(value, numbers.Number):
                yield path, value
            else:
                yield from walk(value, path)
    return list(walk(tree, prefix))


# ---------------------------------------------------------------------------------------------------------------------
# column path
def read_submission(path):
    """json.load of one detection file (ensemble.py:79) through the native reader -> columns."""
    lib = _lib.lib()
    h = C.c_void_p()
    _lib.check(lib.wt_detjson_read(str(path).encode(), C.byref(h)), 'wt_detjson_read')
    try:
        lib.wt_detjson_num_rows.restype = C.c_int64
        lib.wt_detjson_num_images.restype = C.c_int32
        lib.wt_detjson_image_id.restype = C.c_char_p
        n, ni = lib.wt_detjson_num_rows(h), lib.wt_detjson_num_images(h)

        def arr(fn, ctype, dtype):
            f = getattr(lib, fn)
            f.restype = C.POINTER(ctype)
            return np.ctypeslib.as_array(f(h), shape=(n,)).astype(dtype, copy=True) if n else np.zeros(0, dtype)
        cols = dict(image=arr('wt_detjson_image', C.c_int32, np.int32), category=arr('wt_detjson_category', C.c_int32, np.int32),
                    x=arr('wt_detjson_x', C.c_double, np.float64), y=arr('wt_detjson_y', C.c_double, np.float64),
                    w=arr('wt_detjson_w', C.c_double, np.float64), h=arr('wt_detjson_h', C.c_double, np.float64),
                    score=arr('wt_detjson_score', C.c_double, np.float64))
        cols['image_ids'] = [lib.wt_detjson_image_id(h, C.c_int32(i)).decode() for i in range(ni)]
        return cols
    finally:
        lib.wt_detjson_free(h)


def submission_columns(det_list):
    """The same columns from an already parsed JSON list (API callers / tests)."""
    ids = {}
    image = np.asarray([ids.setdefault(d['image_id'], len(ids)) for d in det_list], np.int32)
    bbox = np.asarray([d['bbox'] for d in det_list], np.float64).reshape(-1, 4)
    return dict(image=image, category=np.asarray([d['category_id'] for d in det_list], np.int32),
                x=bbox[:, 0].copy(), y=bbox[:, 1].copy(), w=bbox[:, 2].copy(), h=bbox[:, 3].copy(),
                score=np.asarray([d['score'] for d in det_list], np.float64), image_ids=list(ids))


def load_input_submissions(input_files, input_weights, min_score=0):
    """ensemble.py:78-84 on columns.  Returns (image_ids, category_ids, rows): rows = all kept detections of all inputs with
    columns image (index into image_ids), category, input (file index), score (already weighted), x, y, w, h - in input-file
    order.  Image ids come in first-appearance order (the reference iterates a set: its output order depends on
    PYTHONHASHSEED); category ids ascending."""
    subs = [read_submission(f) for f in input_files]
    return merge_inputs(subs, input_weights, min_score)


def merge_inputs(subs, input_weights, min_score=0):
    category_ids = sorted(set(int(c) for s in subs for c in np.unique(s['category'])))
    image_index = {}
    parts = []
    for k, (s, weight) in enumerate(zip(subs, input_weights)):
        score = s['score'] * weight
        keep = (s['w'] > 0) & (s['h'] > 0) & (score >= min_score)          # ensemble.py:41,46
        local = np.asarray([image_index.setdefault(i, len(image_index)) for i in s['image_ids']], np.int64)
        parts.append(dict(image=local[s['image'][keep]], category=s['category'][keep], input=np.full(int(keep.sum()), k, np.int32),
                          score=score[keep], x=s['x'][keep], y=s['y'][keep], w=s['w'][keep], h=s['h'][keep]))
    rows = {k: np.concatenate([p[k] for p in parts]) if parts else np.zeros(0) for k in ('image', 'category', 'input', 'score', 'x', 'y', 'w', 'h')}
    # compact the image list to the images that kept at least one row, in first-appearance order of those rows
    names = [None] * len(image_index)
    for name, i in image_index.items():
        names[i] = name
    present, first = np.unique(rows['image'], return_index=True)
    order = present[np.argsort(first, kind='stable')]
    remap = np.full(len(names), -1, np.int64)
    remap[order] = np.arange(len(order))
    rows['image'] = remap[rows['image'].astype(np.int64)]
    image_ids = [names[i] for i in order]
    return image_ids, category_ids, rows


def pack_groups(n_images, category_ids, rows, k_inputs, image_lo=0, image_hi=None):
    """(image, category) groups of images [image_lo, image_hi) in the CSR layout of wt_ensemble_groups: inside a group the rows
    of input 0 come first, then input 1, ... each in file order (ensemble.py:52-55 + np.vstack of tta.py:12)."""
    image_hi = n_images if image_hi is None else image_hi
    ncat = len(category_ids)
    cat_rank = np.full(max(category_ids) + 2 if category_ids else 1, -1, np.int64)
    cat_rank[np.asarray(category_ids, np.int64)] = np.arange(ncat)
    sel = np.nonzero((rows['image'] >= image_lo) & (rows['image'] < image_hi))[0]
    group = (rows['image'][sel] - image_lo) * ncat + cat_rank[rows['category'][sel]]
    order = np.lexsort((sel, rows['input'][sel], group))          # group, then input file, then file order
    sel, group = sel[order], group[order]
    G = (image_hi - image_lo) * ncat
    dets5 = np.stack([rows[c][sel] for c in ('score', 'x', 'y', 'w', 'h')], axis=1).astype(np.float64) if len(sel) else np.zeros((0, 5))
    offsets = np.zeros(G + 1, np.int64)
    np.cumsum(np.bincount(group, minlength=G), out=offsets[1:])
    sizes = np.bincount(group * k_inputs + rows['input'][sel], minlength=G * k_inputs).astype(np.int32).reshape(G, k_inputs)
    return dict(dets5=np.ascontiguousarray(dets5), group_offsets=offsets, input_sizes=sizes, n_groups=G, image_lo=image_lo, ncat=ncat)


def merge_groups(packed, k_inputs, method, iou_thresh, soft_nms_cut):
    """wt_ensemble_groups_host on one packed block -> (out5, counts)."""
    d = packed['dets5']
    G = packed['n_groups']
    out5 = np.zeros((len(d) + 1, 5), dtype=np.float64)
    counts = np.zeros(G + 1, dtype=np.int64)
    if G:
        rc = _lib.lib().wt_ensemble_groups_host(
            _lib.ptr(d), _lib.ptr(packed['group_offsets']), _lib.ptr(packed['input_sizes']), C.c_int64(G), C.c_int(k_inputs),
            C.c_int(METHODS[method]), C.c_double(iou_thresh), C.c_double(soft_nms_cut), _lib.ptr(out5), _lib.ptr(counts))
        _lib.check(rc, 'wt_ensemble_groups_host')
    return out5[:len(d)], counts[:G]


def output_rows(packed, category_ids, out5, counts, min_score):
    """ensemble.py:59-63 on columns: score > min_score, bbox.astype(int) (truncation), round(score, 5); rows in group order."""
    off = packed['group_offsets'][:-1]
    G = packed['n_groups']
    idx = np.concatenate([np.arange(o, o + c) for o, c in zip(off.tolist(), counts.tolist())]) if G and counts.sum() else np.zeros(0, np.int64)
    group = np.repeat(np.arange(G), counts)
    s = out5[idx, 0]
    keep = s > min_score
    idx, group, s = idx[keep], group[keep], s[keep]
    score = np.asarray([round(v, 5) for v in s.tolist()], dtype=np.float64)
    bbox = np.trunc(out5[idx, 1:5]).astype(np.int64).reshape(-1, 4)
    cats = np.asarray(category_ids, np.int32)
    return dict(image=(group // packed['ncat'] + packed['image_lo']).astype(np.int32), category=cats[group % packed['ncat']] if len(group) else np.zeros(0, np.int32),
                bbox=bbox, score=score)


def ensemble_columns(image_ids, category_ids, rows, k_inputs, method='weighted_fusion', iou_thresh=0.5, soft_nms_cut=1.0,
                     min_score=0.0, merge_fn=None):
    """All groups -> output columns; with torch.distributed initialised every rank merges the groups of its contiguous block
    of images and rank 0 receives all rows (None elsewhere).  merge_fn (tests: the CPU oracle) replaces the HIP call."""
    from .. import distributed as D
    w, r = D.world()
    lo, hi = D.contiguous_split(len(image_ids), w)[r]
    packed = pack_groups(len(image_ids), category_ids, rows, k_inputs, lo, hi)
    out5, counts = (merge_fn or merge_groups)(packed, k_inputs, method, iou_thresh, soft_nms_cut)
    return D.gather_columns_rank0(output_rows(packed, category_ids, out5, counts, min_score))


def ensemble_all(image_ids, category_ids, input_detections, method='weighted_fusion', iou_thresh=0.5,
                 soft_nms_cut=1.0, min_score=0.0):
    """Dict API (the reference's `ensemble(image_id, detections, category_ids)` for all images at once, ensemble.py:50-64,
    144-157): input_detections = [convert_submission(...) per input]; returns the output JSON rows."""
    index = {k: i for i, k in enumerate(image_ids)}
    cols = {k: [] for k in ('image', 'category', 'input', 'score', 'x', 'y', 'w', 'h')}
    for k, det in enumerate(input_detections):
        for image_id, per_cat in det.items():
            if image_id not in index:
                continue
            for category_id, lst in per_cat.items():
                for s, x, y, w, h in lst:
                    cols['image'].append(index[image_id]); cols['category'].append(category_id); cols['input'].append(k)
                    cols['score'].append(s); cols['x'].append(x); cols['y'].append(y); cols['w'].append(w); cols['h'].append(h)
    rows = dict(image=np.asarray(cols['image'], np.int64), category=np.asarray(cols['category'], np.int32),
                input=np.asarray(cols['input'], np.int32),
                **{c: np.asarray(cols[c], np.float64) for c in ('score', 'x', 'y', 'w', 'h')})
    out = ensemble_columns(image_ids, list(category_ids), rows, len(input_detections), method, iou_thresh, soft_nms_cut, min_score)
    if out is None:
        return None
    return [{'image_id': image_ids[i], 'category_id': int(c), 'bbox': b.tolist(), 'score': float(s)}
            for i, c, b, s in zip(out['image'].tolist(), out['category'].tolist(), out['bbox'], out['score'].tolist())]


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter,
                                     fromfile_prefix_chars='@')
    parser.add_argument('inputs', type=str, nargs='+', help='input json files')
    parser.add_argument('-o', '--output', type=str, help='output json file')
    parser.add_argument('-m', '--method', choices=("weighted_fusion", "nms", "soft_nms"), default="weighted_fusion",
                        help='method to merge bbox detections')
    parser.add_argument('--iou-thresh', type=float, default=0.5, help='IOU threshold for merging bboxes')
    parser.add_argument('--soft-nms-cut', type=float, default=1.0, help='cutout IoU threshold for soft nms')
    parser.add_argument('--min-score', type=float, default=0, help='minimal score to keep')
    parser.add_argument('-j', '--jobs', type=int, default=1,
                        help='accepted for compatibility: groups are merged in parallel on the GPU; use torchrun for N GPUs')
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))

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
            input_files, input_weights = zip(*load_yml_input_and_weight(yaml.safe_load(fp)))
        print(input_files, input_weights)

    assert len(input_files) > 1
    if rank == 0:
        print('input files:', input_files)
    if not input_weights:
        input_weights = [1] * len(input_files)     # the reference crashes here; intended value
    top = max(input_weights)
    input_weights = [w / top for w in input_weights]
    if rank == 0:
        print('weights', input_weights)

    output_file = Path(args.output)
    output_file.parent.mkdir(parents=True, exist_ok=True)
    if output_file.exists():
        raise RuntimeError(f"output file {output_file} exists!")

    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get('WT_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group(backend)
    image_ids, category_ids, rows = load_input_submissions(input_files, input_weights, args.min_score)
    if rank == 0:
        print('No. Images:', len(image_ids))
        print('No. categories:', len(category_ids))
    out = ensemble_columns(image_ids, category_ids, rows, len(input_files), args.method, args.iou_thresh, args.soft_nms_cut,
                           args.min_score)
    if rank == 0:
        from .export import write_detections_json
        write_detections_json(output_file, image_ids, out)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
