"""TEST INFRASTRUCTURE - generates tests/golden/metric_g8.json by running the REFERENCE's evaluation code on a toy data set:
/root/reference/detnet/data/metric.py (compute_truth_and_false_positive :143-270, update_tp_fp :274-301, voc_ap :107-140,
f2_score :304-320, filter_detections :9-28, evaluate_detections :323-401), the Waymo metric of /root/reference/data/__init__.py:
9-84 (metric_fun: IoU 0.7 for vehicle, 0.5 otherwise) and the ground-truth conversion COCOAnnotationTransform of
/root/reference/detnet/data/coco.py:54-118.  The functions are loaded from the reference files AT GENERATION TIME (their
modules cannot be imported as packages here: pycocotools / torch._six / collections.Iterable); nothing of them is stored in the
repo - only the toy inputs and the numbers they produced.

    python oracle/gen_golden_metric.py            # needs /root/reference (this container only)

Detections are listed per image in DESCENDING score order (detectron2's output order): with unsorted input the reference pairs
confidences and TP/FP flags wrongly, which the native tool does not reproduce; one extra case records
the reference's output on shuffled input for documentation.
"""
import ast
import json
import os
import sys
import types
import warnings

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'metric_g8.json')


def load_symbols(path, names, namespace):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            code = compile(ast.Module(body=[node], type_ignores=[]), path, 'exec')
            exec(code, namespace)
    return namespace


class FakeImage(object):
    def __init__(self, w, h):
        self.size = (w, h)


class FakeDataset(list):
    classnames = ['background', 'vehicle', 'pedestrian', 'sign', 'cyclist']


def main():
    if not os.path.isdir(REF):
        sys.exit('needs ' + REF)
    import pandas as pd
    import torch
    ns = dict(np=np, torch=torch, pd=pd, warnings=warnings, pickle=__import__('pickle'), rle_decode=None,
              get_num_workers=lambda n: 1, multiprocessing=None)
    load_symbols(os.path.join(REF, 'detnet/data/metric.py'),
                 {'filter_detections', 'voc_ap', 'compute_truth_and_false_positive', 'update_tp_fp', 'f2_score', 'evaluate_detections', 'voc_eval'}, ns)
    load_symbols(os.path.join(REF, 'data/__init__.py'), {'metric_fun'}, ns)
    load_symbols(os.path.join(REF, 'detnet/data/coco.py'), {'COCOAnnotationTransform'}, ns)

    rng = np.random.default_rng(8)
    sizes = [(1920, 1280), (1920, 886), (1920, 1280), (1920, 1280), (1920, 886), (1920, 1280)]
    images, annotations, predictions = [], [], {}
    ann_id = 1
    for i, (w, h) in enumerate(sizes):
        image_id = 'seg%d/%d/%s' % (i // 3, 1550000000000000 + 100000 * i, 'FRONT' if h == 1280 else 'SIDE_LEFT')
        images.append(dict(id=image_id, width=w, height=h, file_name=image_id + '.jpg'))
        gts = []
        n_gt = [6, 0, 9, 3, 5, 7][i]
        for _ in range(n_gt):
            cat = int(rng.choice([1, 2, 4], p=[0.6, 0.3, 0.1]))
            bw, bh = float(rng.uniform(12, 400)), float(rng.uniform(12, 300))
            x, y = float(rng.uniform(0, w - bw)), float(rng.uniform(0, h - bh))
            box = [round(x, 1), round(y, 1), round(bw, 1), round(bh, 1)]
            annotations.append(dict(id=ann_id, image_id=image_id, category_id=cat, bbox=box))
            gts.append((cat, box))
            ann_id += 1
        if i == 2:                                   # a duplicated ground-truth box (np.unique in the reference removes it)
            annotations.append(dict(id=ann_id, image_id=image_id, category_id=gts[0][0], bbox=list(gts[0][1]))); ann_id += 1
        # detections: jittered copies of the ground truth (some good, some poor), duplicates and clutter, per class
        per_class = [[] for _ in range(4)]
        for cat, (x, y, bw, bh) in gts:
            if rng.uniform() < 0.15:
                continue                              # missed object
            for rep in range(int(rng.choice([1, 1, 2]))):
                j = rng.normal(0, [2.0, 12.0][int(rng.uniform() < 0.3)], 4)
                cx, cy = (x + bw / 2 + j[0]) / w, (y + bh / 2 + j[1]) / h
                per_class[cat - 1].append([float(rng.uniform(0.05, 1)), cx, cy, (bw + j[2]) / w, (bh + j[3]) / h])
        for _ in range(int(rng.integers(0, 5))):      # clutter (also in the class without ground truth)
            c = int(rng.integers(0, 4))
            per_class[c].append([float(rng.uniform(0.02, 0.6)), float(rng.uniform(0.1, 0.9)), float(rng.uniform(0.1, 0.9)),
                                 float(rng.uniform(0.01, 0.2)), float(rng.uniform(0.01, 0.2))])
        dets = []
        for c in range(4):
            d = np.asarray(per_class[c], dtype=np.float32).reshape(-1, 5)
            d = d[np.argsort(-d[:, 0], kind='stable')]
            dets.append(d)
        if i == 4:
            dets = [np.zeros((0, 5), np.float32) for _ in range(4)]     # an image without detections
        predictions[image_id] = dets

    def dataset_from(images, annotations):
        tf = ns['COCOAnnotationTransform'](None, None, False)
        ds = FakeDataset()
        by_img = {}
        for a in annotations:
            by_img.setdefault(a['image_id'], []).append(a)
        for im in sorted(images, key=lambda d: d['id']):
            s = dict(image_id=str(im['id']))
            s.update(tf(by_img.get(im['id'], []), FakeImage(im['width'], im['height'])))
            ds.append(s)
        return ds

    def run(preds, threshold=0.01):
        ds = dataset_from(images, annotations)

        class Dets(dict):
            classnames = ['vehicle', 'pedestrian', 'sign', 'cyclist']

            def __iter__(self):
                return iter(self.items())
        ev = ns['evaluate_detections'](Dets(preds), ds, num_processes=1, print_out=False, metric_fun=ns['metric_fun'], threshold=threshold)
        out = {}
        for cls, v in ev.items():
            out[cls] = {k: (None if isinstance(x, float) and np.isnan(x) else float(x)) for k, x in v.items()}
        # also the plain VOC evaluation of metric.py (IoU 0.5 / 0.75, size buckets at 0.5) per class
        voc = {}
        for i, cls in enumerate(Dets.classnames):
            per = {k: ns['filter_detections'](v, i, threshold) for k, v in preds.items()}
            r = ns['voc_eval'](per, ds, cls, ovthresh=(0.5, 0.75), size_ovthreshs=0.5)
            voc[cls] = {k: (None if isinstance(x, float) and np.isnan(x) else float(x)) for k, x in r.items()}
        return out, voc

    waymo, voc = run(predictions)
    shuffled = {k: [d[rng.permutation(len(d))] for d in v] for k, v in predictions.items()}
    waymo_shuffled, _ = run(shuffled)
    doc = dict(source='detnet/data/metric.py:9-401 + data/__init__.py:9-84 + detnet/data/coco.py:54-118 run on the toy set below',
               classnames=['vehicle', 'pedestrian', 'sign', 'cyclist'], threshold=0.01,
               annotations=dict(images=images, annotations=annotations,
                                categories=[dict(id=i + 1, name=n) for i, n in enumerate(['vehicle', 'pedestrian', 'sign', 'cyclist'])]),
               predictions={k: [d.tolist() for d in v] for k, v in predictions.items()},
               expected_waymo=waymo, expected_voc=voc, reference_on_shuffled_input_D10=waymo_shuffled,
               voc_ap_cases=[dict(rec=r, prec=p, ap=float(ns['voc_ap'](np.asarray(r), np.asarray(p))),
                                  ap07=float(ns['voc_ap'](np.asarray(r), np.asarray(p), True)))
                             for r, p in ([[0.1, 0.2, 0.2, 0.5, 1.0], [1.0, 1.0, 0.66, 0.75, 0.5]], [[], []], [[0.0, 0.0], [0.0, 0.0]],
                                          [[0.25, 0.5, 0.75], [1.0, 0.5, 0.6]])])
    json.dump(doc, open(OUT, 'w'), separators=(',', ':'))
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')
    print(json.dumps(waymo, indent=0)[:600])
    print('shuffled (D-10):', {k: v.get('ap') for k, v in waymo_shuffled.items() if isinstance(v, dict)})


if __name__ == '__main__':
    main()
