"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/ensemble_*.{json,npz} by importing the REFERENCE's
own soft-NMS / ensemble code (/root/reference/detnet/{ensemble.py,nn/tta.py,utils/box_utils.py}) in this
container (python3.10, torch CPU float64) - a true oracle for SURVEY rows a10-a15.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_ensemble.py

torchvision / shapely are absent: they are stubbed (only the hard-NMS and rotated-box branches use them,
neither is exercised here).  Only input/output DATA is written; no reference source is copied.
"""
import argparse
import importlib.util
import json
import os
import sys
import types
from functools import partial

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLDEN = os.path.join(REPO, 'tests', 'golden')
REF = '/root/reference'


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    stub('torchvision'); stub('torchvision.ops', nms=None); stub('torchvision.utils')
    stub('shapely'); stub('shapely.geometry', asPolygon=None)
    sys.path.insert(0, REF)
    import detnet  # noqa: F401  namespace package
    stub('detnet.nn').__path__ = [REF + '/detnet/nn']            # bypass nn/__init__.py (torch._six)
    stub('detnet.trainer').__path__ = [REF + '/detnet/trainer']  # bypass trainer/__init__.py
    from detnet.nn.tta import nms_detections, merge_detections
    from detnet.utils.box_utils import nms
    import detnet.ensemble as E
    return nms_detections, merge_detections, nms, E


def main():
    import torch
    torch.set_num_threads(1)
    nms_detections, merge_detections, nms, E = import_reference()
    syn = _load('synthetic', os.path.join(REPO, 'waymo_2d_tracking_amd', 'synthetic.py'))
    os.makedirs(GOLDEN, exist_ok=True)
    rng = np.random.default_rng(424242)

    # ---------------- G1: nms(soft=True) and nms_detections on (n_i,5) lists ----------------
    g1 = {}
    case = 0
    for n in (0, 1, 2, 5, 37, 100, 400):
        for (thr, cut) in ((0.5, 1.0), (0.5, 0.9), (0.3, 0.7)):
            for k in (1, 3):
                if n == 0:
                    groups = [np.zeros((0, 5)) for _ in range(k)]
                else:
                    per = max(1, n // k)
                    groups = syn.ensemble_group(rng, per, k)
                    # to centre form [score, cx, cy, w, h] like ensemble.py:55
                    groups = [np.concatenate([g[:, :1], g[:, 1:3] + g[:, 3:5] / 2, g[:, 3:5]], axis=1) for g in groups]
                total = sum(len(g) for g in groups)
                if total == 0:
                    out = np.zeros((0, 5))    # reference: np.vstack of empties works, nms on empty returns empty
                    try:
                        out = nms_detections([g.copy() for g in groups], iou_thresh=thr, soft=True, soft_nms_cut=cut)
                    except Exception as e:   # record the reference behaviour
                        g1['case%02d_error' % case] = np.array(str(type(e).__name__))
                else:
                    out = nms_detections([g.copy() for g in groups], iou_thresh=thr, soft=True, soft_nms_cut=cut)
                g1['case%02d_in' % case] = np.concatenate(groups) if total else np.zeros((0, 5))
                g1['case%02d_sizes' % case] = np.array([len(g) for g in groups])
                g1['case%02d_params' % case] = np.array([thr, cut])
                g1['case%02d_out' % case] = np.asarray(out, dtype=np.float64).reshape(-1, 5)
                case += 1
    # raw nms() API: keep list + scores, also conf_thresh > 0 and top_k (box_utils.py:324-327,379-381)
    for n, thr, cut, conf, top_k in ((50, 0.5, 0.9, 0.0, 0), (50, 0.5, 0.9, 0.25, 0), (80, 0.4, 1.0, 0.1, 30),
                                     (3, 0.5, 0.9, 0.0, 0), (64, 0.5, 0.9, 0.0, 0), (65, 0.5, 0.9, 0.0, 0),
                                     (129, 0.45, 0.8, 0.05, 0)):
        g = syn.ensemble_group(rng, max(1, n // 2), 2)
        g = np.concatenate(g)[:n]
        boxes = np.stack([g[:, 1], g[:, 2], g[:, 1] + g[:, 3], g[:, 2] + g[:, 4]], axis=1)
        scores = g[:, 0].copy()
        keep, new_scores = nms(torch.from_numpy(boxes), torch.from_numpy(scores), overlap=thr, top_k=top_k,
                               soft=True, conf_thresh=conf, soft_nms_cut=cut)
        g1['raw%02d_boxes' % case] = boxes
        g1['raw%02d_scores' % case] = scores
        g1['raw%02d_params' % case] = np.array([thr, cut, conf, top_k])
        g1['raw%02d_keep' % case] = np.asarray(keep, dtype=np.int64)
        g1['raw%02d_out' % case] = new_scores.numpy().astype(np.float64)
        case += 1
    np.savez_compressed(os.path.join(GOLDEN, 'ensemble_g1_softnms.npz'), **g1)

    # ---------------- G3: merge_detections (weighted fusion, tta.py:22-66) ----------------
    g3 = {}
    for c, (n, k, thr) in enumerate(((0, 2, 0.5), (1, 2, 0.5), (12, 2, 0.5), (40, 3, 0.5), (100, 4, 0.6), (25, 5, 0.3))):
        if n == 0:
            groups = [np.zeros((0, 5)), np.zeros((0, 5))]
        else:
            groups = syn.ensemble_group(rng, n, k)
            groups = [np.concatenate([g[:, :1], g[:, 1:3] + g[:, 3:5] / 2, g[:, 3:5]], axis=1) for g in groups]
            # ragged: drop a few rows from later inputs; make one input empty in one case
            groups = [g[: len(g) - (i * 3) % max(1, len(g))] if i else g for i, g in enumerate(groups)]
            if c == 3:
                groups[1] = np.zeros((0, 5))
        out = merge_detections([g.copy() for g in groups], nms_thresh=thr)
        g3['case%d_in' % c] = np.concatenate(groups)
        g3['case%d_sizes' % c] = np.array([len(g) for g in groups])
        g3['case%d_thr' % c] = np.array(thr)
        g3['case%d_out' % c] = np.asarray(out, dtype=np.float64).reshape(-1, 5)
    np.savez_compressed(os.path.join(GOLDEN, 'ensemble_g3_fusion.npz'), **g3)

    # ---------------- G2: ensemble() at JSON level (ensemble.py:31-64,78-84) ----------------
    subs = syn.ensemble_inputs_json(99, n_images=6, k_inputs=3, n_objects=40)
    # one image missing from one input, zero-size boxes, low scores
    subs[1] = [e for e in subs[1] if not e['image_id'].endswith('/%i/FRONT' % syn.frame_timestamp(2))]
    subs[0][3]['bbox'][2] = 0
    subs[2][5]['score'] = 0.001
    weights = [1.0, 0.8, 0.5]
    for i, s in enumerate(subs):
        with open(os.path.join(GOLDEN, 'ensemble_g2_input%d.json' % i), 'wt') as fp:
            json.dump(s, fp)
    expected = {}
    for method, kw in (('soft_nms', dict(iou_thresh=0.5, soft=True, soft_nms_cut=0.9)),
                       ('weighted_fusion', dict(nms_thresh=0.5))):
        E.args = argparse.Namespace(min_score=0.01)
        E.merge_func = partial(nms_detections, **kw) if method == 'soft_nms' else partial(merge_detections, **kw)
        dets = [E.convert_submission(json.loads(json.dumps(s)), w, E.args.min_score) for s, w in zip(subs, weights)]
        image_ids = sorted(set(sum([list(d.keys()) for d in dets], [])))
        category_ids = sorted(set(sum([[d['category_id'] for d in s] for s in subs], [])))
        out = []
        for image_id in image_ids:
            out += E.ensemble(image_id, [d[image_id] for d in dets], category_ids)
        expected[method] = out
    with open(os.path.join(GOLDEN, 'ensemble_g2_expected.json'), 'wt') as fp:
        json.dump({'weights': weights, 'min_score': 0.01, 'iou_thresh': 0.5, 'soft_nms_cut': 0.9,
                   'outputs': expected}, fp)
    print('G1 cases', case, 'G2 rows', {k: len(v) for k, v in expected.items()})


if __name__ == '__main__':
    main()
