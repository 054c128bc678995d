"""The CPU oracle (oracle/softnms_oracle.c) against golden vectors produced by importing the reference's own
detnet.utils.box_utils.nms / detnet.nn.tta.{nms_detections,merge_detections} / detnet.ensemble.ensemble
(tests/golden/ensemble_*, generator oracle/gen_golden_ensemble.py).  float64, bit-exact."""
import json
import os

import numpy as np
import pytest


def assert_json_rows_equal(got, exp):
    """image ids, categories and int boxes exact; score equal up to ONE unit of the 5th decimal: the fixture was
    generated under numpy 2 where round(np.float64, 5) is numpy's scale-rint-unscale, while the reference's numpy
    1.18 (and our host code) use Python's correctly rounded float.__round__ - they differ on x.xxxxx5 half-ways."""
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert (g['image_id'], g['category_id'], g['bbox']) == (e['image_id'], e['category_id'], e['bbox'])
        assert abs(g['score'] - e['score']) <= 1.0000001e-5


def _cases(z, prefix):
    return sorted({k.split('_')[0] for k in z.files if k.startswith(prefix)})


def test_nms_detections_soft_bit_exact(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, 'ensemble_g1_softnms.npz'))
    names = _cases(z, 'case')
    assert len(names) >= 40
    for c in names:
        sizes = z[c + '_sizes']
        thr, cut = z[c + '_params']
        off = np.cumsum(np.concatenate([[0], sizes]))
        groups = [z[c + '_in'][off[i]:off[i + 1]] for i in range(len(sizes))]
        got = oracle.nms_detections(groups, iou_thresh=thr, soft=True, soft_nms_cut=cut)
        exp = z[c + '_out']
        assert got.shape == exp.shape, c
        assert np.array_equal(got, exp), (c, np.abs(got - exp).max())


def test_raw_softnms_keep_and_scores_bit_exact(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, 'ensemble_g1_softnms.npz'))
    names = _cases(z, 'raw')
    assert len(names) >= 7
    for c in names:
        thr, cut, conf, top_k = z[c + '_params']
        keep, sc = oracle.softnms(z[c + '_boxes'], z[c + '_scores'], thr, cut, conf, int(top_k))
        assert np.array_equal(keep, z[c + '_keep']), c
        assert np.array_equal(sc, z[c + '_out']), c


def test_merge_detections_bit_exact(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, 'ensemble_g3_fusion.npz'))
    for c in _cases(z, 'case'):
        sizes = z[c + '_sizes']
        off = np.cumsum(np.concatenate([[0], sizes]))
        groups = [z[c + '_in'][off[i]:off[i + 1]] for i in range(len(sizes))]
        got = oracle.merge_detections(groups, nms_thresh=float(z[c + '_thr']))
        exp = z[c + '_out']
        assert got.shape == exp.shape, c
        assert np.array_equal(got, exp), (c, np.abs(got - exp).max())


@pytest.mark.parametrize('method', ['soft_nms', 'weighted_fusion'])
def test_ensemble_json_level(oracle, golden_dir, method):
    from waymo_2d_tracking_amd.detnet import ensemble as E
    exp = json.load(open(os.path.join(golden_dir, 'ensemble_g2_expected.json')))
    subs = [json.load(open(os.path.join(golden_dir, 'ensemble_g2_input%d.json' % i))) for i in range(3)]
    image_ids, category_ids, rows = E.merge_inputs([E.submission_columns(s) for s in subs], exp['weights'], exp['min_score'])
    order = np.argsort(image_ids)                                  # the fixture lists images in sorted order
    rank = np.empty(len(order), np.int64); rank[order] = np.arange(len(order))
    rows['image'] = rank[rows['image']]
    image_ids = [image_ids[i] for i in order]
    packed = E.pack_groups(len(image_ids), category_ids, rows, len(subs))
    m = {'weighted_fusion': 0, 'nms': 1, 'soft_nms': 2}[method]
    out5, counts = oracle.ensemble_groups(packed['dets5'], packed['group_offsets'], packed['input_sizes'],
                                          len(subs), m, exp['iou_thresh'], exp['soft_nms_cut'])
    cols = E.output_rows(packed, category_ids, out5[:len(packed['dets5'])], counts, exp['min_score'])
    got = [{'image_id': image_ids[i], 'category_id': int(c), 'bbox': b, 'score': s} for i, c, b, s in
           zip(cols['image'].tolist(), cols['category'].tolist(), cols['bbox'].tolist(), cols['score'].tolist())]
    assert_json_rows_equal(got, exp['outputs'][method])
