"""HIP ensemble kernels (through the C ABI of libwaymotrack.so) against the golden vectors generated from the
reference and against the CPU oracle on seeded inputs.  float64: bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from test_oracle_ensemble import _cases, assert_json_rows_equal

pytestmark = pytest.mark.gpu


def test_nms_detections_golden_bit_exact(golden_dir):
    from waymo_2d_tracking_amd.detnet.nn.tta import nms_detections
    z = np.load(os.path.join(golden_dir, 'ensemble_g1_softnms.npz'))
    for c in _cases(z, 'case'):
        sizes = z[c + '_sizes']
        thr, cut = z[c + '_params']
        off = np.cumsum(np.concatenate([[0], sizes]))
        groups = [z[c + '_in'][off[i]:off[i + 1]] for i in range(len(sizes))]
        got = nms_detections(groups, iou_thresh=thr, soft=True, soft_nms_cut=cut)
        exp = z[c + '_out']
        assert got.shape == exp.shape, c
        assert np.array_equal(got, exp), (c, np.abs(got - exp).max())


def test_raw_nms_golden_bit_exact(golden_dir):
    from waymo_2d_tracking_amd.detnet.utils.box_utils import nms
    z = np.load(os.path.join(golden_dir, 'ensemble_g1_softnms.npz'))
    for c in _cases(z, 'raw'):
        thr, cut, conf, top_k = z[c + '_params']
        keep, sc = nms(torch.from_numpy(z[c + '_boxes']), torch.from_numpy(z[c + '_scores']), overlap=thr,
                       top_k=int(top_k), soft=True, conf_thresh=conf, soft_nms_cut=cut)
        assert keep == z[c + '_keep'].tolist(), c
        assert np.array_equal(sc.numpy(), z[c + '_out']), c


def test_merge_detections_golden_bit_exact(golden_dir):
    from waymo_2d_tracking_amd.detnet.nn.tta import merge_detections
    z = np.load(os.path.join(golden_dir, 'ensemble_g3_fusion.npz'))
    for c in _cases(z, 'case'):
        sizes = z[c + '_sizes']
        off = np.cumsum(np.concatenate([[0], sizes]))
        groups = [z[c + '_in'][off[i]:off[i + 1]] for i in range(len(sizes))]
        got = merge_detections(groups, nms_thresh=float(z[c + '_thr']))
        exp = z[c + '_out']
        assert got.shape == exp.shape, c
        assert np.array_equal(got, exp), (c, np.abs(got - exp).max())


@pytest.mark.parametrize('method', ['soft_nms', 'weighted_fusion'])
def test_ensemble_json_golden(golden_dir, method):
    from waymo_2d_tracking_amd.detnet import ensemble as E
    exp = json.load(open(os.path.join(golden_dir, 'ensemble_g2_expected.json')))
    subs = [json.load(open(os.path.join(golden_dir, 'ensemble_g2_input%d.json' % i))) for i in range(3)]
    dets = [E.convert_submission(s, w, exp['min_score']) for s, w in zip(subs, exp['weights'])]
    image_ids = sorted(set(sum([list(d.keys()) for d in dets], [])))
    category_ids = sorted(set(sum([[d['category_id'] for d in s] for s in subs], [])))
    got = E.ensemble_all(image_ids, category_ids, dets, method, exp['iou_thresh'], exp['soft_nms_cut'], exp['min_score'])
    assert_json_rows_equal(got, exp['outputs'][method])


def _random_groups(rng, n_groups, max_n, k):
    from waymo_2d_tracking_amd import synthetic as syn
    rows, off, sizes = [], [0], []
    for g in range(n_groups):
        n = int(rng.integers(0, max_n + 1))
        per = n // k
        gs = syn.ensemble_group(rng, per, k) if per else [np.zeros((0, 5))] * k
        if g % 7 == 3:
            gs = [gs[0]] + [x[: len(x) // 2] for x in gs[1:]]          # ragged inputs
        for x in gs:
            rows.append(x); sizes.append(len(x))
        off.append(off[-1] + sum(len(x) for x in gs))
    d = np.ascontiguousarray(np.vstack(rows)) if rows else np.zeros((0, 5))
    return d, np.asarray(off, np.int64), np.asarray(sizes, np.int32).reshape(n_groups, k)


@pytest.mark.parametrize('method', [0, 1, 2])
@pytest.mark.parametrize('max_n,k', [(120, 2), (400, 4), (1700, 13)])
def test_batched_groups_vs_oracle(oracle, method, max_n, k):
    """Many ragged groups in one launch; the (1700, 13) case exceeds the LDS budget -> global scratch path."""
    import ctypes as C
    from waymo_2d_tracking_amd import _lib
    rng = np.random.default_rng(1000 + max_n + method)
    n_groups = 60 if max_n <= 400 else 6
    d, off, sizes = _random_groups(rng, n_groups, max_n, k)
    exp, exp_cnt = oracle.ensemble_groups(d, off, sizes, k, method, 0.5, 0.9)
    out = np.zeros((len(d) + 1, 5)); cnt = np.zeros(n_groups + 1, np.int64)
    _lib.check(_lib.lib().wt_ensemble_groups_host(_lib.ptr(d), _lib.ptr(off), _lib.ptr(sizes), C.c_int64(n_groups),
                                                  C.c_int(k), C.c_int(method), C.c_double(0.5), C.c_double(0.9),
                                                  _lib.ptr(out), _lib.ptr(cnt)), 'wt_ensemble_groups_host')
    assert np.array_equal(cnt[:n_groups], exp_cnt)
    for g in range(n_groups):
        a, b = int(off[g]), int(off[g]) + int(exp_cnt[g])
        assert np.array_equal(out[a:b], exp[a:b]), (g, np.abs(out[a:b] - exp[a:b]).max())


def test_hard_nms_vs_oracle(oracle):
    from waymo_2d_tracking_amd.detnet.utils.box_utils import nms
    from waymo_2d_tracking_amd import synthetic as syn
    rng = np.random.default_rng(77)
    for n, top_k in ((1, 0), (40, 0), (300, 0), (300, 50)):
        g = np.concatenate(syn.ensemble_group(rng, max(1, n // 3), 3))[:n]
        boxes = np.stack([g[:, 1], g[:, 2], g[:, 1] + g[:, 3], g[:, 2] + g[:, 4]], axis=1)
        keep, sc = nms(torch.from_numpy(boxes), torch.from_numpy(g[:, 0].copy()), overlap=0.5, top_k=top_k)
        ek, es = oracle.hardnms(boxes, g[:, 0], 0.5, top_k)
        assert keep.tolist() == ek.tolist()
        assert np.array_equal(sc.numpy(), es)


def test_full_size_properties():
    """Config-4 sized batch (K=13 inputs, 198 images): size-independent invariants of linear soft-NMS."""
    import ctypes as C
    from waymo_2d_tracking_amd import _lib
    rng = np.random.default_rng(4)
    d, off, sizes = _random_groups(rng, 198 * 3, 1300, 13)
    G = len(off) - 1
    out = np.zeros((len(d) + 1, 5)); cnt = np.zeros(G + 1, np.int64)
    _lib.check(_lib.lib().wt_ensemble_groups_host(_lib.ptr(d), _lib.ptr(off), _lib.ptr(sizes), C.c_int64(G), C.c_int(13),
                                                  C.c_int(2), C.c_double(0.5), C.c_double(0.9), _lib.ptr(out), _lib.ptr(cnt)),
               'wt_ensemble_groups_host')
    assert np.array_equal(cnt[:G], np.diff(off))                     # every box survives (conf_thresh = 0)
    for g in range(0, G, 37):
        a, b = int(off[g]), int(off[g + 1])
        if b - a < 2:
            continue
        src, res = d[a:b], out[a:b]
        order = np.argsort(-src[:, 0], kind='stable')
        assert np.all(res[:, 0] <= src[order, 0] + 0)               # scores only decay
        assert res[0, 0] == src[order[0], 0]                        # the top box is untouched
        np.testing.assert_allclose(res[:, 3:5], src[order, 3:5], rtol=1e-12)   # geometry is preserved, in rank order


def test_soft_nms_groups_with_nan_negative_and_degenerate_boxes(oracle):
    """Groups the dependency-free kernel must hand to the serial one (NaN / negative scores), zero-area boxes,
    duplicates and exact score ties - all still bit-identical to the oracle."""
    import ctypes as C
    from waymo_2d_tracking_amd import _lib
    from waymo_2d_tracking_amd import synthetic as syn
    rng = np.random.default_rng(31)
    rows, off = [], [0]
    for g in range(12):
        gs = np.concatenate(syn.ensemble_group(rng, 30, 3))
        if g % 4 == 0:
            gs[5, 0] = np.nan
        if g % 4 == 1:
            gs[7, 0] = -0.25
        if g % 4 == 2:
            gs[3, 3] = 0.0                      # zero width
            gs[9] = gs[8]                       # exact duplicate (score tie, IoU 1)
        rows.append(gs); off.append(off[-1] + len(gs))
    d = np.ascontiguousarray(np.vstack(rows)); off = np.asarray(off, np.int64)
    sizes = np.full((12, 3), 30, np.int32)
    for cut in (0.9, 1.0):
        exp, exp_cnt = oracle.ensemble_groups(d, off, sizes, 3, 2, 0.5, cut)
        out = np.zeros((len(d) + 1, 5)); cnt = np.zeros(13, np.int64)
        _lib.check(_lib.lib().wt_ensemble_groups_host(_lib.ptr(d), _lib.ptr(off), _lib.ptr(sizes), C.c_int64(12), C.c_int(3),
                                                      C.c_int(2), C.c_double(0.5), C.c_double(cut), _lib.ptr(out), _lib.ptr(cnt)),
                   'wt_ensemble_groups_host')
        assert np.array_equal(cnt[:12], exp_cnt)
        for g in range(12):
            a, b = int(off[g]), int(off[g]) + int(exp_cnt[g])
            assert np.array_equal(out[a:b], exp[a:b], equal_nan=True), g


def test_ensemble_cli_end_to_end(golden_dir, tmp_path):
    """python -m detnet.ensemble drop-in: files + yml weights in, JSON out (order-insensitive vs the reference rows)."""
    import yaml
    from waymo_2d_tracking_amd.detnet import ensemble as E
    exp = json.load(open(os.path.join(golden_dir, 'ensemble_g2_expected.json')))
    yml = tmp_path / 'weights.yml'
    yml.write_text(yaml.safe_dump({golden_dir: {'ensemble_g2_input%d.json' % i: w for i, w in enumerate(exp['weights'])}}))
    for method in ('soft_nms', 'weighted_fusion'):
        out = tmp_path / (method + '.json')
        E.main([str(yml), '-o', str(out), '-m', method, '--iou-thresh', str(exp['iou_thresh']), '--soft-nms-cut',
                str(exp['soft_nms_cut']), '--min-score', str(exp['min_score'])])
        got = json.load(open(out))
        key = lambda r: (r['image_id'], r['category_id'], tuple(r['bbox']))
        g = sorted(got, key=key); e = sorted(exp['outputs'][method], key=key)
        assert len(g) == len(e)
        for a, b in zip(g, e):
            assert key(a) == key(b) and abs(a['score'] - b['score']) <= 1.0000001e-5


def test_all_groups_empty_and_single_box(oracle):
    import ctypes as C
    from waymo_2d_tracking_amd import _lib
    for method in (0, 1, 2):
        d = np.zeros((0, 5)); off = np.zeros(5, np.int64); sizes = np.zeros((4, 2), np.int32)
        out = np.zeros((1, 5)); cnt = np.full(5, -1, np.int64)
        _lib.check(_lib.lib().wt_ensemble_groups_host(_lib.ptr(d), _lib.ptr(off), _lib.ptr(sizes), C.c_int64(4), C.c_int(2),
                                                      C.c_int(method), C.c_double(0.5), C.c_double(0.9), _lib.ptr(out), _lib.ptr(cnt)),
                   'wt_ensemble_groups_host')
        assert cnt[:4].tolist() == [0, 0, 0, 0]
        d = np.array([[0.7, 10.0, 20.0, 30.0, 40.0]]); off = np.array([0, 1], np.int64); sizes = np.array([[1, 0]], np.int32)
        out = np.zeros((2, 5)); cnt = np.zeros(2, np.int64)
        _lib.check(_lib.lib().wt_ensemble_groups_host(_lib.ptr(d), _lib.ptr(off), _lib.ptr(sizes), C.c_int64(1), C.c_int(2),
                                                      C.c_int(method), C.c_double(0.5), C.c_double(0.9), _lib.ptr(out), _lib.ptr(cnt)),
                   'wt_ensemble_groups_host')
        exp, ec = oracle.ensemble_groups(d, off, sizes, 2, method, 0.5, 0.9)
        assert cnt[0] == ec[0] == 1 and np.array_equal(out[:1], exp[:1])
