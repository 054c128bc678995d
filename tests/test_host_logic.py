"""CPU tests of host-side logic: wire formats, CLI parsers, synthetic generators (no GPU, no compute kernels)."""
import json
import os

import numpy as np
import pytest


def test_load_prediction_matches_reference_fixture(golden_dir):
    """detection-JSON wire format (coco.py:229-252) against rows produced by the reference's own load_prediction."""
    from waymo_2d_tracking_amd.detnet.inference import load_prediction
    g = json.load(open(os.path.join(golden_dir, 'export_g6.json')))
    sizes = {k: (v['width'], v['height']) for k, v in g['images'].items()}
    preds = {k: [np.asarray(a, dtype=np.float32).reshape(-1, 5) for a in v] for k, v in g['predictions'].items()}
    rows = load_prediction(sizes, g['classnames'], preds)
    assert len(rows) == len(g['rows'])
    for a, b in zip(rows, g['rows']):
        assert (a['image_id'], a['category_id'], a['bbox']) == (b['image_id'], b['category_id'], b['bbox'])
        assert abs(a['score'] - b['score']) <= 1.0000001e-5      # numpy-2 vs Python round on exact half-ways
        assert all(isinstance(v, int) for v in a['bbox'])


def test_track_cli_flags_match_reference():
    from waymo_2d_tracking_amd.tracking.track import build_parser
    a = build_parser().parse_args([])
    assert a.max_age == 1 and a.min_hits == 0                                   # track.py:21-22
    assert a.score_threshold == [0.95, 0.6, 1.0, 0.9] and a.iou_threshold == [0.01, 0.01, 1.0, 0.0]   # :23-26
    b = build_parser().parse_args(['--input', 'x.json', '--output', 'y.json', '--max-age=2', '--min-hits=0',
                                   '--score-threshold=0.95,0.6,1.0,0.9', '--segment-id', 's', '--ground-truth', 'nofile'])
    assert b.max_age == 2 and b.segment_id == 's'


def test_ensemble_cli_refuses_existing_output_and_parses(tmp_path, golden_dir):
    from waymo_2d_tracking_amd.detnet import ensemble as E
    out = tmp_path / 'exists.json'
    out.write_text('[]')
    with pytest.raises(RuntimeError):                                           # ensemble.py:131-132
        E.main([os.path.join(golden_dir, 'ensemble_g2_input0.json'), os.path.join(golden_dir, 'ensemble_g2_input1.json'),
                '-o', str(out), '-m', 'soft_nms'])
    # column path: native reader == json.load, grouping == the dict path of the reference
    files = [os.path.join(golden_dir, 'ensemble_g2_input%d.json' % i) for i in range(2)]
    subs = [json.load(open(f)) for f in files]
    for f, s in zip(files, subs):
        nat, ref = E.read_submission(f), E.submission_columns(s)
        assert nat['image_ids'] == ref['image_ids']
        for k in ('image', 'category', 'x', 'y', 'w', 'h', 'score'):
            assert np.array_equal(nat[k], ref[k]), k
    image_ids, category_ids, rows = E.load_input_submissions(files, [1.0, 0.5], 0.01)
    dets = [E.convert_submission(s, w, 0.01) for s, w in zip(subs, [1.0, 0.5])]
    assert set(image_ids) == set(k for d in dets for k in d) and category_ids == [1, 2, 4]
    packed = E.pack_groups(len(image_ids), category_ids, rows, 2)
    assert packed['group_offsets'][-1] == len(packed['dets5']) == len(rows['score'])
    assert packed['input_sizes'].shape == (len(image_ids) * 3, 2) and packed['input_sizes'].sum() == len(packed['dets5'])
    for gi in (0, 4, len(image_ids) * 3 - 1):                     # rows of a group: input 0 then input 1, file order
        image_id, cat = image_ids[gi // 3], category_ids[gi % 3]
        want = [r for d in dets for r in d.get(image_id, {}).get(cat, [])]
        got = packed['dets5'][packed['group_offsets'][gi]:packed['group_offsets'][gi + 1]]
        assert np.array_equal(got, np.asarray(want, np.float64).reshape(-1, 5))
    # blocks of images (multi-GPU shards) partition the groups
    a, b = E.pack_groups(len(image_ids), category_ids, rows, 2, 0, 5), E.pack_groups(len(image_ids), category_ids, rows, 2, 5, len(image_ids))
    assert np.array_equal(np.vstack((a['dets5'], b['dets5'])), packed['dets5'])


def test_native_detection_json_writer_is_byte_identical_to_json_dump(tmp_path, golden_dir):
    """wt_detections_write_json == json.dump of the reference's row dicts (coco.py:249-251, ensemble.py:61-62) on G2 / G6
    rows and on strings that need escaping."""
    from waymo_2d_tracking_amd.detnet.export import write_detections_json
    cases = [json.load(open(os.path.join(golden_dir, 'ensemble_g2_expected.json')))['outputs']['soft_nms'],
             json.load(open(os.path.join(golden_dir, 'export_g6.json')))['rows'],
             [{'image_id': 'a"b\\c/\u00fc/\u4e2d\U0001F600\n\x7f', 'category_id': 1, 'bbox': [1, -2, 3, 4], 'score': 1e-07}], []]
    for i, rows in enumerate(cases):
        ids = list(dict.fromkeys(r['image_id'] for r in rows))
        ix = {k: j for j, k in enumerate(ids)}
        cols = dict(image=np.array([ix[r['image_id']] for r in rows], np.int32), category=np.array([r['category_id'] for r in rows], np.int32),
                    bbox=np.array([r['bbox'] for r in rows], np.int64).reshape(-1, 4), score=np.array([r['score'] for r in rows], np.float64))
        out = tmp_path / ('w%d.json' % i)
        write_detections_json(out, ids, cols)
        assert open(out).read() == json.dumps(rows), i


def test_prediction_store_and_export_rows(golden_dir):
    """Predictions (columnar trainer/predictions.py) keeps the mapping interface; export rows == load_prediction on G6."""
    from waymo_2d_tracking_amd.detnet.trainer import Predictions
    from waymo_2d_tracking_amd.detnet import export as X
    g6 = json.load(open(os.path.join(golden_dir, 'export_g6.json')))
    classnames = g6['classnames']
    p = Predictions(classnames)
    sizes = {}
    for image_id, info in g6['images'].items():
        p[image_id] = [np.asarray(d, np.float32).reshape(-1, 5) for d in g6['predictions'][image_id]]
        sizes[image_id] = (info['width'], info['height'])
    assert len(p) == len(g6['images']) and set(p.keys()) == set(g6['images'])
    first = next(iter(g6['images']))
    got = p[first]
    assert len(got) == 4 and all(np.array_equal(a, np.asarray(b, np.float32).reshape(-1, 5)) for a, b in zip(got, g6['predictions'][first]))
    assert p['missing'] is None
    rows = X.detection_rows(p, sizes)
    exp = g6['rows']
    assert [p.image_ids[i] for i in rows['image']] == [r['image_id'] for r in exp]
    assert rows['category'].tolist() == [r['category_id'] for r in exp]
    assert rows['bbox'].tolist() == [r['bbox'] for r in exp]
    assert np.abs(rows['score'] - np.array([r['score'] for r in exp])).max() <= 1.0000001e-5      # half-way rounding, see DESIGN 2
    # shards -> one store; save / open
    a, b = Predictions(classnames, p.image_ids), Predictions(classnames, p.image_ids)
    for k, (image_id, det) in enumerate(p):
        (a if k % 2 == 0 else b)[image_id] = det
    merged = Predictions.from_shards(classnames, p.image_ids, [a.shard_columns()[0], b.shard_columns()[0]], [a.tested, b.tested])
    assert len(merged) == len(p) and all(np.array_equal(x, y) for k in p.keys() for x, y in zip(merged[k], p[k]))


def test_yaml_weights_loader():
    from waymo_2d_tracking_amd.detnet.ensemble import load_yml_input_and_weight
    got = load_yml_input_and_weight({'a': {'x.json': 1, 'y.json': 0.5}, 'b.json': 2})     # ensemble.py:67-75
    assert got == [('a/x.json', 1), ('a/y.json', 0.5), ('b.json', 2)]


def test_synthetic_streams_shape():
    from waymo_2d_tracking_amd import synthetic as syn
    dets = syn.make_sequence_json(0, n_segments=1, n_frames=10, n_objects=100)
    per_frame = len(dets) / (10 * 5)
    assert 80 < per_frame < 120                                                 # ~100 boxes/frame
    assert set(d['category_id'] for d in dets) <= {1, 2, 4}
    assert all(isinstance(v, int) for v in dets[0]['bbox'])
    seg, ts, cam = dets[0]['image_id'].split('/')
    assert cam in syn.CAMERAS and int(ts) > 0


def test_inference_cli_flags():
    from waymo_2d_tracking_amd.detnet.inference import build_parser
    a = build_parser().parse_args(['-i', 'imgs', '--export', 's.json', '--tta', 'x1.5,hflip', '--batch-size=1', '-j', '8'])
    assert a.tta == 'x1.5,hflip' and a.export == 's.json' and a.threshold == 0.01


def _lib_or_skip():
    from waymo_2d_tracking_amd import build
    build.build(verbose=False)
    from waymo_2d_tracking_amd.tracking import utils as T
    return T


@pytest.mark.parametrize('fixture,thr', [('sort_g4_input.json', [0.3, 0.3, 1.0, 0.2]), ('sort_g5_input.json', [0.95, 0.6, 1.0, 0.9]),
                                         ('sort_g4_input.json', [0.0, 0.0, 0.0, 0.0])])
def test_native_json_reader_matches_python_path(golden_dir, fixture, thr):
    """wt_detfile_read == read_data_file + pack_streams (filters, empty frames, stream / frame order, dtypes)."""
    T = _lib_or_skip()
    path = os.path.join(golden_dir, fixture)
    ref = T.pack_streams(T.read_data_file(path, thr))
    nat = T.NativeDetFile(path, thr)
    got = nat.packed()
    assert got['stream_keys'] == ref['stream_keys']
    for k in ('x', 'y', 'w', 'h', 'score', 'category', 'frame_det_offsets', 'stream_frame_offsets', 'frame_ids', 'clip_w', 'clip_h'):
        assert np.array_equal(got[k], ref[k]), k
    nat.close()


def test_native_json_writer_is_byte_identical_to_json_dump(golden_dir, tmp_path):
    """wt_tracks_write_json == json.dump(format_tracks(...)) byte for byte (Python float repr, separators, key order)."""
    T = _lib_or_skip()
    exp = json.load(open(os.path.join(golden_dir, 'sort_g4_expected_a.json')))
    p = exp['params']
    path = os.path.join(golden_dir, 'sort_g4_input.json')
    nat = T.NativeDetFile(path, p['score_threshold'])
    packed = nat.packed()
    rng = np.random.default_rng(0)
    n = 500
    frames = np.sort(rng.integers(0, len(packed['frame_ids']), n))
    out = dict(frame=frames, category=rng.integers(1, 5, n).astype(np.int32),
               bbox=np.concatenate([rng.uniform(0, 1900, (n, 2)), rng.uniform(1, 300, (n, 2))], axis=1),
               score=rng.uniform(0.2, 1.0, n), object_id=rng.integers(1, 10 ** 6, n))
    out['bbox'][::7] = np.round(out['bbox'][::7])                 # integral floats -> "12.0"
    out['bbox'][3] = [0.0, 1e-5, 123456789012345678.0, 1e16]      # exponent forms, zero
    out['score'][5] = 0.1 + 0.2                                    # 0.30000000000000004
    a, b = tmp_path / 'native.json', tmp_path / 'python.json'
    nat.write_tracks(a, out)
    with open(b, 'wt') as fp:
        json.dump(T.format_tracks(packed, out), fp)
    assert open(a, 'rb').read() == open(b, 'rb').read()
    nat.close()


def test_python_float_repr_port():
    import ctypes as C
    _lib_or_skip()
    from waymo_2d_tracking_amd import _lib
    lib = _lib.lib()
    buf = C.create_string_buffer(64)
    rng = np.random.default_rng(1)
    vals = [0.0, -0.0, 1.0, -1.5, 1e-4, 1e-5, 123456.789, 1e15, 1e16, 1e22, 5e-324, 1.7976931348623157e308, 0.1, 1 / 3, 2.5e-7,
            float('nan'), float('inf'), -float('inf')]
    vals += list(rng.uniform(-1e6, 1e6, 200)) + list(10.0 ** rng.uniform(-20, 20, 200))
    for v in vals:
        n = lib.wt_format_double(C.c_double(v), buf, 64)
        assert n > 0
        assert buf.value.decode() == json.dumps(float(v)), (v, buf.value, json.dumps(float(v)))


def _g7_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, 'tta_g7.npz'))
    n = len([k for k in g.files if k.endswith('_spec')])
    return g, n


def test_tta_operators_match_reference_fixture(golden_dir):
    """G7 (row a21): the host-side TTA classes (resize / flips, SequentialTTA order, box un-flipping) against tensors
    and box lists produced by the reference's own detnet/nn/tta.py (oracle/gen_golden_tta.py).  torch CPU ops only."""
    import torch
    from waymo_2d_tracking_amd.detnet.nn import tta as T

    class Fake(object):
        def __init__(self, dets):
            self.dets, self.seen = dets, []

        def predict(self, x):
            self.seen.append(x.clone())
            return [[d.copy() for d in self.dets] for _ in range(x.shape[0])]

    g, n = _g7_cases(golden_dir)
    assert n >= 8
    for ci in range(n):
        spec = str(g['c%d_spec' % ci])
        dets = [g['c%d_det%d' % (ci, k)] for k in range(4)]
        det = Fake(dets)
        y = T.TTA(det, spec.split(',')).predict(torch.from_numpy(g['c%d_x' % ci]))
        np.testing.assert_allclose(det.seen[0].numpy(), g['c%d_pre' % ci], rtol=0, atol=1e-4, err_msg=spec)
        for b in range(2):
            for k in range(4):
                np.testing.assert_array_equal(np.asarray(y[b][k]), g['c%d_post_b%d_k%d' % (ci, b, k)], err_msg=spec)


def test_tta_fused_pre_detection():
    """TTA._fused_pre folds resize-then-flip sequences into (scale, hflip, vflip) and refuses anything else."""
    from waymo_2d_tracking_amd.detnet.nn import tta as T
    mk = lambda spec: T.TTA(object(), spec.split(','))
    assert mk('x1.5,hflip')._fused_pre() == (1.5, True, False)
    assert mk('orig')._fused_pre() == (1.0, False, False)
    assert mk('x2,hflip,vflip')._fused_pre() == (2.0, True, True)
    assert mk('x1.5,x2')._fused_pre() is None
    with pytest.raises(NotImplementedError):
        mk('x1.5,brute')


def test_packaged_gemm_tuning_file():
    """waymo_2d_tracking_amd/tuning: the TunableOp selections shipped with the package parse (validator header + one
    entry per GEMM signature) and enabling them without a GPU is a no-op."""
    from waymo_2d_tracking_amd import tuning
    lines = [l.strip().split(',') for l in open(tuning.PACKAGED) if l.strip()]
    validators = [l for l in lines if l[0] == 'Validator']
    entries = [l for l in lines if l[0] != 'Validator']
    assert {v[1] for v in validators} >= {'PT_VERSION', 'HIPBLASLT_VERSION', 'GCN_ARCH_NAME'}
    assert any('gfx950' in v[2] for v in validators)
    assert len(entries) > 50 and all(len(e) == 4 and float(e[3]) > 0 for e in entries)
    assert len({(e[0], e[1]) for e in entries}) == len(entries)
    import torch
    if not torch.cuda.is_available():
        assert tuning.enable_gemm_tuning() is False


def test_detections_to_wire_tta_scaling():
    """Device-side wire conversion: boxes of a resized (TTA x1.5) pass are normalised by the transformed size and scaled by
    the ORIGINAL size (boxes are normalised in the reference: tta.py ResizeTTA.post_process is the identity), integer
    truncation and 5-decimal scores as in coco.py:249-251."""
    import torch
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import detections_to_wire
    boxes = torch.tensor([[150.0, 300.0, 450.9, 600.3], [0.0, 0.0, 2880.0, 1920.0]])
    scores = torch.tensor([0.123456, 0.999994])
    classes = torch.tensor([0, 3])
    xywh, score, cat = detections_to_wire(boxes, scores, classes, 2880, 1920, 1920, 1280)
    assert xywh.tolist() == [[100.0, 200.0, 200.0, 200.0], [0.0, 0.0, 1920.0, 1280.0]]
    assert [round(v, 5) for v in score.tolist()] == [0.12346, 0.99999] and cat.tolist() == [1, 4]
    same, _, _ = detections_to_wire(boxes, scores, classes, 2880, 1920)
    assert same.tolist() == [[150.0, 300.0, 300.0, 300.0], [0.0, 0.0, 2880.0, 1920.0]]


def test_image_loader_reports_original_size_and_contrast_precedes_resize(tmp_path):
    """detnet/inference.py:170-178 (ToRGB, AutoContrast, Resize) and export.py:159-165 -> coco.py:243-246: exported boxes are
    scaled by the ORIGINAL image size, and with --auto-contrast --resize the contrast stretch sees the unresized pixels."""
    from PIL import Image, ImageOps
    from waymo_2d_tracking_amd.detnet.inference import ImageLoader, resize_size
    rng = np.random.default_rng(3)
    arr = rng.integers(40, 200, (60, 90, 3), dtype=np.uint8)
    path = str(tmp_path / 'a.png')
    Image.fromarray(arr).save(path)
    plain, size = ImageLoader([], resize=None)._decode(path)
    assert size == (90, 60) and np.array_equal(plain, arr)
    ld = ImageLoader([], resize=30, auto_contrast=True)
    assert ld.auto_contrast_in_loader
    got, size = ld._decode(path)
    assert size == (90, 60)                                       # not the resized (45, 30)
    oh, ow = resize_size(90, 60, 30)
    exp = np.asarray(ImageOps.autocontrast(Image.fromarray(arr)).resize((ow, oh), Image.BILINEAR))
    assert got.shape == (30, 45, 3) and np.array_equal(got, exp)
    assert not ImageLoader([], resize=None, auto_contrast=True).auto_contrast_in_loader      # GPU autocontrast_ then


def test_round_score_halfway_deviation_is_explicit():
    """ensemble.py:62 `round(bbox[0], 5)` on a numpy float64.  Under the reference's python 3.7 / numpy 1.18 the scalar falls
    back to float.__round__ (correctly rounded decimal); numpy >= 1.19 (the interpreters available here, and the one the fixtures
    were generated with) rounds by scale - rint - unscale.  The two differ by one unit of the 5th decimal on many x.xxxxx5
    inputs.  The host code follows Python's float rounding - this test constructs such scores and pins that choice, so the
    deviation from the fixtures' numpy is documented rather than hidden behind a tolerance."""
    from waymo_2d_tracking_amd.detnet import ensemble as E
    scores = np.array([5e-06 + 0.5, 2.5e-05 + 0.25, 0.300005, 0.123455, 0.99999499999, 0.5], np.float64)
    out5 = np.zeros((len(scores), 5))
    out5[:, 0] = scores
    out5[:, 1:] = [10.9, 20.1, 30.5, 40.99]
    packed = dict(group_offsets=np.array([0, len(scores)], np.int64), n_groups=1, ncat=1, image_lo=0)
    rows = E.output_rows(packed, [1], out5, np.array([len(scores)], np.int64), 0.0)
    assert rows['score'].tolist() == [round(float(v), 5) for v in scores]              # Python's correctly rounded result
    numpy_way = (np.rint(scores * 1e5) / 1e5)
    diff = np.abs(rows['score'] - numpy_way)
    assert (diff > 0).any() and diff.max() < 1.0000001e-5                               # differs, by exactly one unit
    assert rows['bbox'].tolist() == [[10, 20, 30, 40]] * len(scores)                    # astype(int): truncation


def test_autocontrast_lut_equals_pil_for_every_range():
    """ImageOps.autocontrast builds lut[v] = int(v * (255.0 / (hi - lo)) + (-lo * scale)) in Python floats; the tensor version must
    round the same way for EVERY (lo, hi) - a reciprocal-multiply instead of the division moves 131 -> 150 instead of 149 at
    (lo, hi) = (1, 222) (found through a JPEG-decoded test image).  All 32 640 ranges, on CPU tensors (the same torch code runs
    on the GPU in the loader)."""
    import torch
    from waymo_2d_tracking_amd.detnet.inference import autocontrast_
    v = np.arange(256)
    bad = []
    for lo in range(0, 255):
        his = np.arange(lo + 1, 256)
        # one 3-channel image per 3 values of hi: channel c holds the values {lo, hi_c} plus the full ramp clipped to that range
        for k in range(0, len(his), 3):
            hs = [int(his[min(k + c, len(his) - 1)]) for c in range(3)]
            img = np.stack([np.clip(v, lo, h) for h in hs], -1).astype(np.uint8).reshape(1, 256, 3)
            got = autocontrast_(torch.from_numpy(img)).numpy()[0]
            for c, h in enumerate(hs):
                scale = 255.0 / (h - lo)
                offset = -lo * scale
                exp = np.array([min(255, max(0, int(x * scale + offset))) for x in np.clip(v, lo, h)])
                if not np.array_equal(got[:, c], exp):
                    bad.append((lo, h))
    assert not bad, bad[:10]
