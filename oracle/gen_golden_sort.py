"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/sort_*.{json,npz} by running the REFERENCE's own
SORT control flow (/root/reference/tracking/{utils.py,sort/sort.py,sort/tracker_sort.py}) in this container.

Run with the legacy-promotion interpreter (numpy 1.26 behaves like the reference's numpy 1.18;
numpy >= 2 changes float32/python-float promotion and therefore the results):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/gen_golden_sort.py

The third-party arithmetic that is absent here (filterpy KalmanFilter, sklearn 0.22.2
linear_assignment, numba.jit) is injected from oracle/thirdparty_restated.py.  Only input/output
DATA is written to tests/golden/; no reference source or bytecode is copied.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF_TRACKING = '/root/reference/tracking'
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install_stubs():
    restated = _load('thirdparty_restated', os.path.join(HERE, 'thirdparty_restated.py'))

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub('numba', jit=lambda f=None, **kw: f if callable(f) else (lambda g: g))
    stub('filterpy')
    stub('filterpy.kalman', KalmanFilter=restated.KalmanFilter)
    import sklearn.utils  # noqa: F401  (real package; only the removed sub-module is injected)
    stub('sklearn.utils.linear_assignment_', linear_assignment=restated.linear_assignment)
    sys.path.insert(0, REF_TRACKING)
    return restated


def main():
    assert np.__version__.startswith('1.'), 'run under numpy 1.x (legacy scalar promotion)'
    install_stubs()
    import utils as ref_utils            # /root/reference/tracking/utils.py
    import sort.sort as ref_sort         # /root/reference/tracking/sort/sort.py
    syn = _load('synthetic', os.path.join(REPO, 'waymo_2d_tracking_amd', 'synthetic.py'))
    os.makedirs(GOLDEN, exist_ok=True)

    # ---------------- G4: synthetic streams -> tracking JSON (+ traces) ----------------
    rng = np.random.default_rng(20200601)
    streams = []
    seg_a = 'segment-10203656353524179475_7625_000_7645_000_with_camera_labels'
    seg_b = 'segment-1024360143612057520_3580_000_3600_000_with_camera_labels'
    for seg, cam, n_obj, integer in ((seg_a, 'FRONT', 14, True), (seg_a, 'SIDE_LEFT', 9, True),
                                     (seg_b, 'FRONT_RIGHT', 12, False)):
        d = syn.stream_detections(rng, 60, n_obj, cam, lifetimes=True, clutter=0.15, score_lo=0.05,
                                  integer_boxes=integer)
        streams.append((seg, cam, d))
    dets = syn.detections_json(streams, integer_boxes=True)
    # stream 3 keeps fractional boxes (exercises the float32 roundings of sort.py:56-62,38-46)
    frac = syn.detections_json([streams[2]], integer_boxes=False)
    dets = [e for e in dets if not e['image_id'].startswith(seg_b)] + frac
    # an all-filtered frame (frame key exists, tracker still ticks: utils.py:76-87) ...
    t20 = syn.frame_timestamp(20)
    for e in dets:
        if e['image_id'] == '%s/%i/FRONT' % (seg_a, t20):
            e['score'] = 0.001
    # ... a frame missing from the input altogether (tracker does not tick)
    t33 = syn.frame_timestamp(33)
    dets = [e for e in dets if e['image_id'] != '%s/%i/SIDE_LEFT' % (seg_a, t33)]
    # degenerate boxes dropped by read_data_file (utils.py:79) and entries without a score (gt style)
    dets.insert(7, {'image_id': '%s/%i/FRONT' % (seg_a, syn.frame_timestamp(0)), 'category_id': 1,
                    'bbox': [100, 100, 0, 50], 'score': 0.99})
    dets.insert(11, {'image_id': '%s/%i/FRONT' % (seg_a, syn.frame_timestamp(1)), 'category_id': 2,
                     'bbox': [400.5, 300.25, 60.5, 120.75]})
    # disjoint class-4 boxes far apart (zero-IoU assignment with threshold 0.0, SURVEY "hard parts")
    for f in range(5, 15):
        for k in range(3 if f % 3 else 2):
            dets.append({'image_id': '%s/%i/FRONT' % (seg_a, syn.frame_timestamp(f)), 'category_id': 4,
                         'bbox': [50 + 600 * k + 70 * (f % 2), 1100, 40, 60], 'score': 0.97})
    with open(os.path.join(GOLDEN, 'sort_g4_input.json'), 'wt') as fp:
        json.dump(dets, fp)

    variants = {
        'a': dict(score_threshold=[0.3, 0.3, 1.0, 0.2], iou_threshold=[0.01, 0.01, 1.0, 0.0], max_age=2, min_hits=0),
        'b': dict(score_threshold=[0.5, 0.2, 1.0, 0.5], iou_threshold=[0.3, 0.3, 0.3, 0.3], max_age=1, min_hits=3),
        'c': dict(score_threshold=[0.0, 0.0, 0.0, 0.0], iou_threshold=[0.1, 0.05, 1.0, 0.0], max_age=4, min_hits=1),
    }
    trace = {}
    for name, v in variants.items():
        ref_sort.KalmanBoxTracker.count = 0          # fresh process semantics (sort.py:86)
        log = {'assoc': [], 'state': []}
        orig_assoc = ref_sort.associate_detections_to_trackers
        orig_update = ref_sort.Sort.update

        def assoc(detections, trackers, iou_threshold=0.3, _o=orig_assoc, _log=log):
            r = _o(detections, trackers, iou_threshold=iou_threshold)
            _log['assoc'].append((np.asarray(r[0]).reshape(-1, 2).copy(), np.asarray(r[1]).astype(int).copy(),
                                  np.asarray(r[2]).astype(int).reshape(-1)[:len(trackers)].copy()
                                  if len(trackers) else np.zeros(0, int)))
            return r

        def update(self, dets_, iou_threshold, _o=orig_update, _log=log):
            r = _o(self, dets_, iou_threshold)
            xs = np.array([t.kf.x[:, 0] for t in self.trackers]).reshape(-1, 7)
            ps = np.array([t.kf.P for t in self.trackers]).reshape(-1, 7, 7)
            ids = np.array([t.id for t in self.trackers], dtype=int)
            _log['state'].append((ids, xs, ps))
            return r

        ref_sort.associate_detections_to_trackers = assoc
        ref_sort.Sort.update = update
        try:
            predictions = ref_utils.read_data_file(os.path.join(GOLDEN, 'sort_g4_input.json'), v['score_threshold'])
            out = []
            for segment_id in predictions.keys():                       # track.py:43-47
                for camera_id in predictions[segment_id]:
                    out += ref_utils.track_sort(predictions, segment_id, camera_id, v['iou_threshold'],
                                                v['max_age'], v['min_hits'])
        finally:
            ref_sort.associate_detections_to_trackers = orig_assoc
            ref_sort.Sort.update = orig_update
        out = [dict(e, bbox=[float(b) for b in e['bbox']], score=float(e['score'])) for e in out]
        with open(os.path.join(GOLDEN, 'sort_g4_expected_%s.json' % name), 'wt') as fp:
            json.dump({'params': v, 'tracks': out, 'n_ids': int(ref_sort.KalmanBoxTracker.count)}, fp)
        # flatten traces (ragged -> concatenated + offsets)
        def ragged(arrs, width):
            arrs = [np.asarray(a).reshape(-1, width) if width else np.asarray(a).reshape(-1) for a in arrs]
            off = np.cumsum([0] + [len(a) for a in arrs])
            return (np.concatenate(arrs) if arrs else np.zeros((0, width))), off
        m, mo = ragged([a[0] for a in log['assoc']], 2)
        ud, udo = ragged([a[1] for a in log['assoc']], 0)
        ut, uto = ragged([a[2] for a in log['assoc']], 0)
        ids, io = ragged([s[0] for s in log['state']], 0)
        xs, _ = ragged([s[1] for s in log['state']], 7)
        ps, _ = ragged([s[2].reshape(-1, 49) for s in log['state']], 49)
        trace.update({name + '_matched': m.astype(np.int32), name + '_matched_off': mo,
                      name + '_unmatched_dets': ud.astype(np.int32), name + '_unmatched_dets_off': udo,
                      name + '_unmatched_trks': ut.astype(np.int32), name + '_unmatched_trks_off': uto,
                      name + '_ids': ids.astype(np.int32), name + '_state_off': io,
                      name + '_x': xs, name + '_P': ps})
        print('variant', name, ': tracks', len(out), 'ids', ref_sort.KalmanBoxTracker.count,
              'assoc calls', len(log['assoc']), 'updates', len(log['state']))
    np.savez_compressed(os.path.join(GOLDEN, 'sort_g4_trace.npz'), **trace)

    # ---------------- G5: read_data_file ordering / filters on a toy file ----------------
    toy = {'annotations': [
        {'image_id': 'segB/200/FRONT', 'category_id': 2, 'bbox': [1, 2, 30, 40], 'score': 0.7},
        {'image_id': 'segA/100/SIDE_LEFT', 'category_id': 1, 'bbox': [5, 6, 0.5, 40], 'score': 0.99},
        {'image_id': 'segA/100/SIDE_LEFT', 'category_id': 1, 'bbox': [5, 6, 50, 40], 'score': 0.2},
        {'image_id': 'segA/90/FRONT', 'category_id': 4, 'bbox': [7, 8, 20, 21], 'object_id': 'gt-17'},
        {'image_id': 'segB/150/FRONT', 'category_id': 1, 'bbox': [9, 9, 19, 19], 'score': 0.96},
        {'image_id': 'segA/100/SIDE_LEFT', 'category_id': 2, 'bbox': [1.5, 2.5, 10.25, 11.75], 'score': 0.6},
    ]}
    with open(os.path.join(GOLDEN, 'sort_g5_input.json'), 'wt') as fp:
        json.dump(toy, fp)
    entries = ref_utils.read_data_file(os.path.join(GOLDEN, 'sort_g5_input.json'), [0.95, 0.6, 1.0, 0.9])
    flat = [[seg, cam, int(fr), entries[seg][cam][fr]] for seg in entries for cam in entries[seg]
            for fr in entries[seg][cam]]      # insertion order preserved
    with open(os.path.join(GOLDEN, 'sort_g5_expected.json'), 'wt') as fp:
        json.dump(flat, fp)

    # ---------------- single-tracker API vectors: Sort.update call by call ----------------
    ref_sort.KalmanBoxTracker.count = 0
    rng = np.random.default_rng(7)
    d = syn.stream_detections(rng, 25, 8, 'FRONT', lifetimes=True, integer_boxes=False)
    s = ref_sort.Sort(max_age=2, min_hits=1)
    calls_in, calls_out = [], []
    for f in range(25):
        sel = d['frame'] == f
        arr = np.stack([d['x'][sel], d['y'][sel], d['x'][sel] + d['w'][sel], d['y'][sel] + d['h'][sel],
                        d['score'][sel]], axis=1).astype(np.float32)
        if f in (9, 10):
            arr = np.array([], dtype=np.float32)
        r = s.update(arr, 0.2)
        calls_in.append(arr.reshape(-1, 5)); calls_out.append(np.asarray(r, dtype=np.float64).reshape(-1, 6))
    np.savez_compressed(os.path.join(GOLDEN, 'sort_update_calls.npz'),
                        in_off=np.cumsum([0] + [len(a) for a in calls_in]), dets=np.concatenate(calls_in),
                        out_off=np.cumsum([0] + [len(a) for a in calls_out]), rows=np.concatenate(calls_out))
    print('done')


if __name__ == '__main__':
    main()
