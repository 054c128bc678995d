"""TEST INFRASTRUCTURE ONLY.  tests/golden/sort_g9_nonfinite.npz: the REFERENCE's Sort.update
(/root/reference/tracking/sort/sort.py:244-296) run call by call on detections that drive a track's predicted box non-finite.

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/gen_golden_sort_nonfinite.py

Why this fixture exists: the HIP path and the C oracle drop a track whose predicted box has ANY non-finite
coordinate, while sort.py:258-265 pops the tracker only for NaN (`np.isnan`) but removes the row for NaN and +-inf
(`np.ma.masked_invalid`) - for a predicted box with an inf and no NaN the reference's `trks` rows and `self.trackers` would go
out of step.  That state is not reachable:
  * the only overflow a finite float32 detection can cause is the float32 area `s = w*h` (sort.py:60) -> inf, and then
    convert_x_to_bbox gives w = sqrt(inf * r) = inf, h = s / w = inf / inf = NaN; inf / NaN detection coordinates give NaN in
    cx = x1 + w/2 or in w = x2 - x1 the same way;
  * an inf anywhere in the Kalman state x turns into NaN in the other rows of the DENSE product F.x of filterpy's predict()
    (0 * inf), so even a state with an injected inf is popped by the reference - recorded below by injecting x[0] = inf into a
    live tracker (`inf_state_reference_behaviour`: the tracker is gone after the call, the other track is unaffected).
So the two rules ("pop on NaN" vs "drop on any non-finite") are observably the same; this fixture pins the reachable cases.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden_sort import GOLDEN, install_stubs          # noqa: E402


def main():
    assert np.__version__.startswith('1.'), 'run under numpy 1.x (legacy scalar promotion)'
    install_stubs()
    import sort.sort as ref_sort
    np.seterr(all='ignore')
    ref_sort.KalmanBoxTracker.count = 0
    s = ref_sort.Sort(max_age=2, min_hits=0)
    big = 3.0e19                                   # w = h = 3e19 -> float32 area 9e38 overflows to inf
    frames = [
        [[10, 10, 50, 60, 0.9], [0, 0, big, big, 0.8]],
        [[12, 11, 52, 61, 0.9], [200, 200, 260, 280, 0.7]],
        [[14, 12, 54, 62, 0.9], [203, 201, 263, 281, 0.7], [0, 0, big, 2 * big, 0.6]],
        [[16, 13, 56, 63, 0.9], [206, 202, 266, 282, 0.7]],
        [],
        [[20, 15, 60, 65, 0.9], [600, 600, 600 + big, 640, 0.5]],
        [[22, 16, 62, 66, 0.9]],
    ]
    calls_in, calls_out, live = [], [], []
    for f in frames:
        arr = np.asarray(f, dtype=np.float32).reshape(-1, 5) if f else np.array([], dtype=np.float32)
        r = s.update(arr, 0.1)
        calls_in.append(arr.reshape(-1, 5))
        calls_out.append(np.asarray(r, dtype=np.float64).reshape(-1, 6))
        live.append(np.array([t.id for t in s.trackers], dtype=np.int64))
    # state injection (NOT reachable through the API): a predicted box with inf and no NaN
    ref_sort.KalmanBoxTracker.count = 0
    s2 = ref_sort.Sort(max_age=2, min_hits=0)
    s2.update(np.asarray([[10, 10, 50, 60, 0.9], [200, 200, 260, 280, 0.7]], dtype=np.float32), 0.1)
    s2.trackers[0].kf.x[0, 0] = np.inf
    try:
        s2.update(np.asarray([[12, 11, 52, 61, 0.9], [203, 201, 263, 281, 0.7]], dtype=np.float32), 0.1)
        behaviour = 'tracker 0 popped (dense F.x: 0 * inf = NaN); live ids afterwards %s' % [t.id for t in s2.trackers]
        assert 0 not in [t.id for t in s2.trackers]
    except Exception as e:                         # noqa: BLE001
        behaviour = '%s: %s' % (type(e).__name__, e)
    print('inf-state behaviour of the reference:', behaviour)
    np.savez_compressed(os.path.join(GOLDEN, 'sort_g9_nonfinite.npz'),
                        in_off=np.cumsum([0] + [len(a) for a in calls_in]), dets=np.concatenate(calls_in),
                        out_off=np.cumsum([0] + [len(a) for a in calls_out]), rows=np.concatenate(calls_out),
                        live_off=np.cumsum([0] + [len(a) for a in live]), live_ids=np.concatenate(live),
                        inf_state_reference_behaviour=np.array(behaviour))
    for i, (o, l) in enumerate(zip(calls_out, live)):
        print(i, 'rows', o[:, 4].astype(int).tolist(), 'live', l.tolist())


if __name__ == '__main__':
    main()
