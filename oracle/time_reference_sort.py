"""TEST / MEASUREMENT INFRASTRUCTURE ONLY ("reference SORT on stubs").

Times the REFERENCE's own tracking loop (/root/reference/tracking/utils.py:track_sort, timed like tracking/track.py:42-49) on
the config-1 workload bench.py --stage track uses (one synthetic segment: 5 cameras x 198 frames x ~100 boxes per frame, all boxes
tracked, max_age 2, min_hits 0) in THIS container; the reference never travels to the GPU box.  filterpy / sklearn 0.22.2 /
numba are absent: the restatements of oracle/thirdparty_restated.py are injected exactly as for the golden fixtures.

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=1 /opt/conda/bin/python3.9 oracle/time_reference_sort.py

Prints one JSON line; the number is quoted in DESIGN.md section 5 (label: reference-on-stub, container CPU)."""
import importlib.util
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from gen_golden_sort import install_stubs, _load  # noqa: E402


def main():
    assert np.__version__.startswith('1.'), 'run under numpy 1.x (legacy scalar promotion)'
    install_stubs()
    import utils as ref_utils            # /root/reference/tracking/utils.py
    syn = _load('synthetic', os.path.join(REPO, 'waymo_2d_tracking_amd', 'synthetic.py'))
    rng = np.random.default_rng(1000)    # bench.py: build_predictions(1000, 1)
    seg = 'segment-10203656353524179475_7625_000_7645_000_with_camera_labels'
    predictions = {seg: {}}
    n_dets = 0
    for cam in syn.CAMERAS:
        d = syn.stream_detections(rng, 198, 100, cam)
        frames = {}
        for i in range(len(d['frame'])):
            frames.setdefault(syn.frame_timestamp(int(d['frame'][i])), []).append(
                {'bbox': [float(d['x'][i]), float(d['y'][i]), float(d['w'][i]), float(d['h'][i])], 'score': float(d['score'][i]),
                 'category_id': int(d['cat'][i])})
        for f in range(198):
            frames.setdefault(syn.frame_timestamp(f), [])
        predictions[seg][cam] = frames
        n_dets += len(d['frame'])
    iou_thresholds = [0.01, 0.01, 1.0, 0.0]
    start_time = time.time()
    rows = 0
    for segment_id in predictions.keys():
        for camera_id in predictions[segment_id].keys():
            rows += len(ref_utils.track_sort(predictions, segment_id, camera_id, iou_thresholds, 2, 0))
    dt = time.time() - start_time
    print(json.dumps(dict(what='reference tracking/utils.py:track_sort on stubs (restated filterpy / sklearn 0.22.2 Munkres), 1 thread',
                          frames=990, detections=n_dets, rows=rows, seconds=round(dt, 2), frames_per_s=round(990 / dt, 1),
                          ms_per_frame=round(1e3 * dt / 990, 2), cpu='container: 8-vCPU Xeon @ 2.10 GHz', numpy=np.__version__)))


if __name__ == '__main__':
    main()
