"""Wire-format I/O at Waymo scale (host code, runs without a GPU): native reader / writers of libwaymotrack.so
against Python's json module and (for the protobuf) against building the messages one by one the way the reference does."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from waymo_2d_tracking_amd import synthetic as syn, waymo_proto as W
from waymo_2d_tracking_amd.detnet import export as E
from waymo_2d_tracking_amd.detnet import ensemble as ENS

n_segments = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dets = syn.make_sequence_json(7, n_segments=n_segments, n_frames=198, n_objects=100)
tmp = tempfile.mkdtemp()
src = os.path.join(tmp, 'det.json')
t0 = time.perf_counter(); json.dump(dets, open(src, 'w')); t_dump = time.perf_counter() - t0
t0 = time.perf_counter(); rows_py = json.load(open(src)); t_load = time.perf_counter() - t0
n = len(rows_py)
print('%d detection rows (%d segments x 5 cameras x 198 frames), file %.1f MB' % (n, n_segments, os.path.getsize(src) / 1e6))
print('python  json.load %.2f s   json.dump %.2f s' % (t_load, t_dump))
t0 = time.perf_counter(); sub = ENS.read_submission(src); t_read = time.perf_counter() - t0
print('native  wt_detjson_read -> columns %.2f s (%.1fx)' % (t_read, t_load / t_read))
out = os.path.join(tmp, 'out.json')
image_ids = sub['image_ids']
rows = dict(image=np.asarray(sub['image'], np.int32), category=np.asarray(sub['category'], np.int32),
            bbox=np.stack([sub['x'], sub['y'], sub['w'], sub['h']], 1).astype(np.int64), score=np.asarray(sub['score'], np.float64))
t0 = time.perf_counter(); E.write_detections_json(out, image_ids, rows); t_write = time.perf_counter() - t0
print('native  wt_detections_write_json %.2f s (%.1fx), bytes identical to json.dump: %s' % (
    t_write, t_dump / t_write, open(out, 'rb').read() == open(src, 'rb').read()))
pb = os.path.join(tmp, 'sub.bin')
t0 = time.perf_counter(); c = W.entries_to_columns(rows_py); t_cols = time.perf_counter() - t0
t0 = time.perf_counter(); nbytes = W.write(pb, c, submission=dict(task=W.DETECTION_2D, account_name='a@b.c', unique_method_name='m',
                                                                   authors=['x'], affiliation='y', description='z', sensor_type=W.CAMERA_ALL))
t_pb = time.perf_counter() - t0
print('protobuf Submission: rows -> columns %.2f s (python), native encode + write %.3f s, %.1f MB' % (t_cols, t_pb, nbytes / 1e6))
