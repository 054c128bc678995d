"""Drop-in for /root/reference/tracking/track.py (same flags, same prints, same output JSON):

    python -m waymo_2d_tracking_amd.tracking.track --input det.json --output tracks.json \
        --max-age=2 --min-hits=0 --score-threshold=0.95,0.6,1.0,0.9

All (segment, camera) streams are tracked in one GPU call (utils.track_all) instead of the sequential loop of
track.py:43-47; `--ground-truth` is accepted but, like in the reference, its content is never used (the file
is not required to exist).
"""
import argparse
import json
import time

import os

from .utils import read_data_file, track_all


def _floats(s):
    return [float(item) for item in s.split(',')]


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--ground-truth", type=str, default='/data/waymo/det2d/validation/images.json',
                        help='ground-truth json')
    parser.add_argument("--input", type=str, default='submission_12545.json', help='submission.json')
    parser.add_argument("--output", type=str, default='tracker_predictions.json',
                        help='file to save the tracker predictions')
    parser.add_argument("--max-age", type=int, default=1, help='SORT max-age')
    parser.add_argument("--min-hits", type=int, default=0, help='SORT min-hits')
    parser.add_argument("--score-threshold", type=_floats, default=[0.95, 0.6, 1.0, 0.9],
                        help='score threshold to track')
    parser.add_argument("--iou-threshold", type=_floats, default=[0.01, 0.01, 1.0, 0.0],
                        help='IOU threshold for tracking')
    parser.add_argument("--segment-id", type=str, help='track only a single segment')
    parser.add_argument("--python-io", action='store_true',
                        help='parse / write JSON with the Python json module (default: native reader / writer of libwaymotrack)')
    return parser


def main_native(args):
    """file -> SoA in HBM -> file without per-row Python objects (wt_detfile_read / wt_tracks_write_json)."""
    from . import utils as T
    nat = T.NativeDetFile(args.input, args.score_threshold)
    packed = nat.packed()
    start_time = time.time()
    for segment_id in dict.fromkeys(s for s, _ in packed['stream_keys']):
        print(segment_id)
    out, births = T.track_packed(packed, args.iou_threshold, args.max_age, args.min_hits, None, T._GLOBAL_IDS['next'])
    T._GLOBAL_IDS['next'] += births
    print("duration: %.2fs" % (time.time() - start_time))
    nat.write_tracks(args.output, out)
    nat.close()
    return 0


def main_sharded(args, world, rank):
    """torchrun: one process per GPU.  Every rank parses the file natively, tracks its block of streams and the result
    columns reach rank 0 through one tensor gather (distributed.track_packed_sharded); rank 0 writes the JSON natively."""
    import torch
    import torch.distributed as dist
    from . import utils as T
    from ..distributed import track_packed_sharded
    backend = os.environ.get('WT_DIST_BACKEND', 'nccl')
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    dist.init_process_group(backend)
    nat = T.NativeDetFile(args.input, args.score_threshold)
    packed = nat.packed()
    if args.segment_id:
        keep = [i for i, (s, _) in enumerate(packed['stream_keys']) if s == args.segment_id]
        packed = T.slice_streams(packed, keep[0], keep[-1] + 1) if keep else T.slice_streams(packed, 0, 0)
    start_time = time.time()
    if rank == 0:
        for segment_id in dict.fromkeys(s for s, _ in packed['stream_keys']):
            print(segment_id)
    cols, births = track_packed_sharded(packed, args.iou_threshold, args.max_age, args.min_hits, None, None, T._GLOBAL_IDS['next'])
    T._GLOBAL_IDS['next'] += births
    if rank == 0:
        print("duration: %.2fs" % (time.time() - start_time))
        if args.segment_id:
            with open(args.output, 'wt') as fp:
                json.dump(T.format_tracks(packed, cols), fp)
        else:
            nat.write_tracks(args.output, cols)
    nat.close()
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main(argv=None):
    args = build_parser().parse_args(argv)
    print(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        return main_sharded(args, world, rank)
    if not args.python_io and not args.segment_id:
        return main_native(args)
    predictions = read_data_file(args.input, args.score_threshold)
    if args.segment_id:
        predictions = {k: v for k, v in predictions.items() if k in [args.segment_id]}
    start_time = time.time()
    for segment_id in predictions.keys():
        print(segment_id)
    tracked_predictions = track_all(predictions, args.iou_threshold, args.max_age, args.min_hits)
    print("duration: %.2fs" % (time.time() - start_time))
    with open(args.output, 'wt') as fp:
        json.dump(tracked_predictions, fp)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
