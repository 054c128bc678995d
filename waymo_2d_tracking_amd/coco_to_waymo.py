"""Drop-in for /root/reference/coco_to_waymo.py (same flags, same output bytes): detection / tracking JSON -> Waymo
`Submission` protobuf.  The messages are serialised natively from columns (waymo_proto.write)."""
import argparse
import json
from pathlib import Path

from . import waymo_proto as W


def create_pb_submission(path, detections, unique_method_name, description, account_name, tracking):
    """coco_to_waymo.py:63-82 - writes the serialised Submission to `path`, returns its size in bytes."""
    if description is None:
        raise TypeError('description must be a string')              # protobuf refuses None for a string field
    cols = W.entries_to_columns(detections)                         # object.type = category_id (coco_to_waymo.py:48)
    return W.write(path, cols, metrics_mode=False, submission=dict(
        task=W.TRACKING_2D if tracking else W.DETECTION_2D, account_name=account_name,
        authors=['Yuan Xu', 'Erdene-Ochir Tuguldur'], affiliation='DAInamite', unique_method_name=unique_method_name,
        description=description, sensor_type=W.CAMERA_ALL))


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('detection', type=str, nargs='+', help='detection result json file')
    parser.add_argument('--unique-method-name', type=str, required=True, help='unique method name. Max 25 chars.')
    parser.add_argument('--description', type=str, help='detailed description of method.')
    parser.add_argument('--account-name', type=str, required=True, help='email')
    parser.add_argument('--tracking', action='store_true', help='tracking submission')
    parser.add_argument('-o', '--output', type=str, help='output submission file')
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    detections = []
    for f in args.detection:
        detections += json.load(open(f))
    output = Path(args.output)
    output.parent.mkdir(parents=True, exist_ok=True)
    create_pb_submission(args.output, detections, args.unique_method_name, args.description, args.account_name, args.tracking)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
