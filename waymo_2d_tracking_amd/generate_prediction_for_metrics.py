"""Drop-in for /root/reference/generate_prediction_for_metrics.py (same flags, same output bytes): submission.json or
annotations.json -> `metrics.Objects` protobuf for the official Waymo metrics tools."""
import argparse
import json

from . import waymo_proto as W

OBJECT_TYPES = {1: W.TYPE_VEHICLE, 2: W.TYPE_PEDESTRIAN, 3: W.TYPE_SIGN, 4: W.TYPE_CYCLIST}     # :34-39
DIFFICULTY_LEVELS = {1: 1, 2: 2}                                                                  # :40-43


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--type", choices=['prediction', 'ground-truth'], default='prediction',
                        help='to generate ground truth or prediction')
    parser.add_argument("--input", type=str, required=True, help='either submission.json or annotations.json')
    parser.add_argument("--output", type=str, default='out.bin', help='output file')
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    print(args)
    entries = json.load(open(args.input))
    if args.type == 'ground-truth':
        entries = entries['annotations']
    cols = W.entries_to_columns(entries, OBJECT_TYPES, DIFFICULTY_LEVELS)
    W.write(args.output, cols, metrics_mode=True)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
