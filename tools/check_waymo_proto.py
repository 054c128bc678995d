#!/usr/bin/env python
"""One-command check of the RECALLED Waymo protobuf schema for whoever has the `waymo_open_dataset` package (it is not in the build image and
there is no network, so the field numbers in csrc/waymo_proto.hip / tests/test_waymo_proto.py / INTEGRATION.md are "parity unpinned").

    python tools/check_waymo_proto.py            # exit 0 iff every recalled (message, field, number, type) matches the installed package
                                                 # and a file written by wt_waymo_objects_write parses back to the same values

Compares the table below against label_pb2 / metrics_pb2 / submission_pb2 / dataset_pb2 descriptors, then round-trips a small Submission."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (proto file, message, field, number, type as the descriptor names it)
RECALLED = [
    ('label.proto', 'Label', 'box', 1, 'message'), ('label.proto', 'Label', 'type', 3, 'enum'), ('label.proto', 'Label', 'id', 4, 'string'),
    ('label.proto', 'Label', 'detection_difficulty_level', 5, 'enum'), ('label.proto', 'Label', 'tracking_difficulty_level', 6, 'enum'),
    ('label.proto', 'Label', 'num_lidar_points_in_box', 7, 'int32'),
    ('label.proto', 'Label.Box', 'center_x', 1, 'double'), ('label.proto', 'Label.Box', 'center_y', 2, 'double'),
    ('label.proto', 'Label.Box', 'center_z', 3, 'double'), ('label.proto', 'Label.Box', 'width', 4, 'double'),
    ('label.proto', 'Label.Box', 'length', 5, 'double'), ('label.proto', 'Label.Box', 'height', 6, 'double'),
    ('label.proto', 'Label.Box', 'heading', 7, 'double'),
    ('metrics.proto', 'Object', 'object', 1, 'message'), ('metrics.proto', 'Object', 'score', 2, 'float'),
    ('metrics.proto', 'Object', 'overlap_with_nlz', 3, 'bool'), ('metrics.proto', 'Object', 'context_name', 4, 'string'),
    ('metrics.proto', 'Object', 'frame_timestamp_micros', 5, 'int64'), ('metrics.proto', 'Object', 'camera_name', 6, 'enum'),
    ('metrics.proto', 'Objects', 'objects', 1, 'message'),
    ('submission.proto', 'Submission', 'task', 1, 'enum'), ('submission.proto', 'Submission', 'account_name', 2, 'string'),
    ('submission.proto', 'Submission', 'unique_method_name', 3, 'string'), ('submission.proto', 'Submission', 'authors', 4, 'string'),
    ('submission.proto', 'Submission', 'affiliation', 5, 'string'), ('submission.proto', 'Submission', 'description', 6, 'string'),
    ('submission.proto', 'Submission', 'method_link', 7, 'string'), ('submission.proto', 'Submission', 'sensor_type', 8, 'enum'),
    ('submission.proto', 'Submission', 'number_past_frames_exclude_current', 9, 'int32'),
    ('submission.proto', 'Submission', 'number_future_frames_exclude_current', 10, 'int32'),
    ('submission.proto', 'Submission', 'inference_results', 11, 'message'),
]
RECALLED_ENUMS = [
    ('label.proto', 'Label.Type', {'TYPE_UNKNOWN': 0, 'TYPE_VEHICLE': 1, 'TYPE_PEDESTRIAN': 2, 'TYPE_SIGN': 3, 'TYPE_CYCLIST': 4}),
    ('label.proto', 'Label.DifficultyLevel', {'UNKNOWN': 0, 'LEVEL_1': 1, 'LEVEL_2': 2}),
    ('dataset.proto', 'CameraName.Name', {'UNKNOWN': 0, 'FRONT': 1, 'FRONT_LEFT': 2, 'FRONT_RIGHT': 3, 'SIDE_LEFT': 4, 'SIDE_RIGHT': 5}),
    ('submission.proto', 'Submission.Task', {'UNKNOWN': 0, 'DETECTION_2D': 1, 'DETECTION_3D': 2, 'TRACKING_2D': 3, 'TRACKING_3D': 4}),
    ('submission.proto', 'Submission.SensorType', {'INVALID': 0, 'LIDAR_ALL': 1, 'LIDAR_TOP': 2, 'CAMERA_ALL': 3, 'CAMERA_LIDAR_TOP': 4,
                                                    'CAMERA_LIDAR_ALL': 5}),
]


def main():
    try:
        from waymo_open_dataset import dataset_pb2, label_pb2
        from waymo_open_dataset.protos import metrics_pb2, submission_pb2
    except ImportError as e:
        print('waymo_open_dataset is not installed here (%s): nothing checked.  The table in INTEGRATION.md stays "recalled".' % e)
        return 2
    from google.protobuf import descriptor as D
    tname = {D.FieldDescriptor.TYPE_MESSAGE: 'message', D.FieldDescriptor.TYPE_ENUM: 'enum', D.FieldDescriptor.TYPE_STRING: 'string',
             D.FieldDescriptor.TYPE_INT32: 'int32', D.FieldDescriptor.TYPE_INT64: 'int64', D.FieldDescriptor.TYPE_DOUBLE: 'double',
             D.FieldDescriptor.TYPE_FLOAT: 'float', D.FieldDescriptor.TYPE_BOOL: 'bool'}
    roots = {'Label': label_pb2.Label.DESCRIPTOR, 'Object': metrics_pb2.Object.DESCRIPTOR, 'Objects': metrics_pb2.Objects.DESCRIPTOR,
             'Submission': submission_pb2.Submission.DESCRIPTOR, 'CameraName': dataset_pb2.CameraName.DESCRIPTOR}

    def find(path):
        parts = path.split('.')
        d = roots[parts[0]]
        for p in parts[1:]:
            d = d.nested_types_by_name.get(p) or d.enum_types_by_name[p]
        return d
    bad = 0
    for _, msg, field, num, typ in RECALLED:
        f = find(msg).fields_by_name.get(field)
        # the wire format of an enum and of an int32 is the same varint: the recalled writer is right as long as number and wire type agree
        ok = f is not None and f.number == num and (tname.get(f.type) == typ or {tname.get(f.type), typ} <= {'enum', 'int32'})
        bad += not ok
        print('%-4s %-12s %-38s recalled %2d %-8s  installed %s' % ('ok' if ok else 'BAD', msg, field, num, typ,
                                                                  'missing' if f is None else '%d %s' % (f.number, tname.get(f.type))))
    for _, path, values in RECALLED_ENUMS:
        e = find(path)
        for k, v in values.items():
            ok = k in e.values_by_name and e.values_by_name[k].number == v
            bad += not ok
            print('%-4s enum %-26s %-20s recalled %d  installed %s' % ('ok' if ok else 'BAD', path, k, v,
                                                                      e.values_by_name[k].number if k in e.values_by_name else 'missing'))
    # round trip through the native writer
    import tempfile
    from waymo_2d_tracking_amd import waymo_proto as WP
    rows = [{'image_id': 'segment-1_with_camera_labels/1550083467346370/FRONT', 'bbox': [10, 20, 30, 40], 'category_id': 1, 'score': 0.5,
             'object_id': '7'},
            {'image_id': 'segment-1_with_camera_labels/1550083467446370/SIDE_LEFT', 'bbox': [1, 2, 3, 4], 'category_id': 4, 'score': 0.25,
             'object_id': '8'}]
    path = os.path.join(tempfile.mkdtemp(prefix='wt_proto_'), 'sub.bin')
    WP.write(path, WP.entries_to_columns(rows), submission=dict(task=WP.TRACKING_2D, account_name='a@b.c', unique_method_name='m', authors=['x'],
                                                                 affiliation='y', description='z', sensor_type=WP.CAMERA_ALL))
    sub = submission_pb2.Submission()
    sub.ParseFromString(open(path, 'rb').read())
    objs = sub.inference_results.objects
    ok = (len(objs) == 2 and objs[0].object.box.center_x == 25.0 and objs[0].object.box.length == 30.0 and objs[0].object.id == '7'
          and objs[1].camera_name == dataset_pb2.CameraName.SIDE_LEFT and objs[1].object.type == label_pb2.Label.TYPE_CYCLIST
          and abs(objs[1].score - 0.25) < 1e-7 and sub.task == submission_pb2.Submission.TRACKING_2D)
    bad += not ok
    print('%-4s round trip of a 2-object TRACKING_2D Submission through wt_waymo_objects_write' % ('ok' if ok else 'BAD'))
    print('RESULT:', 'schema pinned against the installed waymo_open_dataset' if not bad else '%d mismatches' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
