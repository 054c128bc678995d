"""The native Waymo protobuf writer (csrc/waymo_proto.hip) against google.protobuf's own serialiser on the same schema.

The waymo-open-dataset package is not in the image, so the schema here is the one recalled in csrc/waymo_proto.hip (field
numbers "parity unpinned"); what this pins is the encoding: proto2 presence, field order, varints, nesting - byte for byte -
and the drop-in scripts' logic (coco_to_waymo.py:16-82, generate_prediction_for_metrics.py:43-80 of the reference), which is
replayed below message by message with the dynamic classes."""
import json

import pytest

pb = pytest.importorskip('google.protobuf')
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory  # noqa: E402

F = descriptor_pb2.FieldDescriptorProto


def _schema():
    fdp = descriptor_pb2.FileDescriptorProto(name='waymo_recalled.proto', package='waymo.open_dataset', syntax='proto2')

    def message(parent, name, fields):
        m = parent.message_type.add() if hasattr(parent, 'message_type') else parent.nested_type.add()
        m.name = name
        for fname, num, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = tname
        return m

    O, R = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    label = message(fdp, 'Label', [('box', 1, F.TYPE_MESSAGE, O, '.waymo.open_dataset.Label.Box'), ('type', 3, F.TYPE_INT32, O, ''),
                                   ('id', 4, F.TYPE_STRING, O, ''), ('detection_difficulty_level', 5, F.TYPE_INT32, O, ''),
                                   ('tracking_difficulty_level', 6, F.TYPE_INT32, O, ''), ('num_lidar_points_in_box', 7, F.TYPE_INT32, O, '')])
    message(label, 'Box', [(n, i, F.TYPE_DOUBLE, O, '') for n, i in (('center_x', 1), ('center_y', 2), ('center_z', 3), ('width', 4),
                                                                      ('length', 5), ('height', 6), ('heading', 7))])
    message(fdp, 'Object', [('object', 1, F.TYPE_MESSAGE, O, '.waymo.open_dataset.Label'), ('score', 2, F.TYPE_FLOAT, O, ''),
                            ('overlap_with_nlz', 3, F.TYPE_BOOL, O, ''), ('context_name', 4, F.TYPE_STRING, O, ''),
                            ('frame_timestamp_micros', 5, F.TYPE_INT64, O, ''), ('camera_name', 6, F.TYPE_INT32, O, '')])
    message(fdp, 'Objects', [('objects', 1, F.TYPE_MESSAGE, R, '.waymo.open_dataset.Object')])
    message(fdp, 'Submission', [('task', 1, F.TYPE_INT32, O, ''), ('account_name', 2, F.TYPE_STRING, O, ''),
                                ('unique_method_name', 3, F.TYPE_STRING, O, ''), ('authors', 4, F.TYPE_STRING, R, ''),
                                ('affiliation', 5, F.TYPE_STRING, O, ''), ('description', 6, F.TYPE_STRING, O, ''),
                                ('method_link', 7, F.TYPE_STRING, O, ''), ('sensor_type', 8, F.TYPE_INT32, O, ''),
                                ('number_past_frames_exclude_current', 9, F.TYPE_INT32, O, ''),
                                ('number_future_frames_exclude_current', 10, F.TYPE_INT32, O, ''),
                                ('inference_results', 11, F.TYPE_MESSAGE, O, '.waymo.open_dataset.Objects')])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName('waymo.open_dataset.' + n))
            for n in ('Label', 'Object', 'Objects', 'Submission')}


CAMERAS = ['FRONT', 'FRONT_LEFT', 'FRONT_RIGHT', 'SIDE_LEFT', 'SIDE_RIGHT']


def _rows(tracking, n=57):
    import random
    rnd = random.Random(3 + tracking)
    rows = []
    for i in range(n):
        row = {'image_id': 'segment-%d_with_camera_labels/%d/%s' % (10 ** 15 + i // 9, 1550083467346370 + 100000 * i, CAMERAS[i % 5]),
               'bbox': [rnd.randint(0, 1800), rnd.randint(0, 1200), rnd.randint(0, 300), rnd.randint(0, 300)],
               'score': round(rnd.random(), 5), 'category_id': 1 + i % 4}
        if i % 7 == 0:
            row['bbox'] = [rnd.random() * 1000, rnd.random() * 1000, rnd.random() * 100, 0.0]      # float boxes, zero height
        if tracking:
            row['object_id'] = '%i' % (i // 3)
        rows.append(row)
    return rows


@pytest.mark.parametrize('tracking', [False, True])
def test_submission_bytes_equal_protobuf(tmp_path, tracking):
    from waymo_2d_tracking_amd import coco_to_waymo as M
    P = _schema()
    rows = _rows(tracking)
    files = []
    for k in range(2):                                              # the CLI concatenates several detection files
        f = tmp_path / ('det%d.json' % k)
        f.write_text(json.dumps(rows[k::2]))
        files.append(str(f))
    out = tmp_path / 'sub' / 'submission.bin'
    argv = files + ['--unique-method-name', 'x152-dconv', '--description', 'Cascade R-CNN + SORT ü', '--account-name', 'a@b.c',
                    '-o', str(out)] + (['--tracking'] if tracking else [])
    assert M.main(argv) == 0
    # the reference's create_pb_submission / create_pd_object, message by message
    sub = P['Submission']()
    sub.task = 3 if tracking else 1
    sub.account_name = 'a@b.c'
    sub.authors.append('Yuan Xu'); sub.authors.append('Erdene-Ochir Tuguldur')
    sub.affiliation = 'DAInamite'
    sub.unique_method_name = 'x152-dconv'
    sub.description = 'Cascade R-CNN + SORT ü'
    sub.method_link = ''
    sub.sensor_type = 3
    sub.number_past_frames_exclude_current = 0
    sub.number_future_frames_exclude_current = 0
    objects = P['Objects']()
    for d in rows[0::2] + rows[1::2]:
        ctx, stamp, cam = d['image_id'].split('/')
        o = P['Object']()
        o.context_name = ctx
        o.frame_timestamp_micros = int(stamp)
        o.camera_name = 1 + CAMERAS.index(cam)
        bbox = d['bbox']
        box = P['Label'].Box()
        box.center_x = bbox[0] + bbox[2] * 0.5
        box.center_y = bbox[1] + bbox[3] * 0.5
        box.length = bbox[2]
        box.width = bbox[3]
        o.object.box.CopyFrom(box)
        o.score = d['score']
        if 'object_id' in d:
            o.object.id = d['object_id']
        o.object.type = d['category_id']
        objects.objects.append(o)
    sub.inference_results.CopyFrom(objects)
    assert out.read_bytes() == sub.SerializeToString()


@pytest.mark.parametrize('kind', ['prediction', 'ground-truth'])
def test_metrics_objects_bytes_equal_protobuf(tmp_path, kind):
    from waymo_2d_tracking_amd import generate_prediction_for_metrics as M
    P = _schema()
    rows = _rows(True)
    if kind == 'ground-truth':
        for i, r in enumerate(rows):
            del r['score']
            if i % 2:
                r['tracking_difficulty_level'] = 1 + i % 2
            if i % 3:
                r['detection_difficulty_level'] = 1 + (i // 3) % 2
            if i % 5 == 0:
                del r['object_id']
    src = tmp_path / 'in.json'
    src.write_text(json.dumps({'annotations': rows} if kind == 'ground-truth' else rows))
    out = tmp_path / 'out.bin'
    assert M.main(['--type', kind, '--input', str(src), '--output', str(out)]) == 0
    objects = P['Objects']()
    for e in rows:
        ctx, stamp, cam = e['image_id'].split('/')
        o = P['Object']()
        o.context_name = ctx
        o.frame_timestamp_micros = int(stamp)
        o.camera_name = 1 + CAMERAS.index(cam)
        bbox = e['bbox']
        box = P['Label'].Box()
        box.center_x = bbox[0] + bbox[2] / 2
        box.center_y = bbox[1] + bbox[3] / 2
        box.center_z = 0
        box.length = bbox[2]
        box.width = bbox[3]
        box.height = 0
        box.heading = 0
        o.object.box.CopyFrom(box)
        o.object.type = {1: 1, 2: 2, 3: 3, 4: 4}[e['category_id']]
        if 'score' in e:
            o.score = e['score']
        if 'object_id' in e:
            o.object.id = e['object_id']
        if 'tracking_difficulty_level' in e:
            o.object.tracking_difficulty_level = e['tracking_difficulty_level']
        if 'detection_difficulty_level' in e:
            o.object.detection_difficulty_level = e['detection_difficulty_level']
        o.object.num_lidar_points_in_box = 100
        objects.objects.append(o)
    assert out.read_bytes() == objects.SerializeToString()


def test_empty_and_malformed_inputs(tmp_path):
    from waymo_2d_tracking_amd import waymo_proto as W
    assert W.write(tmp_path / 'empty.bin', W.entries_to_columns([])) == 0
    assert (tmp_path / 'empty.bin').read_bytes() == b''
    with pytest.raises(KeyError):
        W.entries_to_columns([{'image_id': 's/1/TOP', 'bbox': [0, 0, 1, 1], 'category_id': 1}])     # unknown camera
    with pytest.raises(ValueError):
        W.entries_to_columns([{'image_id': 's/x/FRONT', 'bbox': [0, 0, 1, 1], 'category_id': 1}])    # timestamp not an int
    with pytest.raises(AssertionError):
        W.entries_to_columns([{'image_id': 's/1/FRONT', 'bbox': [0, 0, 1, 1], 'category_id': 0}])    # TYPE_UNKNOWN
