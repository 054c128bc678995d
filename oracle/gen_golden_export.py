"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/export_g6.json by calling the REFERENCE's own
COCODetection.load_prediction (/root/reference/detnet/data/coco.py:229-252) on fake per-class prediction arrays
(standing in for detectron2 outputs): the detection-JSON wire format (pixel scale, centre->left/top, int()
truncation, round(score, 5)).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_export.py
"""
import collections
import collections.abc
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = '/root/reference'


def main():
    collections.Iterable = collections.abc.Iterable          # py3.10 compat of coco.py:4

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    sys.path.insert(0, REF)
    import detnet  # noqa
    stub('detnet.nn').__path__ = [REF + '/detnet/nn']
    stub('detnet.trainer').__path__ = [REF + '/detnet/trainer']
    stub('detnet.trainer.data', WeightedRandomDataset=object)
    stub('detnet.data').__path__ = [REF + '/detnet/data']
    stub('detnet.data.anchor_box_dataset', AnchorBoxDataset=object)
    from detnet.data.coco import COCODetection

    rng = np.random.default_rng(66)
    classnames = ['vehicle', 'pedestrian', 'sign', 'cyclist']
    imgs = {'segA/1550000000000000/FRONT': dict(width=1920, height=1280),
            'segA/1550000000100000/SIDE_LEFT': dict(width=1920, height=886),
            'segB/1550000000000000/FRONT_RIGHT': dict(width=1920, height=1280)}
    preds = {}
    for image_id in imgs:
        per_class = []
        for c in range(4):
            n = int(rng.integers(0, 6))
            cxcy = rng.uniform(0.05, 0.95, (n, 2)); wh = rng.uniform(0.01, 0.3, (n, 2)); s = rng.uniform(0.01, 1, (n, 1))
            per_class.append(np.concatenate([s, cxcy, wh], axis=1).astype(np.float32))
        preds[image_id] = per_class

    class FakePredictions(dict):
        pass
    fp = FakePredictions(preds)
    fp.classnames = classnames
    fake = types.SimpleNamespace(coco=types.SimpleNamespace(imgs=imgs),
                                 get_category_id=lambda name: classnames.index(name) + 1)
    rows = COCODetection.load_prediction(fake, fp)
    rows = [dict(r, score=float(r['score'])) for r in rows]
    out = {'images': imgs, 'classnames': classnames,
           'predictions': {k: [a.tolist() for a in v] for k, v in preds.items()}, 'rows': rows}
    with open(os.path.join(REPO, 'tests', 'golden', 'export_g6.json'), 'wt') as f:
        json.dump(out, f)
    print('rows', len(rows))


if __name__ == '__main__':
    main()
