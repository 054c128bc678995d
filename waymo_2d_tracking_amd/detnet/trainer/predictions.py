"""Columnar prediction store - the MI355X-native counterpart of /root/reference/detnet/trainer/predictions.py:7-98.

The reference keeps `{image_id: [per class ndarray (n, 5) [score, cx, cy, w, h] normalised]}` in per-process `shelve` files
that the parent merges by file name (trainer/test.py:227-276).  Here the same content lives in flat columns
(image index, class, score, cx, cy, w, h) that every rank fills for its shard and that travel to rank 0 in ONE tensor
gather (distributed.gather_columns_rank0: RCCL over xGMI) before they are handed to the JSON writer or the evaluator.
The mapping interface of the reference class (`len`, `[]`, iteration, `keys`, `update`, `save`, `open`) is kept.
"""
import pickle
from pathlib import Path

import numpy as np

COLUMNS = (('image', np.int32), ('cls', np.int32), ('score', np.float32), ('cx', np.float32), ('cy', np.float32),
           ('w', np.float32), ('h', np.float32))


class Predictions(object):
    def __init__(self, classnames=None, image_ids=None, columns=None, tested=None):
        """image_ids: global list of image ids the `image` column indexes; tested: bool mask of images that were run
        (an image without detections is still a tested sample, predictions.py:31-35 returns empty arrays for it)."""
        self.classnames = list(classnames) if classnames is not None else None
        self.image_ids = list(image_ids) if image_ids is not None else []
        self._index = {k: i for i, k in enumerate(self.image_ids)}
        self.columns = {k: np.zeros(0, dt) for k, dt in COLUMNS} if columns is None else \
            {k: np.ascontiguousarray(columns[k], dtype=dt) for k, dt in COLUMNS}
        self.tested = np.zeros(len(self.image_ids), bool) if tested is None else np.asarray(tested, bool).copy()
        self._pending = []
        self._order = None

    # ---- filling ----
    def _image_index(self, image_id):
        i = self._index.get(image_id)
        if i is None:
            i = len(self.image_ids)
            self.image_ids.append(image_id)
            self._index[image_id] = i
            self.tested = np.append(self.tested, False)
        return i

    def __setitem__(self, image_id, detections):
        """detections: per class ndarray (n, 5) [score, cx, cy, w, h] (Detectron2Det.predict's per-image result)."""
        i = self._image_index(str(image_id))
        if self.tested[i]:                                   # overwrite (predictions.py:43-45 dict semantics)
            self._flush()
            keep = self.columns['image'] != i
            self.columns = {k: v[keep] for k, v in self.columns.items()}
        self.tested[i] = True
        for c, d in enumerate(detections):
            d = np.asarray(d, dtype=np.float32).reshape(-1, 5)
            if len(d):
                self._pending.append((i, c, d))
        self._order = None

    def _flush(self):
        if not self._pending:
            return
        n = sum(len(d) for _, _, d in self._pending)
        new = {k: np.empty(n, dt) for k, dt in COLUMNS}
        o = 0
        for i, c, d in self._pending:
            m = len(d)
            new['image'][o:o + m] = i; new['cls'][o:o + m] = c
            for j, k in enumerate(('score', 'cx', 'cy', 'w', 'h')):
                new[k][o:o + m] = d[:, j]
            o += m
        self.columns = {k: np.concatenate((self.columns[k], new[k])) for k, _ in COLUMNS}
        self._pending = []

    # ---- mapping interface of the reference ----
    def __len__(self):
        return int(self.tested.sum())

    def keys(self):
        for i in np.nonzero(self.tested)[0]:
            yield self.image_ids[i]

    def _rows_of(self, i):
        self._flush()
        if self._order is None:
            self._order = np.argsort(self.columns['image'], kind='stable')
            self._starts = np.searchsorted(self.columns['image'][self._order], np.arange(len(self.image_ids) + 1))
        return self._order[self._starts[i]:self._starts[i + 1]]

    def __getitem__(self, image_id):
        i = self._index.get(str(image_id))
        if i is None or not self.tested[i]:
            return None
        rows = self._rows_of(i)
        cls = self.columns['cls'][rows]
        out = []
        for c in range(len(self.classnames)):
            r = rows[cls == c]
            out.append(np.stack([self.columns[k][r] for k in ('score', 'cx', 'cy', 'w', 'h')], axis=1) if len(r)
                       else np.zeros((0, 5), np.float32))
        return out

    def __iter__(self):
        for k in self.keys():
            yield k, self[k]

    def update(self, other):
        for k, v in other:
            self[k] = v

    # ---- shards <-> one store ----
    def shard_columns(self):
        """This rank's rows as plain arrays (+ the tested mask) for the collation gather."""
        self._flush()
        return dict(self.columns), self.tested.copy()

    @staticmethod
    def from_shards(classnames, image_ids, column_parts, tested_parts):
        cols = {k: np.concatenate([p[k] for p in column_parts]) if column_parts else np.zeros(0, dt) for k, dt in COLUMNS}
        tested = np.zeros(len(image_ids), bool)
        for t in tested_parts:
            tested |= np.asarray(t, bool)
        return Predictions(classnames, image_ids, cols, tested)

    # ---- persistence (`-o OUTPUT` writes OUTPUT/detections.pkl, trainer/test.py:272-276; `--resume` reads it) ----
    def save(self, filename):
        self._flush()
        state = dict(format='waymo_2d_tracking_amd.predictions/1', classnames=self.classnames, image_ids=self.image_ids,
                     columns=self.columns, tested=self.tested)
        Path(filename).parent.mkdir(parents=True, exist_ok=True)
        with open(filename, 'wb') as fp:
            pickle.dump(state, fp, protocol=4)

    @staticmethod
    def open(filename, mode='r', **_):
        path = Path(filename)
        if path.is_dir():
            path = path / 'detections.pkl'
        with open(path, 'rb') as fp:
            state = pickle.load(fp)
        if not isinstance(state, dict) or not str(state.get('format', '')).startswith('waymo_2d_tracking_amd.predictions/'):
            raise ValueError('%s is not a prediction store written by this package' % path)
        return Predictions(state['classnames'], state['image_ids'], state['columns'], state['tested'])
