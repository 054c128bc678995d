"""Synthetic Waymo-shaped inputs: detection streams, ensemble inputs, frames.

Pure numpy, no package-relative imports: the golden-fixture generators under
``oracle/`` load this file by path under two different interpreters (numpy 1.26 and
2.2); ``numpy.random.default_rng`` (PCG64) gives the same stream on both.

Conventions follow the reference's wire format
(/root/reference/waymo_to_coco.py:53, /root/reference/detnet/data/coco.py:249-251):
``image_id = "<segment>/<timestamp_micros>/<CAMERA>"``, ``category_id`` in 1..4,
``bbox = [int x, int y, int w, int h]``, ``score`` rounded to 5 decimals.
"""
import numpy as np

CAMERAS = ('FRONT', 'FRONT_LEFT', 'FRONT_RIGHT', 'SIDE_LEFT', 'SIDE_RIGHT')
# /root/reference/tracking/utils.py:11-17
IMAGE_SIZES = {
    'FRONT': (1920, 1280),
    'FRONT_LEFT': (1920, 1280),
    'FRONT_RIGHT': (1920, 1280),
    'SIDE_LEFT': (1920, 886),
    'SIDE_RIGHT': (1920, 886),
}
CLASS_IDS = np.array([1, 2, 4])
CLASS_P = np.array([0.6, 0.3, 0.1])


def make_objects(rng, n_objects, n_frames, width, height, lifetimes=False):
    """Constant-velocity ground-truth objects; returns dict of per-object arrays."""
    w = rng.uniform(20, 300, n_objects)
    h = rng.uniform(20, 300, n_objects)
    cx = rng.uniform(w / 2, width - w / 2)
    cy = rng.uniform(h / 2, height - h / 2)
    vx = rng.normal(0, 5, n_objects)
    vy = rng.normal(0, 5, n_objects)
    cls = CLASS_IDS[rng.choice(3, size=n_objects, p=CLASS_P)]
    if lifetimes:
        start = rng.integers(0, max(1, n_frames // 2), n_objects)
        length = rng.integers(max(2, n_frames // 6), n_frames, n_objects)
        end = np.minimum(start + length, n_frames)
    else:
        start = np.zeros(n_objects, dtype=np.int64)
        end = np.full(n_objects, n_frames, dtype=np.int64)
    return dict(w=w, h=h, cx=cx, cy=cy, vx=vx, vy=vy, cls=cls, start=start, end=end)


def stream_detections(rng, n_frames, n_objects, camera='FRONT', jitter=2.0, dropout=0.1,
                      clutter=0.1, lifetimes=False, score_lo=0.01, integer_boxes=True):
    """One (segment, camera) stream.  Returns per-detection arrays sorted by frame:
    frame (int, 0-based), cat, x, y, w, h, score (float64)."""
    width, height = IMAGE_SIZES[camera]
    obj = make_objects(rng, n_objects, n_frames, width, height, lifetimes)
    frames, cats, xs, ys, ws, hs, scores = [], [], [], [], [], [], []
    for f in range(n_frames):
        alive = (obj['start'] <= f) & (f < obj['end'])
        keep = alive & (rng.uniform(size=n_objects) >= dropout)
        idx = np.nonzero(keep)[0]
        cx = obj['cx'][idx] + obj['vx'][idx] * f + rng.normal(0, jitter, idx.size)
        cy = obj['cy'][idx] + obj['vy'][idx] * f + rng.normal(0, jitter, idx.size)
        w = obj['w'][idx] + rng.normal(0, jitter, idx.size)
        h = obj['h'][idx] + rng.normal(0, jitter, idx.size)
        c = obj['cls'][idx]
        n_clutter = rng.binomial(n_objects, clutter)
        cw = rng.uniform(20, 300, n_clutter)
        ch = rng.uniform(20, 300, n_clutter)
        ccx = rng.uniform(cw / 2, width - cw / 2)
        ccy = rng.uniform(ch / 2, height - ch / 2)
        cc = CLASS_IDS[rng.choice(3, size=n_clutter, p=CLASS_P)]
        cx = np.concatenate([cx, ccx]); cy = np.concatenate([cy, ccy])
        w = np.concatenate([w, cw]); h = np.concatenate([h, ch]); c = np.concatenate([c, cc])
        perm = rng.permutation(cx.size)
        cx, cy, w, h, c = cx[perm], cy[perm], w[perm], h[perm], c[perm]
        x = cx - w / 2
        y = cy - h / 2
        s = rng.uniform(score_lo, 1.0, cx.size)
        if integer_boxes:   # detnet/data/coco.py:250 int() truncation; :249 round(score, 5)
            x, y, w, h = np.trunc(x), np.trunc(y), np.trunc(w), np.trunc(h)
            s = np.round(s, 5)
        frames.append(np.full(cx.size, f, dtype=np.int64))
        cats.append(c.astype(np.int64)); xs.append(x); ys.append(y); ws.append(w); hs.append(h)
        scores.append(s)
    cat = lambda a: np.concatenate(a) if a else np.zeros(0)
    return dict(frame=cat(frames).astype(np.int64), cat=cat(cats).astype(np.int64), x=cat(xs), y=cat(ys),
                w=cat(ws), h=cat(hs), score=cat(scores))


def frame_timestamp(f):
    """Synthetic timestamp_micros of frame f (10 Hz, like Waymo)."""
    return 1550000000000000 + int(f) * 100000


def detections_json(streams, integer_boxes=True):
    """streams: list of (segment_id, camera, det-dict from stream_detections) -> COCO-style list."""
    out = []
    for segment, camera, d in streams:
        for i in range(d['frame'].size):
            if integer_boxes:
                bbox = [int(d['x'][i]), int(d['y'][i]), int(d['w'][i]), int(d['h'][i])]
            else:
                bbox = [float(d['x'][i]), float(d['y'][i]), float(d['w'][i]), float(d['h'][i])]
            out.append({'image_id': '%s/%i/%s' % (segment, frame_timestamp(d['frame'][i]), camera),
                        'category_id': int(d['cat'][i]), 'bbox': bbox, 'score': float(d['score'][i])})
    return out


def make_sequence_json(seed, n_segments=1, n_frames=198, n_objects=100, cameras=CAMERAS, **kw):
    """Config-1 style input: S segments x 5 cameras, ~100 boxes/frame."""
    rng = np.random.default_rng(seed)
    streams = []
    for s in range(n_segments):
        seg = 'segment-%05d_with_camera_labels' % s
        for cam in cameras:
            streams.append((seg, cam, stream_detections(rng, n_frames, n_objects, cam, **kw)))
    return detections_json(streams, integer_boxes=kw.get('integer_boxes', True))


def ensemble_group(rng, n_objects, k_inputs, jitter=3.0, width=1920, height=1280):
    """K jittered copies of the same objects of one class in one image ->
    list of K arrays (n_i, 5) [score, x_left, y_top, w, h] float64 (tie-free scores)."""
    w = rng.uniform(20, 300, n_objects)
    h = rng.uniform(20, 300, n_objects)
    cx = rng.uniform(w / 2, width - w / 2)
    cy = rng.uniform(h / 2, height - h / 2)
    base = rng.uniform(0.05, 1.0, n_objects)
    outs = []
    for _ in range(k_inputs):
        jx = cx + rng.normal(0, jitter, n_objects)
        jy = cy + rng.normal(0, jitter, n_objects)
        jw = np.maximum(w + rng.normal(0, jitter, n_objects), 2)
        jh = np.maximum(h + rng.normal(0, jitter, n_objects), 2)
        # tie-free by construction (no clipping plateaus): torch.sort's tie order is unspecified
        s = 0.05 + 0.9 * base + np.clip(rng.normal(0, 0.01, n_objects), -0.03, 0.03)
        outs.append(np.stack([s, jx - jw / 2, jy - jh / 2, jw, jh], axis=1))
    return outs


def ensemble_inputs_json(seed, n_images, k_inputs, n_objects=100, integer_boxes=True):
    """K submission lists (config 4): the same objects per image with independent jitter."""
    rng = np.random.default_rng(seed)
    subs = [[] for _ in range(k_inputs)]
    for im in range(n_images):
        image_id = 'segment-%05d_with_camera_labels/%i/FRONT' % (im // 198, frame_timestamp(im % 198))
        cls = CLASS_IDS[rng.choice(3, size=n_objects, p=CLASS_P)]
        groups = ensemble_group(rng, n_objects, k_inputs)
        for k in range(k_inputs):
            g = groups[k]
            for i in range(n_objects):
                if integer_boxes:
                    bbox = [int(g[i, 1]), int(g[i, 2]), int(g[i, 3]), int(g[i, 4])]
                else:
                    bbox = [float(v) for v in g[i, 1:5]]
                subs[k].append({'image_id': image_id, 'category_id': int(cls[i]), 'bbox': bbox,
                                'score': round(float(g[i, 0]), 5)})
    return subs


def synthetic_frames(seed, n, height=1280, width=1920):
    """uint8 U{0..255} frames (n, height, width, 3)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, height, width, 3), dtype=np.uint8)
