"""Synthetic JPEG files for the decoder tests (encoded with PIL, i.e. by the libjpeg-turbo whose decoder is the reference)."""
import io

import numpy as np


def synth(h, w, kind, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:                                                     # smooth ramps
        img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 3) % 256], -1)
    elif kind == 1:                                                   # white noise: long codes, every coefficient alive
        img = rng.integers(0, 256, (h, w, 3))
    else:                                                             # photo-like: low-frequency structure + sensor noise
        img = (128 + 100 * np.sin(xx[..., None] / 5.0 + np.arange(3)) * np.cos(yy[..., None] / 7.0)).clip(0, 255)
        img = img + rng.normal(0, 12, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def encode(arr, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, 'JPEG', **kw)
    return buf.getvalue()


def pil_rgb(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))


def small_cases():
    """(name, file bytes): sizes around the MCU edges x subsampling x quality x restart interval x optimised tables."""
    out = []
    sizes = [(16, 16), (8, 8), (1, 1), (2, 3), (17, 33), (48, 64), (50, 70), (5, 100), (100, 5), (31, 47), (64, 3), (3, 64), (9, 6)]
    for i, (h, w) in enumerate(sizes):
        for sub in (0, 1, 2):
            kind, q, ri = (i + sub) % 3, (30, 75, 95, 100)[(i + sub) % 4], (0, 1, 3)[(i + 2 * sub) % 3]
            kw = dict(quality=q, subsampling=sub)
            if ri:
                kw['restart_marker_blocks'] = ri
            out.append(('%dx%d_s%d_q%d_r%d_k%d' % (h, w, sub, q, ri, kind), encode(synth(h, w, kind, seed=i), **kw)))
    for h, w in ((16, 16), (17, 33), (5, 7)):
        out.append(('gray_%dx%d' % (h, w), encode(synth(h, w, 2, seed=5)[..., 0], quality=80)))
    out.append(('optimised_40x56', encode(synth(40, 56, 2, seed=9), quality=85, optimize=True)))
    out.append(('optimised_noise_33x20', encode(synth(33, 20, 1, seed=9), quality=100, subsampling=0, optimize=True)))
    return out


def medium_cases():
    """larger files (thousands of subsequences, several workgroups of the synchronisation kernel)"""
    out = []
    for (h, w, sub, q, ri, kind) in [(240, 320, 2, 90, 0, 2), (241, 323, 2, 75, 0, 2), (200, 300, 0, 100, 0, 1), (300, 200, 1, 95, 0, 1),
                                     (256, 384, 2, 85, 5, 2), (886, 1920, 2, 90, 0, 2), (480, 640, 2, 100, 0, 1)]:
        kw = dict(quality=q, subsampling=sub)
        if ri:
            kw['restart_marker_blocks'] = ri
        out.append(('%dx%d_s%d_q%d_r%d_k%d' % (h, w, sub, q, ri, kind), encode(synth(h, w, kind, seed=h), **kw)))
    return out


def real_world_files(limit=16):
    """JPEG files that ship with Python packages of this image (matplotlib, scikit-learn, scikit-image, ... sample data): written by
    other encoders than PIL's - optimised Huffman tables, other quantisation tables, restart intervals, progressive ones.  The GPU box
    runs the same image, so the same files exist there.  [] if none is found."""
    import glob
    pats = ['/usr/local/lib/python3*/dist-packages/*/datasets/images/*.jpg', '/usr/local/lib/python3*/dist-packages/matplotlib/mpl-data/sample_data/*.jpg',
            '/opt/conda/lib/python3*/site-packages/skimage/data/*.jpg', '/opt/conda/lib/python3*/site-packages/*/static/images/*.jpg',
            '/opt/conda/lib/python3*/site-packages/*/static/images/logos/*.jpg', '/opt/conda/lib/python3*/site-packages/nbconvert/tests/files/*.jpeg',
            '/opt/conda/lib/python3*/site-packages/IPython/core/tests/*.jpg', '/opt/conda/doc/global/template/images/*.jpg',
            '/usr/share/javascript/highlight.js/styles/*.jpg']
    out = []
    for p in pats:
        for f in sorted(glob.glob(p)):
            if f not in out:
                out.append(f)
    return out[:limit]
