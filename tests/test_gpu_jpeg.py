"""GPU JPEG decoder (csrc/jpeg_decode.hip) against PIL - the decoder the reference uses
(detnet/inference.py:170 ToRGB on `Image.open`): bit-exact RGB on every case."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import jpeg_cases as JC                                                                      # noqa: E402

pytestmark = pytest.mark.gpu


def test_decode_is_bit_exact_with_pil_small_and_medium():
    from waymo_2d_tracking_amd.detnet.nn import ops
    bad = []
    for name, data in JC.small_cases() + JC.medium_cases():
        got, rounds = ops.jpeg_decode(data, return_rounds=True)
        exp = JC.pil_rgb(data)
        if got.shape != exp.shape or not np.array_equal(got.cpu().numpy(), exp) or rounds != 2:
            bad.append((name, rounds))
    assert not bad, bad


@pytest.mark.parametrize('h,w,sub,q', [(1280, 1920, 2, 90), (886, 1920, 2, 95), (1280, 1920, 0, 100)])
def test_decode_full_size_frames(h, w, sub, q):
    """BASELINE config sizes (front 1920x1280 and side 1920x886 cameras); quality 100 / 4:4:4 noise = 10 MB of scan data,
    the largest stream the path sees"""
    from waymo_2d_tracking_amd.detnet.nn import ops
    data = JC.encode(JC.synth(h, w, 1 if q == 100 else 2, seed=7), quality=q, subsampling=sub)
    got, rounds = ops.jpeg_decode(data, return_rounds=True)
    assert rounds == 2
    assert torch.equal(got.cpu(), torch.from_numpy(JC.pil_rgb(data).copy()))
    assert ops.jpeg_info(data)[:3] == (w, h, 3)


def test_decode_rejects_unsupported_flavours_loudly():
    from waymo_2d_tracking_amd.detnet.nn import ops
    from waymo_2d_tracking_amd._lib import WaymoTrackError
    with pytest.raises(WaymoTrackError, match='progressive'):
        ops.jpeg_decode(JC.encode(JC.synth(32, 32, 2), quality=80, progressive=True))
    data = JC.medium_cases()[0][1]
    with pytest.raises(WaymoTrackError, match='truncated|ends before'):
        ops.jpeg_decode(data[:len(data) // 2])
    with pytest.raises(WaymoTrackError):
        ops.jpeg_decode(b'not a jpeg at all')
    # and the decoder still works afterwards (contexts are released on the error paths)
    assert np.array_equal(ops.jpeg_decode(data).cpu().numpy(), JC.pil_rgb(data))
    # advisor, round 3: (a) two restart markers in a row = a restart segment without a byte: refused, not decoded to zero blocks
    rst = JC.encode(JC.synth(64, 96, 2), quality=85, subsampling=2, restart_marker_blocks=2)
    i = rst.find(b'\xff\xd0')
    assert i > 0
    doubled = rst[:i + 2] + b'\xff\xd1' + rst[i + 2:]
    with pytest.raises(WaymoTrackError, match='empty restart segment|restart markers'):
        ops.jpeg_decode(doubled)
    # (b) a few hundred bytes announcing a huge frame with a restart interval of one MCU: refused BEFORE multi-GB buffers are allocated
    # for segments the file cannot hold (the contexts never shrink)
    small = bytearray(JC.encode(JC.synth(16, 16, 2), quality=50, subsampling=0, restart_marker_blocks=1))
    sof = small.find(b'\xff\xc0')
    small[sof + 5:sof + 9] = (4000).to_bytes(2, 'big') + (4000).to_bytes(2, 'big')          # height, width
    before = torch.cuda.memory_reserved()
    with pytest.raises(WaymoTrackError, match='more restart segments than the file holds'):
        ops.jpeg_decode(bytes(small), out=torch.empty((4000, 4000, 3), dtype=torch.uint8, device='cuda'))
    assert np.array_equal(ops.jpeg_decode(data).cpu().numpy(), JC.pil_rgb(data))
    del before


def test_image_loader_decodes_jpeg_on_the_gpu(tmp_path):
    """the loader's thread pool: 4 threads, own streams, 12 files of different sizes - every frame equals PIL's decode, and
    the 'pil' decoder gives the same tensors"""
    from waymo_2d_tracking_amd.detnet.inference import ImageLoader
    items = []
    for i in range(12):
        h, w = 120 + 16 * i, 200 + 24 * (i % 5)
        path = str(tmp_path / ('%02d.jpg' % i))
        with open(path, 'wb') as f:
            f.write(JC.encode(JC.synth(h, w, 2, seed=i), quality=70 + 2 * i, subsampling=(2, 1, 0)[i % 3]))
        items.append((i, path))
    prog = str(tmp_path / 'p.jpg')                                     # progressive: not decoded on the GPU -> PIL, announced once
    with open(prog, 'wb') as f:
        f.write(JC.encode(JC.synth(64, 80, 2, seed=77), quality=80, progressive=True))
    items.append((98, prog))
    rgbj = str(tmp_path / 'r.jpg')                                     # RGB-coded samples (Adobe transform 0): PIL as well
    with open(rgbj, 'wb') as f:
        f.write(JC.encode(JC.synth(48, 40, 2, seed=78), quality=85, keep_rgb=True))
    items.append((97, rgbj))
    png = str(tmp_path / 'x.png')
    from PIL import Image
    Image.fromarray(JC.synth(40, 50, 0)).save(png)
    items.append((99, png))
    gpu = list(ImageLoader(items, workers=4, depth=6, decoder='gpu'))
    host = list(ImageLoader(items, workers=4, depth=6, decoder='pil'))
    assert [g[0] for g in gpu] == [i for i, _ in items]
    for (ia, ta, sa), (ib, tb, sb) in zip(gpu, host):
        assert ia == ib and sa == sb and ta.dtype == torch.uint8 and torch.equal(ta.cpu(), tb.cpu())


def test_decode_files_written_by_other_encoders():
    """sample images that ship with the Python packages of the image (matplotlib, scikit-learn, scikit-image ...): other encoders,
    optimised Huffman tables, a restart interval of 32, sizes up to 1411 x 1411 - bit-exact with PIL; progressive ones are refused"""
    from waymo_2d_tracking_amd.detnet.nn import ops
    from waymo_2d_tracking_amd._lib import WaymoTrackError
    files = JC.real_world_files()
    if not files:
        pytest.skip('no JPEG sample files on this machine')
    decoded = 0
    for f in files:
        data = open(f, 'rb').read()
        try:
            got = ops.jpeg_decode(data)
        except WaymoTrackError as e:
            assert 'unsupported' in str(e), (f, str(e))
            continue
        assert np.array_equal(got.cpu().numpy(), JC.pil_rgb(data)), f
        decoded += 1
    assert decoded >= 1


def test_corrupt_files_raise_or_decode_but_never_break_the_device():
    """300 mutated files (byte flips, truncation, header damage, insertions - the mutations of the CPU sanitizer run in
    tests/test_jpeg_ref.py) through the GPU decoder: each call either returns an image of the announced size or raises
    WaymoTrackError; afterwards the device still decodes a good file bit-exactly."""
    from waymo_2d_tracking_amd.detnet.nn import ops
    from waymo_2d_tracking_amd._lib import WaymoTrackError
    rng = np.random.default_rng(0)
    seeds = [d for _, d in JC.small_cases()[:12]] + [JC.medium_cases()[0][1]]
    ok = bad = 0
    for it in range(300):
        d = bytearray(seeds[it % len(seeds)])
        mode = it % 4
        if mode == 0:
            for _ in range(rng.integers(1, 6)):
                d[rng.integers(0, len(d))] = rng.integers(0, 256)
        elif mode == 1:
            d = d[:rng.integers(2, len(d))]
        elif mode == 2:
            d[rng.integers(0, min(len(d), 700))] = rng.integers(0, 256)
        else:
            i = rng.integers(0, len(d))
            d[i:i] = bytes(rng.integers(0, 256, rng.integers(1, 8)).tolist())
        try:
            w, h = ops.jpeg_info(bytes(d))[:2]
            if w * h > 4096 * 4096:
                continue                                                # a damaged size field: not worth the memory
            out = ops.jpeg_decode(bytes(d))
            assert tuple(out.shape) == (h, w, 3)
            ok += 1
        except WaymoTrackError:
            bad += 1
    torch.cuda.synchronize()
    assert ok > 0 and bad > 0
    name, data = JC.medium_cases()[0]
    assert np.array_equal(ops.jpeg_decode(data).cpu().numpy(), JC.pil_rgb(data))
