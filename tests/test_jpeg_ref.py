"""CPU side of the JPEG decoder row: the sequential restatement (oracle/jpeg_ref.py) is pinned bit for bit
against PIL - the reference's own decoder (detnet/inference.py:170 ToRGB on PIL images) - and the parallel
synchronisation scheme of csrc/jpeg_decode.hip, emulated thread by thread on the CPU from the same header
(tests/native/jpeg_sync_emul.cpp + csrc/jpeg_core.h / jpeg_host.h), reproduces the sequential coefficients."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import jpeg_cases as JC                                                                      # noqa: E402
from oracle import jpeg_ref as J                                                             # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restatement_is_bit_exact_with_pil():
    bad = []
    for name, data in JC.small_cases():
        if not np.array_equal(J.decode_rgb(data), JC.pil_rgb(data)):
            bad.append(name)
    assert not bad, bad


def test_restatement_rejects_what_the_gpu_decoder_rejects():
    prog = JC.encode(JC.synth(32, 32, 2), quality=80, progressive=True)
    with pytest.raises(J.Unsupported):
        J.decode_rgb(prog)
    with pytest.raises(ValueError):
        J.decode_rgb(b'\x89PNG\r\n\x1a\n' + bytes(32))


@pytest.fixture(scope='module')
def emul(tmp_path_factory):
    so = str(tmp_path_factory.mktemp('jpeg') / 'libjpeg_emul.so')
    subprocess.run(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-o', so, os.path.join(ROOT, 'tests', 'native', 'jpeg_sync_emul.cpp')],
                   check=True)
    lib = ctypes.CDLL(so)

    def run(data, group=256):
        cap = 1 << 17
        coef = np.zeros((cap, 64), np.int16)
        rounds, total = ctypes.c_int(0), ctypes.c_int(0)
        geo = (ctypes.c_int * 10)()
        err = ctypes.create_string_buffer(256)
        rc = lib.jpeg_emul_coefficients(data, ctypes.c_long(len(data)), group, coef.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(cap),
                                        ctypes.byref(rounds), ctypes.byref(total), geo, err, 256)
        if rc:
            raise RuntimeError(err.value.decode())
        return coef[:total.value], rounds.value, list(geo)
    return run


def scan_order(info, planes):
    hmax, vmax, shape, mx, my = J.geometry(info)
    out = []
    for m in range(mx * my):
        my_i, mx_i = divmod(m, mx)
        for ci, (h, v) in enumerate(shape):
            for by in range(v):
                for bx in range(h):
                    out.append(planes[ci][my_i * v + by, mx_i * h + bx])
    return np.stack(out)


def test_parallel_entropy_decode_equals_sequential(emul):
    """every subsequence settles on the sequential decoder's state: coefficients identical, for workgroups of 256 threads
    (the GPU's) and of 4 (many more cross-workgroup hand-overs)"""
    cases = JC.small_cases() + JC.medium_cases()[:5]
    for name, data in cases:
        info = J.parse(data)
        exp = scan_order(info, J.decode_coefficients(info))
        for group in (256, 4):
            got, rounds, geo = emul(data, group)
            assert got.shape == exp.shape and np.array_equal(got, exp), (name, group)
            if group == 256:
                assert rounds == 2, (name, rounds)                     # candidate sets + the fixed two launches suffice
                if name.startswith('240x320'):
                    plain = emul(data, -256)
                    assert np.array_equal(plain[0], exp) and geo[8] <= 2 < plain[2][8], (geo[8], plain[2][8])   # nothing left to crawl


def test_emulation_reports_truncated_streams(emul):
    name, data = JC.medium_cases()[0]
    with pytest.raises(RuntimeError):
        emul(data[:len(data) // 2])


def test_header_parse_needs_no_gpu():
    """wd_jpeg_info is host-only (the C-ABI library loads and answers without a device)"""
    from waymo_2d_tracking_amd import _lib
    lib = _lib.lib()
    data = JC.encode(JC.synth(50, 70, 2), quality=90, subsampling=2, restart_marker_blocks=3)
    o = [ctypes.c_int32(0) for _ in range(6)]
    assert lib.wd_jpeg_info(data, ctypes.c_int64(len(data)), *[ctypes.byref(v) for v in o]) == 0
    assert [v.value for v in o][:5] == [70, 50, 3, 2, 2] and o[5].value > 0
    prog = JC.encode(JC.synth(32, 32, 2), quality=80, progressive=True)
    assert lib.wd_jpeg_info(prog, ctypes.c_int64(len(prog)), *[ctypes.byref(v) for v in o]) != 0
    assert b'progressive' in lib.wt_last_error()
    # samples stored as RGB (Adobe marker, transform 0): decoding them as YCbCr would give wrong colours without any error
    rgb = JC.encode(JC.synth(32, 32, 2), quality=80, keep_rgb=True)
    assert lib.wd_jpeg_info(rgb, ctypes.c_int64(len(rgb)), *[ctypes.byref(v) for v in o]) != 0
    assert b'unsupported: RGB-coded' in lib.wt_last_error()
    with pytest.raises(J.Unsupported):
        J.decode_rgb(rgb)
    # 3x1 sampling factors on all three components (ratio 4:4:4, but 9 blocks per MCU): the candidate kernel starts one decode per
    # block-in-MCU index with CAND_SLOTS = 8 threads per subsequence - such files go to PIL instead of leaving candidate records unwritten
    # (advisor, round 3).  The header of a 4:4:4 file is patched; the parser reads nothing else.
    base = bytearray(JC.encode(JC.synth(32, 48, 2), quality=80, subsampling=0))
    sof = base.find(b'\xff\xc0')
    assert sof > 0 and base[sof + 9] == 3
    for c in range(3):
        assert base[sof + 11 + 3 * c] == 0x11
        base[sof + 11 + 3 * c] = 0x31
    wide = bytes(base)
    assert lib.wd_jpeg_info(wide, ctypes.c_int64(len(wide)), *[ctypes.byref(v) for v in o]) != 0
    assert b'unsupported: more than 8 blocks per MCU' in lib.wt_last_error()


def test_restatement_on_files_written_by_other_encoders():
    """sample images shipped with the Python packages of this image (not PIL-encoded): the restatement equals PIL on every
    baseline file (only the smaller ones: the Huffman stage of the restatement is a Python loop)"""
    files = JC.real_world_files()
    if not files:
        pytest.skip('no JPEG sample files on this machine')
    checked = 0
    for f in files:
        data = open(f, 'rb').read()
        try:
            info = J.parse(data)
        except J.Unsupported:
            continue
        if info['width'] * info['height'] > 450 * 450:
            continue
        assert np.array_equal(J.decode_rgb(data), JC.pil_rgb(data)), f
        checked += 1
    assert checked >= 1


FUZZ = r'''
import sys, ctypes, numpy as np
sys.path.insert(0, sys.argv[2])
import jpeg_cases as JC
lib = ctypes.CDLL(sys.argv[1])
cap = 1 << 16
coef = np.zeros((cap, 64), np.int16)
def run(data):
    rounds, total = ctypes.c_int(0), ctypes.c_int(0); geo = (ctypes.c_int * 10)(); err = ctypes.create_string_buffer(256)
    return lib.jpeg_emul_coefficients(data, ctypes.c_long(len(data)), 256, coef.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(cap),
                                      ctypes.byref(rounds), ctypes.byref(total), geo, err, 256)
rng = np.random.default_rng(0)
seeds = [d for _, d in JC.small_cases()[:12]] + [JC.medium_cases()[0][1]]
ok = 0
for it in range(1500):
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
        i = rng.integers(0, len(d)); d[i:i] = bytes(rng.integers(0, 256, rng.integers(1, 8)).tolist())
    ok += run(bytes(d)) == 0
print('FUZZ-DONE', ok)
'''


def test_corrupt_files_stay_inside_their_buffers(tmp_path):
    """host parser, unstuffing and the decode loop (the code the kernels run) under AddressSanitizer + UBSan on 1500 mutated
    files: byte flips, truncation, header damage, insertions.  (GPU sanitizers are not available on the pool; this is the CPU
    build of the same sources.)"""
    asan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.exists(asan):
        pytest.skip('no libasan')
    so = str(tmp_path / 'libjpeg_emul_asan.so')
    subprocess.run(['g++', '-O1', '-g', '-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-std=c++17', '-shared', '-fPIC', '-o', so,
                    os.path.join(ROOT, 'tests', 'native', 'jpeg_sync_emul.cpp')], check=True)
    script = tmp_path / 'fuzz.py'
    script.write_text(FUZZ)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1')
    r = subprocess.run([sys.executable, str(script), so, os.path.join(ROOT, 'tests')], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and 'FUZZ-DONE' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-4000:]
