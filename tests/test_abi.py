"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every symbol
the headers under include/ declare.  No compute call is made (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, 'include', header)).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(w[td]_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def lib():
    from waymo_2d_tracking_amd import build
    path = build.build(verbose=False)
    return ctypes.CDLL(path)


@pytest.mark.parametrize('header', [h for h in sorted(os.listdir(os.path.join(ROOT, 'include'))) if h.endswith('.h')])
def test_every_declared_symbol_is_exported(lib, header):
    names = _declared(header)
    assert len(names) >= 5
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_abi_version_and_no_device_error(lib):
    assert lib.wt_abi_version() >= 1
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    # without a GPU the operators must fail loudly (no CPU fallback)
    n = ctypes.c_int(0)
    rc = lib.wt_device_info(ctypes.byref(n), None, 0, None)
    assert rc == 2 and n.value == 0
    lib.wt_last_error.restype = ctypes.c_char_p
    assert b'no HIP device' in lib.wt_last_error()


def test_product_never_imports_oracle():
    bad = []
    pkg = os.path.join(ROOT, 'waymo_2d_tracking_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(d, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b|wt_oracle\.h|libwt_oracle', txt, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_split_pack_descriptor_layout_matches_the_header():
    """ops._pack_desc_dtype (the host-side array handed to wd_gemm_split_pack_batch) == struct WdSplitPackDesc of include/waymodet.h: same field order,
    offsets and size (the header is parsed, not restated)."""
    import ctypes as C
    import re
    from waymo_2d_tracking_amd.detnet.nn import ops
    text = open(os.path.join(ROOT, 'include', 'waymodet.h')).read()
    body = re.search(r'typedef struct WdSplitPackDesc \{(.*?)\} WdSplitPackDesc;', text, re.S).group(1)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        ctype = C.c_void_p if '*' in decl else (C.c_long if decl.startswith('long') else C.c_int)
        names = decl.replace('*', ' ').split(None, 2 if decl.startswith('const') else 1)[-1]
        for name in names.split(','):
            fields.append((name.strip(), ctype))

    class Desc(C.Structure):
        _fields_ = fields

    dt = ops._pack_desc_dtype()
    assert dt.itemsize == C.sizeof(Desc) == 80
    assert [n for n, _ in fields] == list(dt.names) == list(ops._PACK_DESC_FIELDS)
    for name, _ in fields:
        assert dt.fields[name][1] == getattr(Desc, name).offset, name


def test_split_io_struct_layout_matches_the_header():
    """ops._SplitIO (handed to wd_gemm_split_io) == struct WdSplitIO of include/waymodet.h: same field names in the same order, eight 8-byte fields."""
    import ctypes as C
    import re
    from waymo_2d_tracking_amd.detnet.nn import ops
    text = open(os.path.join(ROOT, 'include', 'waymodet.h')).read()
    body = re.search(r'typedef struct WdSplitIO \{(.*?)\} WdSplitIO;', text, re.S).group(1)
    names = [d.replace('*', ' ').split()[-1] for d in body.split(';') if d.strip()]
    assert names == [f[0] for f in ops._SplitIO._fields_]
    assert C.sizeof(ops._SplitIO) == 64
    for (name, ctype), decl in zip(ops._SplitIO._fields_, [d.strip() for d in body.split(';') if d.strip()]):
        assert (ctype is C.c_long) == decl.startswith('long'), (name, decl)


def test_product_library_exports_no_laboratory_entry_points(lib):
    """Round 6: canary / occupant / burner kernels and the per-workgroup stamps live in csrc/debug/ and libwaymotrack_debug.so only."""
    for name in ('wd_debug_canary', 'wd_debug_occupy', 'wd_debug_hold', 'wd_debug_mfma_burn', 'wd_gemm_split_debug_stamps'):
        assert not hasattr(lib, name), name
    from waymo_2d_tracking_amd import build
    dbg = ctypes.CDLL(build.build_debug(verbose=False))
    assert all(hasattr(dbg, n) for n in ('wd_debug_mfma_burn', 'wd_gemm_split_debug_stamps', 'wd_gemm_split_f32'))
