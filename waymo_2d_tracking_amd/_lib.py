"""ctypes binding of libwaymotrack.so (include/waymotrack.h, include/waymodet.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``python -m waymo_2d_tracking_amd.build``
(hipcc --offload-arch=gfx950).  Loading fails loudly: there is no CPU or PyTorch fallback for any operator.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('WT_LIB_PATH') or os.path.join(HERE, 'csrc', 'libwaymotrack.so')    # WT_LIB_PATH: kernel experiments

WT_OK = 0
_STATUS = {1: 'WT_ERR_INVALID', 2: 'WT_ERR_NO_DEVICE', 3: 'WT_ERR_HIP', 4: 'WT_ERR_CAPACITY', 5: 'WT_ERR_NUMERIC'}


class WaymoTrackError(RuntimeError):
    pass


class TrackParams(C.Structure):
    _fields_ = [('max_age', C.c_int32), ('min_hits', C.c_int32), ('n_classes', C.c_int32), ('reserved', C.c_int32),
                ('score_threshold', C.c_void_p), ('iou_threshold', C.c_void_p)]


_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (SONAME libamdhip64.so.7) and loads
    it by file name; if libwaymotrack.so pulled /opt/rocm's copy first, a later ``import torch`` would load a second
    runtime and neither would see the other's devices/streams.  Loading torch's copy first (without importing
    torch) makes both resolve to the same library whatever the import order."""
    import importlib.util
    import sys
    if 'torch' in sys.modules:
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], 'lib')
    for name in ('libhsa-runtime64.so', 'libamdhip64.so'):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    """Load the HIP library (once).  Raises WaymoTrackError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WaymoTrackError(
                '%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % LIB_PATH)
        _preload_torch_hip_runtime()
        _lib = C.CDLL(LIB_PATH)
        _lib.wt_last_error.restype = C.c_char_p
        _lib.wt_idctr_create.restype = C.c_void_p
        _lib.wt_idctr_create.argtypes = [C.c_int64]
        _lib.wt_idctr_get.restype = C.c_int64
        _lib.wt_idctr_get.argtypes = [C.c_void_p]
        _lib.wt_idctr_set.argtypes = [C.c_void_p, C.c_int64]
        _lib.wt_idctr_destroy.argtypes = [C.c_void_p]
        _lib.wt_sort_destroy.argtypes = [C.c_void_p]
        _lib.wt_mct_destroy.argtypes = [C.c_void_p]
        _lib.wt_mct_tracker.restype = C.c_void_p
        _lib.wt_mct_tracker.argtypes = [C.c_void_p, C.c_int]
        _lib.wt_sort_num_tracks.argtypes = [C.c_void_p]
        for name in ('wt_track_streams_workspace', 'wt_ensemble_groups_workspace', 'wt_track_state_bytes',
                     'wt_track_chunk_workspace'):
            getattr(_lib, name).restype = C.c_size_t
        for name in ('wd_workspace_bytes', 'wd_nms_workspace', 'wd_rpn_topk_workspace', 'wd_gemm_nt_workspace', 'wd_deform_table_bytes',
                     'wd_deform_dw_scratch_floats', 'wd_deform_bwd_tables_bytes', 'wd_gemm_split_packed_bytes', 'wd_gemm_split_workspace', 'wd_split_planes_bytes'):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_size_t
    return _lib


def check(rc, what=''):
    if rc != WT_OK:
        msg = lib().wt_last_error()
        raise WaymoTrackError('%s failed: %s (%s)' % (what or 'libwaymotrack call', _STATUS.get(rc, rc),
                                                      msg.decode() if msg else ''))


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(C.c_void_p)


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(shape) if shape is not None else a


def device_info():
    n = C.c_int(0)
    ncu = C.c_int(0)
    arch = C.create_string_buffer(64)
    check(lib().wt_device_info(C.byref(n), arch, C.c_int(64), C.byref(ncu)), 'wt_device_info')
    return n.value, arch.value.decode(), ncu.value
