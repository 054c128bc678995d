"""In-tree build of libwaymotrack.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m waymo_2d_tracking_amd.build [--force]

Each .hip translation unit is compiled to an object (in parallel) and linked into
waymo_2d_tracking_amd/csrc/libwaymotrack.so.  The tracking / ensemble units are built with
-ffp-contract=off because their results must be bit-identical to the reference's elementwise float64 / float32
arithmetic; the detector kernels keep FMA contraction.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libwaymotrack.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'

COMMON = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-fno-fast-math', '-Wall', '-Wno-unused-function']
COMMON += os.environ.get('WD_HIPCC_FLAGS', '').split()          # experiments (e.g. -DWD_DBG=1)
# unit -> extra flags
UNITS = {
    'api.hip': [],
    'json_io.hip': [],
    'waymo_proto.hip': [],
    'ensemble.hip': ['-ffp-contract=off'],
    'sort_engine.hip': ['-ffp-contract=off'],
    'sort_single.hip': ['-ffp-contract=off'],
    'det_roialign.hip': [],
    'det_nms.hip': [],
    'det_deform.hip': [],
    'det_deform_pp.hip': ['-fno-slp-vectorize'],      # explicit 2-vectors in the blend; the SLP pass hoists its tree to the LDS loads
    'det_gconv.hip': [],
    'det_gemm.hip': ['-munsafe-fp-atomics'],
    'det_gemm_lt.hip': [],
    'det_gemm_split.hip': [],          # fp32-equivalent GEMM / 3x3 convolution on the bf16 matrix cores (exact 3-way operand split)
    'det_misc.hip': [],
    'det_preprocess.hip': ['-ffp-contract=off'],
    'jpeg_decode.hip': [],
    'det_tail.hip': ['-ffp-contract=off'],
    'det_backward.hip': ['-munsafe-fp-atomics'],
    'det_deform_bwd.hip': ['-munsafe-fp-atomics'],     # fused deformable backward: global float atomics for the dW / dX / dOffset partial sums
}
# Laboratory kernels (canaries, occupants, the matrix-instruction burner of tools/costream/, per-workgroup stamps of the split kernel): never in the product
# library.  build_debug() makes csrc/libwaymotrack_debug.so = the product's objects, with det_gemm_split.hip rebuilt under -DWD_DEBUG and
# debug/debug_kernels.hip added (objects *.dbg.o); load it with WT_LIB_PATH.  `WD_DEBUG_BUILD=1 python -m waymo_2d_tracking_amd.build` builds both.
DEBUG_UNITS = {'det_gemm_split.hip': [], 'debug/debug_kernels.hip': []}
DEBUG_LIB = os.path.join(CSRC, 'libwaymotrack_debug.so')


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=True):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers += [os.path.join(HERE, '..', 'include', f) for f in os.listdir(os.path.join(HERE, '..', 'include'))]
    units = [u for u in UNITS if os.path.exists(os.path.join(CSRC, u))]
    objs = []
    jobs = []
    for u in units:
        src = os.path.join(CSRC, u)
        obj = os.path.join(CSRC, os.path.basename(u).replace('.hip', '.o'))
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            jobs.append([HIPCC] + COMMON + UNITS[u] + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stderr[-6000:]))
        if verbose and r.stderr.strip():
            print(r.stderr[-3000:])
        return 0

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        run([HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


def build_debug(force=False, verbose=True):
    """libwaymotrack_debug.so (see DEBUG_UNITS); builds the product library first."""
    build(force, verbose)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(CSRC, 'debug', 'waymodet_debug.h')]
    headers += [os.path.join(HERE, '..', 'include', f) for f in os.listdir(os.path.join(HERE, '..', 'include'))]
    objs = [os.path.join(CSRC, os.path.basename(u).replace('.hip', '.o')) for u in UNITS if u not in DEBUG_UNITS]
    jobs = []
    for u, flags in DEBUG_UNITS.items():
        src, obj = os.path.join(CSRC, u), os.path.join(CSRC, os.path.basename(u).replace('.hip', '.dbg.o'))
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            jobs.append([HIPCC] + COMMON + ['-DWD_DEBUG=1'] + UNITS.get(u, []) + flags + ['-c', src, '-o', obj])
    for cmd in jobs:
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stderr[-6000:]))
    if jobs or force or _newer(DEBUG_LIB, objs):
        r = subprocess.run([HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', DEBUG_LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stderr[-6000:])
    return DEBUG_LIB


if __name__ == '__main__':
    if os.environ.get('WD_DEBUG_BUILD') == '1':
        print(build_debug(force='--force' in sys.argv))
    else:
        build(force='--force' in sys.argv)
    print(LIB)
