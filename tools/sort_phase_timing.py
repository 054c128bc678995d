"""Per-phase cycle split of the SORT kernel.  Experiments only: build with
    WD_HIPCC_FLAGS=-DWT_PHASE_TIMING python -m waymo_2d_tracking_amd.build --force
then run this script on the GPU (it runs the track stage of bench.py in-process and reads the counters)."""
import ctypes as C, os, runpy, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [os.path.join(ROOT, 'bench.py'), '--stage', 'track', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--segments', os.environ.get('SEGMENTS', '8')]
try:
    runpy.run_path(sys.argv[0], run_name='__main__')
except SystemExit:
    pass
from waymo_2d_tracking_amd import _lib
out = (C.c_ulonglong * 12)()
if not hasattr(_lib.lib(), 'wt_debug_phase_cycles'):
    sys.exit('library was not built with -DWT_PHASE_TIMING')
_lib.lib().wt_debug_phase_cycles(out, 0)
v = np.array(list(out), dtype=np.float64)
names = ['predict', 'iou matrix', 'munkres', 'match filter', 'kalman update', 'births + emit + reap']
tot = v[:6].sum()
for n, x in zip(names, v[:6]):
    print('%-22s %5.1f %%' % (n, 100 * x / tot))
for n, i in (('  munkres: step 1 (row minima, zero bitmaps)', 9), ('  munkres: greedy stars', 6), ('  munkres: steps 3-5 (cover / prime / augment)', 8), ('  munkres: step 6 (adjust)', 7)):
    print('%-46s %5.1f %%' % (n, 100 * v[i] / tot))
