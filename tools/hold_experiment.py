"""How much does a side-stream kernel that merely HOLDS compute units cost the detector?  (tools only)
The e2e bench loses 4.8 % against the detector alone while the SORT chunk kernel (20 one-wave workgroups, ~40 KB of LDS each, 23-31 ms
per 10-frame chunk) runs next to it.  This script replaces SORT by wd_debug_hold with the same footprint and varies workgroups / LDS."""
# needs the debug library: WD_DEBUG_BUILD=1 python -m waymo_2d_tracking_amd.build, then WT_LIB_PATH=waymo_2d_tracking_amd/csrc/libwaymotrack_debug.so python tools/hold_experiment.py ...

import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd import _lib
from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline

pipe = DetectTrackPipeline(5, 2, seed=0)
lib = _lib.lib()
sink = torch.zeros(4, dtype=torch.int32, device='cuda')
side = torch.cuda.Stream()


def run(n_wg, lds, ms, steps=4):
    cyc = int(ms * 1e-3 * 2.2e9)
    for _ in range(2):
        pipe.step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if n_wg:
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
            _lib.check(lib.wd_debug_hold(C.c_int(n_wg), C.c_int(lds), C.c_longlong(cyc), C.c_void_p(sink.data_ptr()), C.c_void_p(side.cuda_stream)), 'hold')
        pipe.step(False)
    torch.cuda.synchronize()
    return 10 * steps / (time.perf_counter() - t0)


base = run(0, 0, 0)
print('detector alone: %.2f frames/s' % base)
for n_wg, lds, ms in ((20, 40960, 25), (20, 1024, 25), (5, 160 * 1024, 25), (20, 40960, 100), (20, 1024, 100), (2, 160 * 1024, 100), (64, 1024, 100)):
    v = run(n_wg, lds, ms)
    print('hold %3d workgroups x %6d B LDS for %3d ms per 10-frame step: %.2f frames/s (%.1f %%)' % (n_wg, lds, ms, v, 100 * (v / base - 1)), flush=True)
