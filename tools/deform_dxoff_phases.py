"""Cycles per phase of deform_dxoff_kernel (tools only; needs a library built with -DFB_TIMING: tools/build_variant.sh timing "-DFB_TIMING"
det_deform_bwd.hip, WT_LIB_PATH=.../variants/lib_timing.so).  Sums over wave 0 of every workgroup, printed per workgroup."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd import _lib
from waymo_2d_tracking_amd.detnet.nn import ops

SC = float(os.environ.get('OFF_SCALE', '0.5'))
NAMES = ['wait at item start (+ tables / flush)', 'stage patch + dY fragments + barrier', 'MFMA + dOffset + dcol -> LDS (wave 0)',
         'barrier after the MFMA phase', 'gather (wave 0)', 'barrier after the gather (first half only)']
for (Cn, G, H, W) in ((1024, 32, 56, 80), (512, 32, 112, 160)):
    x = torch.randn(1, Cn, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    off = (torch.randn(1, 18, H, W, device='cuda') * SC).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(1, H, W, Cn, device='cuda')
    w = torch.randn(Cn, Cn // G, 3, 3, device='cuda') * 0.05
    ops.deform_dxoff(x, off, dy, w, G)
    torch.cuda.synchronize()
    t = (C.c_ulonglong * 8)()
    L = _lib.lib()
    L.wd_deform_fb_ticks(t, 1)
    ops.deform_dxoff(x, off, dy, w, G)
    torch.cuda.synchronize()
    L.wd_deform_fb_ticks(t, 1)
    nwg = min(512, (H // 8) * (W // 8) * G)
    tot = sum(t[:6])
    print('C=%d %dx%d: %d workgroups, %.0f s_memtime ticks per workgroup (wave 0)' % (Cn, H, W, nwg, tot / nwg))
    for k in range(6):
        print('   %5.1f %%  %9.0f  %s' % (100.0 * t[k] / tot, t[k] / nwg, NAMES[k]))
