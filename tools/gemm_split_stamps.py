"""Where a launch of the split-operand GEMM spends its time, per workgroup: s_memtime stamps at start / main loop / epilogue / end
(wd_gemm_split_debug_stamps).   python tools/gemm_split_stamps.py M N K [epilogue]"""
# needs the debug library: WD_DEBUG_BUILD=1 python -m waymo_2d_tracking_amd.build, then WT_LIB_PATH=waymo_2d_tracking_amd/csrc/libwaymotrack_debug.so python tools/gemm_split_stamps.py ...

import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd import _lib
from waymo_2d_tracking_amd.detnet.nn import ops

m, n, k = (int(v) for v in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
torch.manual_seed(0)
a = torch.randn(m, k, device='cuda')
w = torch.randn(n, k, device='cuda') / k ** 0.5
bias = torch.randn(n, device='cuda')
res = torch.randn(m, n, device='cuda')
pw = ops.split_pack_weight(w)
out = torch.empty(m, n, device='cuda')
for _ in range(5):
    ops.gemm_split(a, pw, n, bias if epi else None, res if epi else None, bool(epi), out=out)
stamps = torch.zeros(8 * 4096, dtype=torch.int64, device='cuda')
L = _lib.lib()
L.wd_gemm_split_debug_stamps(C.c_void_p(stamps.data_ptr()))
ops.gemm_split(a, pw, n, bias if epi else None, res if epi else None, bool(epi), out=out)
torch.cuda.synchronize()
L.wd_gemm_split_debug_stamps(None)
s = stamps.view(-1, 8).cpu()
s = s[s[:, 0] != 0]
t0 = int(s[:, 0].min())
st, mn, ep, en = (s[:, i] - t0 for i in range(4))
r0 = int(s[:, 6].min())
print('wall (100 MHz s_memrealtime): first start -> last end %.1f us; workgroup lifetime median %.1f us; shader clock during a lifetime: median %.0f MHz'
      % ((int(s[:, 7].max()) - r0) / 100.0, float((s[:, 7] - s[:, 6]).float().median()) / 100.0,
         float(((s[:, 3] - s[:, 0]).float() / (s[:, 7] - s[:, 6]).float()).median()) * 100.0))
print('start spread (us): median %.1f max %.1f' % (float((s[:, 6] - r0).float().median()) / 100.0, float((s[:, 6] - r0).float().max()) / 100.0))
print('%d workgroups; s_memtime ticks (100 MHz constant clock if s_memtime is the REFCLK, else shader cycles)' % len(s))
for name, v in (('start', st), ('prologue', mn - st), ('main loop', ep - mn), ('epilogue', en - ep), ('end', en)):
    v = v.float()
    print('%-10s min %9.0f  median %9.0f  max %9.0f' % (name, v.min(), v.median(), v.max()))
byx = {}
for row in s.tolist():
    byx.setdefault(row[5], []).append(row[3] - t0)
print('end by XCC:', {k: max(v) for k, v in sorted(byx.items())})
