"""Cycle split of the ping-pong deformable conv (library built with WD_HIPCC_FLAGS=-DPP_PROF)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
os.environ['WD_DEFORM_PATCH'] = 'pp'
Cc, H, W = 1024, 80, 120
x = torch.randn(1, Cc, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
pw = ops.deform_pack_weight(torch.randn(Cc, 32, 3, 3, device='cuda'), 32)
for std in (0.0, 0.5, 2.0):
    off = (torch.randn(1, 18, H, W, device='cuda') * std).contiguous(memory_format=torch.channels_last)
    ops.deform_conv3x3(x, off, pw, 32, 1, 1); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    ops._lib.lib().wd_debug_pp_prof(buf, 1)
    ops.deform_conv3x3(x, off, pw, 32, 1, 1); torch.cuda.synchronize()
    ops._lib.lib().wd_debug_pp_prof(buf, 1)
    names = ['gather(blend+rest)', 'mfma', 'barrier', 'epilogue', 'prefetch/table', 'loop', 'g:table+B wait', 'g:addr+corner wait']
    nw = 256 * 8
    print('std', std, ' '.join('%s %.0f' % (n, buf[i] / nw) for i, n in enumerate(names)), '(cycles per wave)')
