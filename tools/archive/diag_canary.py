"""Canary workgroups (self-checking LDS pattern / registers / VALU chain / f32 MFMA chain) on stream 0 while stream 1 loops a kernel.
AGGRESSOR = split2 | split5 | none"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd import _lib
from waymo_2d_tracking_amd.detnet.nn import ops
AGG = os.environ.get('AGGRESSOR', 'split2')
L = _lib.lib()
a2 = torch.randn(6144, 256, device='cuda'); w2 = ops.split_pack_weight(torch.randn(256, 256, device='cuda') / 16)
a5 = torch.randn(38400, 512, device='cuda'); w5 = ops.split_pack_weight(torch.randn(512, 512, device='cuda') / 22)
flags = torch.zeros(5, dtype=torch.int32, device='cuda')
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for rep in range(60):
    with torch.cuda.stream(s1):
        for _ in range(4):
            if AGG == 'split2':
                ops.gemm_split(a2, w2, 256, None, None, True)
            elif AGG == 'split5':
                ops.gemm_split(a5, w5, 512, None, None, True)
    with torch.cuda.stream(s0):
        _lib.check(L.wd_debug_canary(C.c_int(256), C.c_int(int(os.environ.get('CANARY_LDS', '49152'))), C.c_int(40), C.c_void_p(flags.data_ptr()),
                                     C.c_void_p(s0.cuda_stream)), 'canary')
torch.cuda.synchronize()
f = flags.cpu().tolist()
print('aggressor %s: canary workgroups %d; mismatches: LDS %d, registers %d, VALU chain %d, MFMA chain %d' % (AGG, f[4], f[0], f[1], f[2], f[3]))
