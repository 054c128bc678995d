"""One LONG canary launch (self-checking LDS / registers / VALU / MFMA) on stream 0 while stream 1 issues split-operand launches all the time."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd import _lib
from waymo_2d_tracking_amd.detnet.nn import ops
L = _lib.lib()
a2 = torch.randn(6144, 256, device='cuda'); w2 = ops.split_pack_weight(torch.randn(256, 256, device='cuda') / 16)
flags = torch.zeros(5, dtype=torch.int32, device='cuda')
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(3):
    ops.gemm_split(a2, w2, 256, None, None, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s0):
    e0.record()
    _lib.check(L.wd_debug_canary(C.c_int(int(os.environ.get('CANARY_WGS', '128'))), C.c_int(51968), C.c_int(int(os.environ.get('CANARY_SPINS', '3000'))),
                                 C.c_void_p(flags.data_ptr()), C.c_void_p(s0.cuda_stream)), 'canary')
    e1.record()
n = 0
with torch.cuda.stream(s1):
    while not e1.query():
        for _ in range(8):
            ops.gemm_split(a2, w2, 256, None, None, True)
        n += 8
torch.cuda.synchronize()
f = flags.cpu().tolist()
print('canary ran %.1f ms next to %d split launches (MT forced by WD_SPLIT_MT=%s): mismatches LDS %d, registers %d, VALU chain %d, MFMA chain %d'
      % (e0.elapsed_time(e1), n, os.environ.get('WD_SPLIT_MT'), f[0], f[1], f[2], f[3]))
