"""Victim / aggressor: stream 0 repeats the res5 stride-2 deformable conv (deform_conv3x3_kernel<64, true>) on fixed inputs and checks every output
against the serial result; stream 1 runs ONE kind of kernel in a loop.  AGGRESSOR = split2 | split5 | gconv | mm | none"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops
torch.manual_seed(0)
AGG = os.environ.get('AGGRESSOR', 'split2')
o1 = (torch.randn(1, 2048, 16, 24, device='cuda') * 0.5).contiguous(memory_format=torch.channels_last)
off = (torch.randn(1, 18, 8, 12, device='cuda') * 0.3).contiguous(memory_format=torch.channels_last)
wgt = ops.deform_pack_weight(torch.randn(2048, 64, 3, 3, device='cuda') / 24, 32)
ref = ops.deform_conv3x3(o1, off, wgt, 32, 2, 1, None, None, True).clone()
a2 = torch.randn(6144, 256, device='cuda'); w2 = ops.split_pack_weight(torch.randn(256, 256, device='cuda') / 16)
a5 = torch.randn(38400, 512, device='cuda'); w5 = ops.split_pack_weight(torch.randn(512, 512, device='cuda') / 22)
xg = torch.randn(1, 256, 64, 96, device='cuda').contiguous(memory_format=torch.channels_last)
wg = ops.deform_pack_weight(torch.randn(256, 8, 3, 3, device='cuda') / 8.5, 32)
ma = torch.randn(6144, 256, device='cuda'); mb = torch.randn(256, 256, device='cuda')
sink = torch.zeros(4, dtype=torch.int32, device='cuda')
cflags = torch.zeros(8, dtype=torch.int32, device='cuda')


def aggress():
    if AGG == 'split2':
        ops.gemm_split(a2, w2, 256, None, None, True)
    elif AGG == 'split5':
        ops.gemm_split(a5, w5, 512, None, None, True)
    elif AGG == 'gconv':
        ops.deform_conv3x3(xg, None, wg, 32, 1, 1, None, None, True)
    elif AGG == 'mm':
        torch.mm(ma, mb.t())
    elif AGG.startswith('burn'):                         # burnK: 512 workgroups of the pure-register matrix-instruction burner, kind K (debug build)
        import ctypes as C
        from waymo_2d_tracking_amd import _lib
        _lib.check(_lib.lib().wd_debug_mfma_burn(C.c_int(int(os.environ.get('BURN_WGS', '512'))), C.c_int(int(AGG[4:])), C.c_int(int(os.environ.get('BURN_ITERS', '400'))),
                                                 C.c_void_p(sink.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'burn')
    elif AGG.startswith('canary'):                       # canaryN: N busy workgroups (VALU / f32 MFMA / LDS reads), 36 KB of LDS, few registers
        import ctypes as C
        from waymo_2d_tracking_amd import _lib
        _lib.check(_lib.lib().wd_debug_canary(C.c_int(int(AGG[6:])), C.c_int(int(os.environ.get('CANARY_LDS', '36864'))), C.c_int(int(os.environ.get('CANARY_SPINS', '30'))),
                                              C.c_void_p(cflags.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'canary')
    elif AGG.startswith('dirty'):                        # dirtyN: N workgroups that fill DIRTY_LDS bytes of LDS with NaN patterns and leave
        import ctypes as C
        from waymo_2d_tracking_amd import _lib
        nb = int(os.environ.get('DIRTY_LDS', '36864'))
        sink[1] = nb
        _lib.check(_lib.lib().wd_debug_occupy(C.c_int(int(AGG[5:])), C.c_int(nb), C.c_longlong(-300), C.c_void_p(sink.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'occupy')
    elif AGG.startswith('occupy'):                       # occupyN: N idle workgroups holding 112 KiB of LDS each for ~30 us
        import ctypes as C
        from waymo_2d_tracking_amd import _lib
        _lib.check(_lib.lib().wd_debug_occupy(C.c_int(int(AGG[6:])), C.c_int(114688), C.c_longlong(3000), C.c_void_p(sink.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'occupy')


VICTIM = os.environ.get('VICTIM', 'deform64')
if VICTIM == 'gconv':
    def victim():
        return ops.deform_conv3x3(xg, None, wg, 32, 1, 1, None, None, True)
elif VICTIM == 'mm':
    def victim():
        return torch.mm(ma, mb.t())
elif VICTIM == 'deform32':
    x32 = (torch.randn(1, 1024, 16, 24, device='cuda') * 0.5).contiguous(memory_format=torch.channels_last)
    off32 = (torch.randn(1, 18, 16, 24, device='cuda') * 0.3).contiguous(memory_format=torch.channels_last)
    w32 = ops.deform_pack_weight(torch.randn(1024, 32, 3, 3, device='cuda') / 17, 32)
    def victim():
        return ops.deform_conv3x3(x32, off32, w32, 32, 1, 1, None, None, True)
else:
    def victim():
        return ops.deform_conv3x3(o1, off, wgt, 32, 2, 1, None, None, True)
ref = victim().clone()
torch.cuda.synchronize()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
bad = 0
outs = []
for rep in range(200):
    with torch.cuda.stream(s1):
        for _ in range(3):
            aggress()
    with torch.cuda.stream(s0):
        outs.append(victim())
torch.cuda.synchronize()
shown = 0
for y in outs:
    if not torch.equal(y, ref):
        bad += 1
        if shown < 3:
            shown += 1
            yf, rf = y.permute(0, 2, 3, 1).reshape(-1), ref.permute(0, 2, 3, 1).reshape(-1)
            d = (yf != rf).nonzero().flatten()
            print('  differing elements: %d of %d; index range %d..%d; rows (pixel) %s; channel range %d..%d' % (
                len(d), yf.numel(), int(d.min()), int(d.max()), sorted(set((d // y.shape[1]).tolist()))[:12], int((d % y.shape[1]).min()), int((d % y.shape[1]).max())))
            print('  got', yf[d[:6]].tolist(), 'ref', rf[d[:6]].tolist())
print('aggressor %s victim %s: %d of %d victim launches differ' % (AGG, VICTIM, bad, len(outs)))
# round 6 (-DWD_VICTIM_CHECK build of det_gconv.hip): the grouped-conv victim compared its LDS patch with global memory after every inner tile
try:
    import ctypes as C
    import numpy as np
    from waymo_2d_tracking_amd import _lib
    fn = getattr(_lib.lib(), 'wd_debug_victim_counters', None)
    if fn is not None:
        c = np.zeros(4, dtype=np.uint32)
        fn(c.ctypes.data_as(C.c_void_p), C.c_int(1))
        print('victim self-check: %d patch float4 in LDS differ from global memory over %d inner tiles' % (c[0], c[1]))
except AttributeError:
    pass
