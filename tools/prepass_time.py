import os, sys, ctypes as C, torch
sys.path.insert(0, '/root/repo')
from waymo_2d_tracking_amd import _lib
lib = _lib.lib(); lib.wd_deform_table_bytes.restype = C.c_size_t
for (h, w) in ((80, 120), (160, 240)):
    partial = torch.randn((h * w, 176), device='cuda') * 0.05
    out = torch.empty((1, 18, h, w), device='cuda').contiguous(memory_format=torch.channels_last)
    table = torch.empty(int(lib.wd_deform_table_bytes(1, h, w)), dtype=torch.uint8, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: lib.wd_deform_offsets_table_f32(C.c_void_p(partial.data_ptr()), 176, None, 1, h, w, C.c_void_p(out.data_ptr()), C.c_void_p(table.data_ptr()), st)
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): f()
    e1.record(); torch.cuda.synchronize()
    print('offsets+table pre-pass %dx%d: %.2f us' % (h, w, e0.elapsed_time(e1) / 200 * 1e3))
