"""Round 6: the split-operand GEMM fed by pre-split activation planes (LDS-DMA loader) against the f32-A kernel on the same shape: results must be
bit-identical, timing by hipGraph replay.   python tools/gemm_planes_one.py M N K [iters] [epilogue 0|1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops

m, n, k = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
epi = int(sys.argv[5]) if len(sys.argv) > 5 else 0
torch.manual_seed(0)
a = torch.randn(m, k, device='cuda')
w = torch.randn(n, k, device='cuda') / k ** 0.5
bias = torch.randn(n, device='cuda') if epi else None
res = torch.randn(m, n, device='cuda') if epi else None
pw = ops.split_pack_weight(w)
ap = ops.split_planes_pack(a)
assert torch.equal(ops.split_planes_unpack(ap, m, k), a), 'pack / unpack round trip'
rp = ops.split_planes_pack(res) if epi else None

y0 = ops.gemm_split(a, pw, n, bias, res, bool(epi))
y1, p1 = ops.gemm_split_io(m, n, k, pw, a_planes=ap, bias=bias, residual=res, relu=bool(epi), want_out=True, want_planes=True)
torch.cuda.synchronize()
print('planes-A == f32-A output:', torch.equal(y0, y1), ' max |diff| %.3g' % float((y0 - y1).abs().max()))
print('out_planes == split(out):', torch.equal(ops.split_planes_unpack(p1, m, n), y1), ' planes bytes equal to a separate pack:',
      torch.equal(p1.view(-1)[: ops.split_planes_bytes(m // 32 * 32, n)], ops.split_planes_pack(y1).view(-1)[: ops.split_planes_bytes(m // 32 * 32, n)]))
if epi:
    y2, _ = ops.gemm_split_io(m, n, k, pw, a_planes=ap, bias=bias, residual_planes=rp, relu=True)
    print('planes residual == f32 residual:', torch.equal(y2, y0))
    # in place on the residual planes (the block output takes the residual's buffer)
    rp2 = rp.clone()
    _, p3 = ops.gemm_split_io(m, n, k, pw, a_planes=ap, bias=bias, residual_planes=rp2, relu=True, want_out=False, out_planes=rp2)
    print('in-place planes residual -> planes out:', torch.equal(ops.split_planes_unpack(p3, m, n), y0))
y3, _ = ops.gemm_split_io(m, n, k, pw, a=a, bias=bias, residual=res, relu=bool(epi))
print('io entry with f32 A == gemm_split:', torch.equal(y3, y0))


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


out = torch.empty(m, n, device='cuda')
op = ops.split_planes_empty(m, n, 'cuda')
t_f32 = timeit(lambda: ops.gemm_split(a, pw, n, bias, res, bool(epi), out=out))
t_pl = timeit(lambda: ops.gemm_split_io(m, n, k, pw, a_planes=ap, bias=bias, residual=res, relu=bool(epi), out=out))
t_pp = timeit(lambda: ops.gemm_split_io(m, n, k, pw, a_planes=ap, bias=bias, residual_planes=rp, relu=bool(epi), want_out=False, out_planes=op))
t_pack = timeit(lambda: ops.split_planes_pack(a, out=ap))
print('%dx%dx%d epi=%d: f32-A ring %.1f us | planes-A -> f32 out %.1f us | planes-A (+planes residual) -> planes out %.1f us | stand-alone pack of A %.1f us'
      % (m, n, k, epi, t_f32, t_pl, t_pp, t_pack))
