"""One shape of the split-operand GEMM, N launches (the program behind tools/pmc_split.sh and rocprofv3 --kernel-trace runs).
    python tools/gemm_split_one.py M N K [iters] [epilogue 0|1] [conv: b c h w n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from waymo_2d_tracking_amd.detnet.nn import ops

m, n, k = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
epi = int(sys.argv[5]) if len(sys.argv) > 5 else 0
torch.manual_seed(0)
a = torch.randn(m, k, device='cuda')
w = torch.randn(n, k, device='cuda') / k ** 0.5
bias = torch.randn(n, device='cuda')
res = torch.randn(m, n, device='cuda')
pw = ops.split_pack_weight(w)
out = torch.empty(m, n, device='cuda')
def run():
    ops.gemm_split(a, pw, n, bias if epi else None, res if epi else None, bool(epi), out=out)


for _ in range(3):
    run()
torch.cuda.synchronize()
# the launches are replayed from a captured hipGraph (as in the pipeline): an eager python loop adds ~20 us of launch gap per call
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    run()
    with torch.cuda.graph(g, stream=st):
        for _ in range(iters):
            run()
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
g.replay()
e1.record()
torch.cuda.synchronize()
print('gemm_split %dx%dx%d epi=%d: %.1f us per launch (graph replay of %d)' % (m, n, k, epi, e0.elapsed_time(e1) / iters * 1e3, iters))
