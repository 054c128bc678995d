"""profiles/r05_split_gemm_error.txt: the error gate of the split-operand kernel (csrc/det_gemm_split.hip).

(1) per detector shape: max / rms error against float64 of the split-operand kernel, of the f32 library GEMM (hipBLASLt) and of the hand-written
    exact-f32 MFMA kernel (wd_gemm_nt_f32, one fmaf chain over K), and the ratios;
(2) whole detector at 1920x1280 (random-init X152, two synthetic frames): end boxes / scores of the split-operand graph against the all-exact-f32
    graph (WD_SPLIT_GEMM=0), next to the drift between TWO exact-f32 graphs that differ only in the f32 GEMM implementation (hipBLASLt vs the
    hand-written f32 kernel) - the noise floor any change of summation order produces.
    python tools/split_error_table.py > profiles/r05_split_gemm_error.txt
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from waymo_2d_tracking_amd import _lib
from waymo_2d_tracking_amd.detnet.nn import cascade_rcnn, ops


def err(y, ref):
    d = y.double() - ref
    return float(d.abs().max()), float(d.pow(2).mean().sqrt())


def chain_f32(a, w):
    m, k = a.shape
    n = w.shape[0]
    out = torch.empty(m, n, device=a.device)
    zero = torch.zeros(m, n, device=a.device)       # keeps wd_gemm_nt_f32 off its split-K branch (+ 0.0 is exact)
    _lib.check(_lib.lib().wd_gemm_nt_f32(C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), None, C.c_void_p(zero.data_ptr()), C.c_int(0), C.c_int(m),
                                         C.c_int(n), C.c_int(k), C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'gemm_nt')
    return out


def main():
    torch.manual_seed(0)
    print('split-operand kernel (f32 operands as exact 3 x bf16 planes, 6 cross terms, f32 accumulate) - error against float64')
    print('%-34s %-23s %-23s %-23s %s' % ('shape (M x N x K)', 'split max / rms', 'hipBLASLt f32 max / rms', 'f32 MFMA chain max / rms', 'rms ratio vs lib / chain; max ratio vs lib / chain'))
    shapes = [('res4 1x1', 9600, 1024, 1024), ('res3 1x1', 38400, 512, 512), ('res2 1x1', 153600, 256, 256), ('res5 1x1', 2400, 2048, 2048),
              ('res5 conv1', 2400, 2048, 1024), ('fpn lateral p4', 9600, 256, 1024), ('fpn lateral p5 (K-sliced)', 2400, 256, 2048),
              ('box-head FC (K-sliced)', 1000, 1024, 12544)]
    for name, m, n, k in shapes:
        a = torch.randn(m, k, device='cuda')
        w = torch.randn(n, k, device='cuda') / k ** 0.5
        ref = a.double() @ w.double().t()
        es = err(ops.gemm_split(a, ops.split_pack_weight(w), n), ref)
        el = err(a @ w.t(), ref)
        ec = err(chain_f32(a, w), ref)
        print('%-34s %.3e / %.3e   %.3e / %.3e   %.3e / %.3e   %.2f / %.2f; %.2f / %.2f' % ('%s %dx%dx%d' % (name, m, n, k), es[0], es[1], el[0], el[1], ec[0], ec[1],
                                                                                              es[1] / el[1], es[1] / ec[1], es[0] / el[0], es[0] / ec[0]))
        del a, w, ref
    for name, b, c, h, w_, n in [('box-head 3x3', 1000, 256, 7, 7, 256), ('fpn out p3 3x3', 1, 256, 160, 240, 256), ('fpn out p5 3x3 (K-sliced)', 1, 256, 40, 60, 256)]:
        x = torch.randn(b, c, h, w_, device='cuda').contiguous(memory_format=torch.channels_last)
        wt = torch.randn(n, c, 3, 3, device='cuda') / (9 * c) ** 0.5
        ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
        es = err(ops.conv_split(x, ops.split_pack_weight(wt), n, 3, 1, 1), ref)
        el = err(F.conv2d(x, wt, None, 1, 1), ref)
        cols = F.unfold(x.contiguous(), 3, 1, 1, 1).permute(0, 2, 1).reshape(-1, 9 * c).contiguous()
        ec = err(chain_f32(cols, wt.reshape(n, -1).contiguous()).view(b, h, w_, n).permute(0, 3, 1, 2), ref)
        print('%-34s %.3e / %.3e   %.3e / %.3e   %.3e / %.3e   %.2f / %.2f; %.2f / %.2f' % ('%s %dx%dx%d' % (name, b * h * w_, n, 9 * c), es[0], es[1], el[0], el[1], ec[0], ec[1],
                                                                                              es[1] / el[1], es[1] / ec[1], es[0] / el[0], es[0] / ec[0]))
        del x, wt, ref, cols
    print('(the "lib" column of the 3x3 rows is MIOpen, whose solvers may sum in a tree; "chain" is the f32 fmaf chain over the unfolded input)')
    # integer GEMM: exact
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-40, 41, (333, 192), generator=g).float().cuda()
    w = torch.randint(-40, 41, (256, 192), generator=g).float().cuda()
    print('integer GEMM 333x256x192 exact: %s' % bool(torch.equal(ops.gemm_split(a, ops.split_pack_weight(w), 256).double(), a.double() @ w.double().t())))

    # whole detector
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
    m = Detectron2Det(seed=0).cuda().eval()
    gen = torch.Generator().manual_seed(1)
    imgs = [torch.randint(0, 256, (1, 3, 1280, 1920), generator=gen).float().cuda() for _ in range(2)]

    def run(split, library=True):
        cascade_rcnn.SPLIT_GEMM = split
        cascade_rcnn.Conv1x1.USE_LIBRARY_GEMM = library
        out = [m.predict_device(im)[0] for im in imgs]
        cascade_rcnn.SPLIT_GEMM, cascade_rcnn.Conv1x1.USE_LIBRARY_GEMM = True, True
        return out

    def drift(xs, ys):
        rows = []
        for (b0, s0, c0), (b1, s1, c1) in zip(xs, ys):
            n = min(len(b0), len(b1))
            same = bool(len(b0) == len(b1) and torch.equal(c0, c1))
            rows.append(dict(n0=len(b0), n1=len(b1), same_count_and_classes=same,
                             max_box_px=float((b0[:n] - b1[:n]).abs().max()) if n else 0.0, max_score=float((s0[:n] - s1[:n]).abs().max()) if n else 0.0))
        return rows
    exact = run(False, True)
    split = run(True, True)
    exact2 = run(False, False)
    print('whole detector, 1920x1280, 2 frames (row k of one graph against row k of the other):')
    print('  split-operand graph vs exact-f32 graph (hipBLASLt 1x1, MIOpen 3x3):', drift(split, exact))
    print('  exact-f32 graph with the hand-written f32 MFMA GEMM for the 1x1 convs vs exact-f32 graph with hipBLASLt:', drift(exact2, exact))
    print('  (both pairs differ only in f32 summation order / rounding; north_star tolerance on boxes and scores: 1e-4)')


if __name__ == '__main__':
    main()
