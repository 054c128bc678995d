"""Split-operand GEMM / convolution (csrc/det_gemm_split.hip: f32 operands as exact 3 x bf16 planes, six cross terms on the bf16 matrix
cores, f32 accumulation) against float64 references.  The gate of the round-4 review: error vs float64 <= 1.25 x the exact-f32 kernels' on the
same inputs; integer GEMMs exact.  Replaces the 1x1 / dense 3x3 convolutions of the detector (logs/12442/job.log:534-546, 1109-1160)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from waymo_2d_tracking_amd.detnet.nn import ops as o
    return o


def _chain_f32(ops, a, w, bias=None):
    """The exact-f32 reference kernel of the gate: wd_gemm_nt_f32 (f32 MFMA, one sequential fmaf chain over K per output).  Its split-K form
    (two half-length chains for K >= 2048 on few tiles, and the v2 kernel's 8 slices) rounds less often than a chain, so it is pinned to one
    slice here: the gate compares the split-operand kernel with THE f32 GEMM, not with a particular summation tree."""
    import ctypes as C
    from waymo_2d_tracking_amd import _lib
    m, k = a.shape
    n = w.shape[0]
    out = torch.empty(m, n, device=a.device)
    zero = torch.zeros(m, n, device=a.device)          # a residual operand (+ 0.0, exact) keeps wd_gemm_nt_f32 off its split-K branch
    _lib.check(_lib.lib().wd_gemm_nt_f32(C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()) if bias is not None else None,
                                         C.c_void_p(zero.data_ptr()),
                                         C.c_int(0), C.c_int(m), C.c_int(n), C.c_int(k), C.c_void_p(out.data_ptr()),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'wd_gemm_nt_f32')
    return out


def _err(y, ref):
    d = y.double() - ref
    return float(d.abs().max()), float(d.pow(2).mean().sqrt())


@pytest.mark.parametrize('m,n,k', [(9600, 1024, 1024), (2400, 2048, 2048), (38400, 512, 512), (1000, 256, 64), (4480, 256, 256)])
def test_split_gemm_error_not_above_exact_f32(ops, m, n, k):
    """max and rms error against float64 <= 1.25 x the error of the exact-f32 GEMMs (library and hand-written f32 MFMA kernel)."""
    torch.manual_seed(m + n + k)
    a = torch.randn(m, k, device='cuda')
    w = torch.randn(n, k, device='cuda') / k ** 0.5
    ref = a.double() @ w.double().t()
    pw = ops.split_pack_weight(w)
    y = ops.gemm_split(a, pw, n)
    e_split = _err(y, ref)
    e_lib = _err(a @ w.t(), ref)                       # hipBLASLt f32: one fmaf chain over K per output (what the detector ran before round 5)
    e_own = _err(_chain_f32(ops, a, w), ref)           # csrc/det_gemm.hip v1 kernel: v_mfma_f32_16x16x4_f32, bitwise an fmaf chain over K
    scale = float(ref.abs().max())
    assert e_split[0] <= 2e-6 * scale * max(1.0, (k / 1024) ** 0.5) * 4, (e_split, scale)     # f32-roundoff class at all
    for e_ref in (e_lib, e_own):
        assert e_split[1] <= 1.25 * e_ref[1], (e_split, e_ref)
        assert e_split[0] <= 1.25 * e_ref[0], (e_split, e_ref)


def test_split_gemm_integer_operands_are_exact(ops):
    """Integer operands below 2^24 split exactly, every partial sum stays below 2^24: the result must be the exact integer matrix."""
    g = torch.Generator().manual_seed(3)
    m, n, k = 333, 256, 192
    a = torch.randint(-40, 41, (m, k), generator=g).float().cuda()
    w = torch.randint(-40, 41, (n, k), generator=g).float().cuda()
    a[0, 0] = 12345677.0                     # a 24-bit operand: needs all three planes
    w[:, 0] = 1.0
    ref = (a.double() @ w.double().t())
    y = ops.gemm_split(a, ops.split_pack_weight(w), n)
    assert torch.equal(y.double(), ref)


def test_split_gemm_planes_reconstruct_large_and_tiny_operands(ops):
    """Every f32 operand is hi + mid + lo exactly: a one-hot weight column copies A through the matrix cores bit for bit."""
    torch.manual_seed(5)
    m, k, n = 320, 128, 256
    a = (torch.randn(m, k, device='cuda') * torch.exp(torch.randn(m, k, device='cuda') * 8)).contiguous()
    w = torch.zeros(n, k, device='cuda')
    idx = torch.arange(n, device='cuda') % k
    w[torch.arange(n, device='cuda'), idx] = 1.0
    y = ops.gemm_split(a, ops.split_pack_weight(w), n)
    assert torch.equal(y, a[:, idx])
    # and the other way round: arbitrary weights times a one-hot activation row
    a2 = torch.zeros(m, k, device='cuda')
    a2[torch.arange(m, device='cuda'), torch.arange(m, device='cuda') % k] = 1.0
    w2 = torch.randn(n, k, device='cuda') * 1e-3
    y2 = ops.gemm_split(a2, ops.split_pack_weight(w2), n)
    assert torch.equal(y2, w2.t()[torch.arange(m, device='cuda') % k])


@pytest.mark.parametrize('m,n,k', [(1, 256, 64), (159, 256, 128), (161, 512, 64), (777, 32, 64), (500, 96, 192), (2401, 2048, 128)])
def test_split_gemm_ragged_shapes_and_fused_epilogue(ops, m, n, k):
    torch.manual_seed(m * 7 + n)
    a = torch.randn(m, k, device='cuda')
    w = torch.randn(n, k, device='cuda') / k ** 0.5
    bias = torch.randn(n, device='cuda')
    res = torch.randn(m, n, device='cuda')
    pw = ops.split_pack_weight(w)
    ref = a.double() @ w.double().t()
    for b, r, relu in ((None, None, False), (bias, None, True), (bias, res, True), (None, res, False)):
        exp = ref.clone()
        if b is not None:
            exp = exp + b.double()
        if r is not None:
            exp = exp + r.double()
        if relu:
            exp = exp.relu()
        y = ops.gemm_split(a, pw, n, b, r, relu)
        assert float((y.double() - exp).abs().max()) <= 1e-5, (m, n, k, b is not None, r is not None, relu)
    # in place on the residual buffer (the block output of the backbone) and a row-strided A view
    buf = res.clone()
    y = ops.gemm_split(a, pw, n, bias, buf, True, out=buf)
    assert y.data_ptr() == buf.data_ptr()
    assert float((buf.double() - (ref + bias.double() + res.double()).relu()).abs().max()) <= 1e-5
    wide = torch.randn(m, 2 * k, device='cuda')
    y = ops.gemm_split(wide[:, k:], pw, n)
    assert float((y.double() - wide[:, k:].double() @ w.double().t()).abs().max()) <= 1e-5


@pytest.mark.parametrize('b,c,h,w,n,ks,stride,pad', [(1, 64, 13, 17, 64, 3, 1, 1), (3, 128, 7, 7, 256, 3, 1, 1), (1, 256, 20, 30, 256, 3, 1, 1),
                                                     (2, 64, 16, 24, 128, 1, 2, 0), (1, 128, 15, 11, 32, 1, 1, 0), (1, 64, 12, 12, 64, 3, 2, 1)])
def test_split_conv_matches_float64_convolution(ops, b, c, h, w, n, ks, stride, pad):
    torch.manual_seed(c + h + w + n)
    x = torch.randn(b, c, h, w, device='cuda').contiguous(memory_format=torch.channels_last)
    wt = torch.randn(n, c, ks, ks, device='cuda') / (ks * ks * c) ** 0.5
    bias = torch.randn(n, device='cuda')
    ref = F.conv2d(x.double(), wt.double(), bias.double(), stride, pad)
    y = ops.conv_split(x, ops.split_pack_weight(wt), n, ks, stride, pad, bias)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    e = _err(y, ref)
    # exact-f32 reference of the gate: the same convolution as an f32 fmaf-chain GEMM over the unfolded input (k = (c, kh, kw))
    cols = F.unfold(x.contiguous(), ks, 1, pad, stride).permute(0, 2, 1).reshape(-1, c * ks * ks).contiguous()
    y32 = _chain_f32(ops, cols, wt.reshape(n, -1).contiguous(), bias).view(b, ref.shape[2], ref.shape[3], n).permute(0, 3, 1, 2)
    e32 = _err(y32, ref)
    assert e[0] <= 1e-5 and e[1] <= 1.25 * e32[1] and e[0] <= 1.25 * e32[0] + 2e-7, (e, e32)
    res = torch.randn_like(y)
    y2 = ops.conv_split(x, ops.split_pack_weight(wt), n, ks, stride, pad, bias, res, True)
    assert float((y2.double() - (ref + res.double()).relu()).abs().max()) <= 1e-5


def test_split_conv_box_head_shape_error_gate(ops):
    """The box-head convolution (1000 ROIs x 7 x 7 x 256, job.log:1146-1160) at full size: error gate against MIOpen's f32 result."""
    torch.manual_seed(11)
    x = torch.randn(1000, 256, 7, 7, device='cuda').contiguous(memory_format=torch.channels_last)
    wt = torch.randn(256, 256, 3, 3, device='cuda') / 48.0
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    y = ops.conv_split(x, ops.split_pack_weight(wt), 256, 3, 1, 1)
    cols = F.unfold(x.contiguous(), 3, 1, 1, 1).permute(0, 2, 1).reshape(-1, 2304).contiguous()
    y32 = _chain_f32(ops, cols, wt.reshape(256, -1).contiguous()).view(1000, 7, 7, 256).permute(0, 3, 1, 2)
    e, e32 = _err(y, ref), _err(y32, ref)
    assert e[1] <= 1.25 * e32[1] and e[0] <= 1.25 * e32[0], (e, e32)
    e_lib = _err(F.conv2d(x, wt, None, 1, 1), ref)          # MIOpen's solver of the day (informative: some solvers sum in a tree)
    print('box-head conv error vs f64: split %s, f32 chain %s, MIOpen %s' % (e, e32, e_lib))


@pytest.mark.parametrize('m,n,k', [(600, 256, 2304), (2400, 256, 2048), (1000, 1024, 12544), (150, 512, 1024)])
def test_split_gemm_k_sliced_small_shapes(ops, m, n, k):
    """Few output tiles (FPN p5 / p6, box-head FC): the kernel cuts K into slices, a second launch sums the partial tiles in slice order -
    same error gate, fused epilogue intact, bit-identical from run to run."""
    import ctypes as C
    from waymo_2d_tracking_amd import _lib
    assert int(_lib.lib().wd_gemm_split_workspace(C.c_long(m), C.c_int(n), C.c_int(k))) > 0          # this shape IS sliced
    torch.manual_seed(k + m)
    a = torch.randn(m, k, device='cuda')
    w = torch.randn(n, k, device='cuda') / k ** 0.5
    bias = torch.randn(n, device='cuda')
    res = torch.randn(m, n, device='cuda')
    pw = ops.split_pack_weight(w)
    ref = a.double() @ w.double().t()
    y = ops.gemm_split(a, pw, n)
    e, e32 = _err(y, ref), _err(_chain_f32(ops, a, w), ref)
    assert e[1] <= 1.25 * e32[1] and e[0] <= 1.25 * e32[0], (e, e32)
    y2 = ops.gemm_split(a, pw, n, bias, res, True)
    assert float((y2.double() - (ref + bias.double() + res.double()).relu()).abs().max()) <= 2e-5
    assert torch.equal(ops.gemm_split(a, pw, n, bias, res, True), y2)
    buf = res.clone()
    ops.gemm_split(a, pw, n, bias, buf, True, out=buf)                  # in place on the residual
    assert torch.equal(buf, y2)


def test_training_functions_forward_and_gradients_vs_float64(ops):
    """The autograd functions of the training graph on the split-operand kernel (LinearActFn: forward + dA; ConvSplitFn: forward + dX) against
    float64 autograd of the plain formulas; weight / bias gradients (library) ride along."""
    torch.manual_seed(21)
    m, n, k = 300, 512, 256
    a = torch.randn(m, k, device='cuda', requires_grad=True)
    w = (torch.randn(n, k, device='cuda') / k ** 0.5).requires_grad_()
    b = torch.randn(n, device='cuda', requires_grad=True)
    r = torch.randn(m, n, device='cuda', requires_grad=True)
    gy = torch.randn(m, n, device='cuda')
    y = ops.LinearActFn.apply(a, w, b, r, True)
    (y * gy).sum().backward()
    a64, w64, b64, r64 = (t.detach().double().requires_grad_() for t in (a, w, b, r))
    y64 = torch.relu(a64 @ w64.t() + b64 + r64)
    (y64 * gy.double()).sum().backward()
    assert float((y.double() - y64).abs().max()) <= 1e-5
    for got, ref in ((a.grad, a64.grad), (w.grad, w64.grad), (b.grad, b64.grad), (r.grad, r64.grad)):
        assert float((got.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), float((got.double() - ref).abs().max())
    # dense 3x3 convolution (box-head / FPN shape family), bias + ReLU fused
    x = torch.randn(3, 256, 9, 11, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_()
    wt = (torch.randn(256, 256, 3, 3, device='cuda') / 48.0).requires_grad_()
    bc = torch.randn(256, device='cuda', requires_grad=True)
    gy = torch.randn(3, 256, 9, 11, device='cuda')
    y = ops.ConvSplitFn.apply(x, wt, bc, 1, 1, True)
    (y * gy).sum().backward()
    x64, w64, b64 = (t.detach().double().requires_grad_() for t in (x, wt, bc))
    y64 = torch.relu(F.conv2d(x64, w64, b64, 1, 1))
    (y64 * gy.double()).sum().backward()
    assert float((y.double() - y64).abs().max()) <= 1e-5
    for got, ref in ((x.grad, x64.grad), (wt.grad, w64.grad), (bc.grad, b64.grad)):
        assert float((got.double() - ref).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max())), float((got.double() - ref).abs().max())


def test_split_gemm_rejects_unsupported_shapes(ops):
    from waymo_2d_tracking_amd._lib import WaymoTrackError
    a = torch.randn(64, 96, device='cuda')
    with pytest.raises((WaymoTrackError, ValueError)):
        ops.split_pack_weight(torch.randn(64, 96, device='cuda'))          # K % 64 != 0
    pw = ops.split_pack_weight(torch.randn(48, 128, device='cuda'))
    with pytest.raises(WaymoTrackError):
        ops.gemm_split(torch.randn(64, 128, device='cuda'), pw, 48)         # N % 32 != 0
    del a


def test_whole_detector_split_graph_vs_exact_f32_graph(monkeypatch):
    """whole-detector boxes of the split-operand graph against the all-exact-f32 graph (hipBLASLt / MIOpen f32).  Neither is the truth -
    both are float32 evaluations; the library graph differs from ITSELF run twice by up to 2 - 4 float32 spacings of a pixel coordinate (split-K atomics),
    and the split graph stays at that floor.  Full size (1920 x 1280, three seeds): profiles/r05_split_box_drift.txt (tools/split_box_drift.py) - same-proposal
    boxes <= 2.4e-4 px vs an exact-vs-exact floor of 1.2e-4 .. 2.4e-4 px, scores <= 2.4e-7, FPN features <= 3.7e-6 relative, no unmatched detection.
    (A bound of 1e-5 px is below the float32 spacing of the coordinates themselves: 6.1e-5 px at 1000 px.)  Here: 640 x 448, gates 8x above the measured values."""
    from waymo_2d_tracking_amd.detnet.nn import cascade_rcnn
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
    net = Detectron2Det(seed=2).eval().cuda().model
    g = torch.Generator().manual_seed(102)
    img = torch.randint(0, 256, (1, 3, 448, 640), generator=g).float().cuda()

    def run(split, proposals=None):
        monkeypatch.setattr(cascade_rcnn, 'SPLIT_GEMM', split)
        inter = {}
        out = net(img, proposals=proposals, intermediates=inter)
        return out, inter

    (eb, es, ec), ei = run(False)
    (sb, ss, sc), si = run(True)
    for a, b in zip(si['feats'], ei['feats']):
        assert float((a - b).abs().max() / b.abs().max()) <= 3e-5
    n = int(ei['n_proposals'].item())
    assert n > 100
    props = ei['proposals'][:n].clone()
    _, pe = run(False, props)
    _, ps = run(True, props)
    assert float((ps['boxes'] - pe['boxes']).abs().max()) <= 2e-3                     # pixels
    assert float((ps['scores'] - pe['scores']).abs().max()) <= 2e-6
    # end to end: the same detections (RPN top-k / NMS decisions included) up to near-ties
    assert abs(sb.shape[0] - eb.shape[0]) <= 2 and eb.shape[0] > 0
    d = (sb.double()[:, None, :] - eb.double()[None, :, :]).abs().amax(-1).min(dim=1).values
    assert float((d < 2e-3).double().mean()) >= 0.97


def test_cached_batch_pack_equals_single_packs_and_follows_the_weight(ops):
    """split_pack_cached (training): packs live on the weight tensor, follow its version counter, and the sweep after an in-place update re-packs every
    used weight in one wd_gemm_split_pack_batch launch - bit-identical to wd_gemm_split_pack_weight on the materialised (permuted / flipped /
    transposed) copy.  A new tensor at a recycled address never hits the old pack."""
    torch.manual_seed(5)
    lin = torch.randn(128, 192, device='cuda')
    conv = torch.randn(64, 128, 3, 3, device='cuda')
    one = torch.randn(128, 64, 1, 1, device='cuda')
    cl = torch.randn(64, 64, 3, 3, device='cuda').contiguous(memory_format=torch.channels_last)      # strides are honoured

    def reference(w, kind):
        if kind == 'T':
            return ops.split_pack_weight(w, transpose=True)
        if kind == 'dx':
            return ops.split_pack_weight(w.flip(2, 3).permute(1, 0, 2, 3).contiguous())
        return ops.split_pack_weight(w.contiguous())

    cases = [(lin, 'fwd'), (lin, 'T'), (conv, 'fwd'), (conv, 'dx'), (one, 'fwd'), (one, 'dx'), (cl, 'fwd'), (cl, 'dx')]
    for w, kind in cases:
        assert torch.equal(ops.split_pack_cached(w, kind), reference(w, kind)), kind
    first = [ops.split_pack_cached(w, kind) for w, kind in cases]
    assert all(a.data_ptr() == ops.split_pack_cached(w, kind).data_ptr() for a, (w, kind) in zip(first, cases))          # cached: the same buffer
    # in-place update of every weight (an optimizer step): the first request re-packs all of them in one launch
    for w in (lin, conv, one, cl):
        w.mul_(1.5).add_(0.01)
    before = len(ops._PACK_REGISTRY)
    got = ops.split_pack_cached(lin, 'fwd')
    assert torch.equal(got, reference(lin, 'fwd'))
    for w, kind in cases:
        e = w._wd_split_packs[kind]
        assert e.version == w._version and torch.equal(e.packed, reference(w, kind)), kind      # already fresh: packed by the sweep
    assert len(ops._PACK_REGISTRY) <= before
    # a used GEMM through the cached pack
    a = torch.randn(70, 192, device='cuda')
    y = ops.gemm_split(a, ops.split_pack_cached(lin), 128)
    assert torch.equal(y, ops.gemm_split(a, ops.split_pack_weight(lin), 128))
    assert float((y.double() - a.double() @ lin.double().t()).abs().max()) <= 1e-4
    # recycled address: a new weight of the same shape where a freed one lay must not see the old planes
    tmp = torch.randn(96, 128, device='cuda')
    ptr = tmp.data_ptr()
    ops.split_pack_cached(tmp)
    del tmp
    again = torch.randn(96, 128, device='cuda')
    if again.data_ptr() == ptr:
        assert torch.equal(ops.split_pack_cached(again), ops.split_pack_weight(again))
    with pytest.raises(ValueError):
        ops.split_pack_cached(torch.randn(64, 96, device='cuda'))                  # K % 64 != 0


# ---- round 6: activation planes (pre-split A pulled by LDS-DMA) ---------------------------------------------------------------------------

@pytest.mark.parametrize('m,n,k,epi', [(9600, 1024, 1024, 1), (9601, 1024, 1024, 1), (2400, 2048, 2048, 0), (38400, 512, 512, 1), (1000, 1024, 12544, 0),
                                       (100, 64, 128, 1), (4000, 224, 192, 1), (33, 32, 64, 0)])
def test_planes_kernel_is_bit_identical_to_the_f32_a_kernel(ops, m, n, k, epi):
    """wd_gemm_split_io with A as activation planes == wd_gemm_split_f32 bit for bit (the planes ARE the in-kernel split, the MFMA order is the same),
    with bias / residual / ReLU, ragged M, N % 256 != 0, K-sliced shapes; the planes output is the split of the f32 output; a planes residual
    (also in place: block output into the residual's buffer) gives the same result as the f32 residual."""
    torch.manual_seed(m + n + k)
    a = torch.randn(m, k, device='cuda')
    w = torch.randn(n, k, device='cuda') / k ** 0.5
    bias = torch.randn(n, device='cuda') if epi else None
    res = torch.randn(m, n, device='cuda') if epi else None
    pw = ops.split_pack_weight(w)
    ap = ops.split_planes_pack(a)
    assert torch.equal(ops.split_planes_unpack(ap, m, k), a)                      # exact both ways
    y0 = ops.gemm_split(a, pw, n, bias, res, bool(epi))
    y1, p1 = ops.gemm_split_io(m, n, k, pw, a_planes=ap, bias=bias, residual=res, relu=bool(epi), want_out=True, want_planes=True)
    assert torch.equal(y0, y1)
    assert torch.equal(ops.split_planes_unpack(p1, m, n), y1)
    full = ops.split_planes_bytes(m // 32 * 32, n) if m >= 32 else 0           # whole row blocks: byte-identical with a stand-alone pack
    assert torch.equal(p1[:full], ops.split_planes_pack(y1)[:full])
    y2, p2 = ops.gemm_split_io(m, n, k, pw, a=a, bias=bias, residual=res, relu=bool(epi), want_planes=True)      # f32-A kernel, both outputs
    assert torch.equal(y2, y0) and torch.equal(ops.split_planes_unpack(p2, m, n), y0)
    if epi:
        rp = ops.split_planes_pack(res)
        y3, _ = ops.gemm_split_io(m, n, k, pw, a_planes=ap, bias=bias, residual_planes=rp, relu=True)
        assert torch.equal(y3, y0)
        _, p4 = ops.gemm_split_io(m, n, k, pw, a_planes=ap, bias=bias, residual_planes=rp, relu=True, want_out=False, out_planes=rp)
        assert torch.equal(ops.split_planes_unpack(p4, m, n), y0)


def test_planes_chain_of_two_gemms_never_touches_f32(ops):
    """conv3 -> conv1 of the next bottleneck as the planes path runs them: block output as planes only, consumed as A AND as the residual."""
    torch.manual_seed(11)
    m, c = 4000, 256
    x = torch.relu(torch.randn(m, c, device='cuda'))
    w3, w1 = torch.randn(c, c, device='cuda') / 16, torch.randn(c, c, device='cuda') / 16
    b3 = torch.randn(c, device='cuda') * 0.1
    p3, p1 = ops.split_pack_weight(w3), ops.split_pack_weight(w1)
    xp = ops.split_planes_pack(x)
    _, blk = ops.gemm_split_io(m, c, c, p3, a_planes=xp, bias=b3, residual_planes=xp, relu=True, want_out=False, want_planes=True)
    y, _ = ops.gemm_split_io(m, c, c, p1, a_planes=blk, relu=True)
    blk_ref = ops.gemm_split(x, p3, c, b3, x.clone(), True)
    assert torch.equal(ops.split_planes_unpack(blk, m, c), blk_ref)
    assert torch.equal(y, ops.gemm_split(blk_ref, p1, c, None, None, True))


def test_split_gemm_error_with_operands_that_maximise_the_small_planes(ops):
    """the dropped cross terms (mid.lo, lo.mid, lo.lo) are largest when mid and lo sit at half an ulp of the plane above.  Operands
    x = 1 + 2^-9 + 2^-18 + ... patterns (every plane at its rounding boundary, random signs) must still keep the error against float64 within 1.25 x
    the f32 GEMM's."""
    g = torch.Generator().manual_seed(17)
    m, n, k = 2048, 256, 1024

    def worst(shape):
        sign = (torch.randint(0, 2, shape, generator=g) * 2 - 1).double()
        e = torch.randint(-3, 4, shape, generator=g).double()
        # hi = 1.xxxxxxx1 (odd last bit), mid just below half an ulp of hi, lo just below half an ulp of mid
        hi = 1.0 + torch.randint(0, 64, shape, generator=g).double() * 2.0 ** -6 + 2.0 ** -7
        v = hi + (2.0 ** -9 - 2.0 ** -16) + (2.0 ** -18 - 2.0 ** -23)
        return (sign * v * 2.0 ** e).float().cuda()

    a, w = worst((m, k)), worst((n, k)) / k ** 0.5
    ref = a.double() @ w.double().t()
    y = ops.gemm_split(a, ops.split_pack_weight(w), n)
    e_split, e_lib = _err(y, ref), _err(a @ w.t(), ref)
    assert e_split[1] <= 1.25 * e_lib[1] and e_split[0] <= 1.25 * e_lib[0], (e_split, e_lib)


def test_split_gemm_nonfinite_and_tiny_operands_follow_the_documented_semantics(ops):
    """include/waymodet.h: an inf / NaN operand makes its output elements NaN (an f32 GEMM gives +-inf where no NaN is involved); operands below
    2^-110 lose their lowest plane (relative error up to 2^-16 instead of 2^-24) - and nothing else in the matrix is disturbed."""
    torch.manual_seed(19)
    m, n, k = 160, 256, 64
    a = torch.randn(m, k, device='cuda')
    w = torch.randn(n, k, device='cuda')
    a[3, 5] = float('inf')
    a[7, 9] = float('nan')
    y = ops.gemm_split(a, ops.split_pack_weight(w), n)
    assert torch.isnan(y[3]).all() and torch.isnan(y[7]).all()
    clean = torch.ones(m, dtype=torch.bool, device='cuda')
    clean[3] = clean[7] = False
    ref = a[clean].double() @ w.double().t()
    assert float((y[clean].double() - ref).abs().max()) <= 1e-4
    tiny = torch.full((m, k), 2.0 ** -120, device='cuda') * (1 + 2.0 ** -20)
    one = torch.zeros(n, k, device='cuda')
    one[:, 0] = 1.0
    yt = ops.gemm_split(tiny, ops.split_pack_weight(one), n)
    rel = float(((yt[:, 0].double() - tiny[:, 0].double()) / tiny[:, 0].double()).abs().max())
    assert rel <= 2.0 ** -15 and torch.isfinite(yt).all()


def test_laboratory_knobs_are_ignored_without_wt_experiment():
    """WD_SPLIT_MT=2 re-enables a tile height next to which two other kernels return wrong results (profiles/r06_costream_victim_side.txt): a product
    process must ignore it (with a warning); WT_EXPERIMENT=1 is the explicit opt-in."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torch, time\n"
            "from waymo_2d_tracking_amd.detnet.nn import ops\n"
            "a = torch.randn(38400, 256, device='cuda'); w = ops.split_pack_weight(torch.randn(256, 256, device='cuda') / 16)\n"
            "ops.gemm_split(a, w, 256); torch.cuda.synchronize()\n"
            "e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)\n"
            "e0.record()\n"
            "for _ in range(20): ops.gemm_split(a, w, 256)\n"
            "e1.record(); torch.cuda.synchronize(); print('OK')\n")
    env = dict(os.environ, WD_SPLIT_MT='2')
    env.pop('WT_EXPERIMENT', None)
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'OK' in r.stdout, r.stderr[-2000:]
    assert 'WD_SPLIT_MT=2 ignored' in r.stderr
    env['WT_EXPERIMENT'] = '1'
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ignored' not in r.stderr, r.stderr[-2000:]


def test_planes_path_through_the_wide_stages_equals_the_f32_path(monkeypatch):
    """WD_SPLIT_PLANES (off by default, profiles/r06_split_presplit.txt): block outputs of res4 / res5 as activation planes.  The GEMMs are bit-identical
    either way; what remains between two runs of the backbone is the library offset convolution's atomics (not bit-repeatable on its own), so the
    feature maps are compared at that level."""
    from waymo_2d_tracking_amd.detnet.nn import cascade_rcnn
    torch.manual_seed(3)
    m = cascade_rcnn.CascadeRCNN(seed=2).cuda().eval()
    x = torch.randn(1, 3, 256, 384, device='cuda').contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        monkeypatch.setattr(cascade_rcnn, 'SPLIT_PLANES', False)
        ref = [f.clone() for f in m.backbone.bottom_up(x)]
        again = [f.clone() for f in m.backbone.bottom_up(x)]
        monkeypatch.setattr(cascade_rcnn, 'SPLIT_PLANES', True)
        got = [f.clone() for f in m.backbone.bottom_up(x)]
    assert len(got) == 4
    for r, a, g in zip(ref, again, got):
        assert g.shape == r.shape and torch.isfinite(g).all()
        floor = float((a - r).abs().max())
        assert float((g - r).abs().max()) <= max(4 * floor, 1e-5 * float(r.abs().max())), (float((g - r).abs().max()), floor)


def test_position_major_convolution_tiles_equal_the_plain_row_order(ops, tmp_path):
    """Round 6: 3x3 / stride 1 / pad 1 convolutions over many small maps (the box heads: 1000 ROIs x 7 x 7) run with position-major tiles that SKIP the taps in
    the zero padding (gemm_split_kernel<MT, 3>).  The walked taps are summed in the same order, so the result must be bit-identical to the plain row order
    (WD_SPLIT_NO_POSMAJOR=1, a laboratory switch, in a fresh process) - with bias / residual / ReLU, ragged map counts, non-square maps, N % 256 != 0 - and
    within the float64 gate."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = [(1000, 256, 7, 7, 256, 1), (300, 64, 5, 9, 32, 0), (257, 128, 3, 3, 64, 1), (513, 64, 9, 9, 288, 1)]
    code = ("import sys, torch\n"
            "from waymo_2d_tracking_amd.detnet.nn import ops\n"
            "out = {}\n"
            "for b, c, h, w, n, epi in %r:\n"
            "    torch.manual_seed(b + c + n)\n"
            "    x = torch.randn(b, c, h, w, device='cuda').contiguous(memory_format=torch.channels_last)\n"
            "    wt = torch.randn(n, c, 3, 3, device='cuda') / (3 * c ** 0.5)\n"
            "    bias = torch.randn(n, device='cuda') if epi else None\n"
            "    res = torch.randn(b, n, h, w, device='cuda').contiguous(memory_format=torch.channels_last) if epi else None\n"
            "    out[(b, c, h, w, n)] = ops.conv_split(x, ops.split_pack_weight(wt), n, 3, 1, 1, bias, res, bool(epi)).cpu()\n"
            "torch.save(out, sys.argv[1])\n" % (shapes,))
    env = dict(os.environ, WD_SPLIT_NO_POSMAJOR='1', WT_EXPERIMENT='1')
    ref_file = str(tmp_path / 'plain.pt')
    r = subprocess.run([sys.executable, '-c', code, ref_file], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = torch.load(ref_file)
    for b, c, h, w, n, epi in shapes:
        torch.manual_seed(b + c + n)
        x = torch.randn(b, c, h, w, device='cuda').contiguous(memory_format=torch.channels_last)
        wt = torch.randn(n, c, 3, 3, device='cuda') / (3 * c ** 0.5)
        bias = torch.randn(n, device='cuda') if epi else None
        res = torch.randn(b, n, h, w, device='cuda').contiguous(memory_format=torch.channels_last) if epi else None
        y = ops.conv_split(x, ops.split_pack_weight(wt), n, 3, 1, 1, bias, res, bool(epi))
        import ctypes as C
        from waymo_2d_tracking_amd import _lib
        if int(_lib.lib().wd_gemm_split_workspace(C.c_long(b * h * w), C.c_int(n), C.c_int(9 * c))) == 0:
            assert torch.equal(y.cpu(), ref[(b, c, h, w, n)]), (b, c, h, w, n)            # same taps, same order
        else:                                       # the plain row order cuts K into slices for this shape: another summation tree
            assert float((y.cpu() - ref[(b, c, h, w, n)]).abs().max()) <= 2e-5 * max(1.0, float(y.abs().max())), (b, c, h, w, n)
        y64 = F.conv2d(x.double(), wt.double(), None if bias is None else bias.double(), 1, 1)
        if epi:
            y64 = (y64 + res.double()).relu()
        assert float((y.double() - y64).abs().max()) <= 2e-5 * max(1.0, float(y64.abs().max()))
