"""Detector custom ops (HIP, through the C ABI) against the PyTorch restatement in oracle/detops_ref.py.
float32 kernels vs float64 references: tolerances stated per test (north_star: 1e-4 on boxes/scores)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.cuda().contiguous(memory_format=torch.channels_last)


def test_roi_pool_fpn_vs_reference():
    from oracle import detops_ref as R
    from waymo_2d_tracking_amd.detnet.nn import ops
    g = torch.Generator().manual_seed(0)
    strides = [4, 8, 16, 32]
    H, W, C = 128, 192, 72                       # C not a multiple of 64: exercises the channel tail
    feats = [torch.randn((2, C, H // s, W // s), generator=g) for s in strides]
    rois = []
    for size in (8, 20, 60, 120, 250, 420):
        for _ in range(6):
            w = size * float(torch.empty(1).uniform_(0.5, 2.0, generator=g))
            h = size * float(torch.empty(1).uniform_(0.5, 2.0, generator=g))
            x1 = float(torch.empty(1).uniform_(-20, W - 10, generator=g))
            y1 = float(torch.empty(1).uniform_(-20, H - 10, generator=g))
            rois.append([float(len(rois) % 2), x1, y1, x1 + w, y1 + h])
    rois.append([0.0, 10.0, 10.0, 10.0, 10.0])   # empty box
    rois.append([1.0, 500.0, 500.0, 600.0, 600.0])   # fully outside
    rois = torch.tensor(rois, dtype=torch.float32)
    exp, lv = R.roi_pool_fpn(feats, rois, [1.0 / s for s in strides])
    assert len(set(lv.tolist())) == 4            # all four levels exercised
    got = ops.roi_pool_fpn([_cl(f) for f in feats], rois.cuda(), [1.0 / s for s in strides])
    assert got.shape == exp.shape and got.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(got.cpu().double().numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('C,groups,stride,H,W,modulated', [(512, 32, 1, 20, 28, False), (512, 32, 2, 21, 30, False),
                                                          (1024, 32, 1, 12, 17, True), (2048, 32, 1, 9, 11, False),
                                                          (2048, 32, 2, 10, 12, False)])
def test_deform_conv_vs_reference(C, groups, stride, H, W, modulated):
    from oracle import detops_ref as R
    from waymo_2d_tracking_amd.detnet.nn import ops
    g = torch.Generator().manual_seed(C + stride)
    x = torch.randn((2, C, H, W), generator=g)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    offset = torch.randn((2, 18, Ho, Wo), generator=g) * 2.5          # samples leave the image at the borders
    weight = torch.randn((C, C // groups, 3, 3), generator=g) / (3 * (C // groups) ** 0.5)
    mask = torch.rand((2, 9, Ho, Wo), generator=g) if modulated else None
    scale = torch.rand(C, generator=g) + 0.5
    bias = torch.randn(C, generator=g)
    exp = R.deform_conv3x3(x, offset, weight, groups, stride, 1, mask)
    exp_affine = torch.relu(exp * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1))
    packed = ops.deform_pack_weight(weight.cuda(), groups)
    got = ops.deform_conv3x3(_cl(x), _cl(offset), packed, groups, stride, 1, mask=None if mask is None else _cl(mask))
    np.testing.assert_allclose(got.cpu().double().numpy(), exp.numpy(), rtol=1e-4, atol=1e-4)
    got2 = ops.deform_conv3x3(_cl(x), _cl(offset), packed, groups, stride, 1, scale=scale.cuda(), bias=bias.cuda(), relu=True,
                              mask=None if mask is None else _cl(mask))
    np.testing.assert_allclose(got2.cpu().double().numpy(), exp_affine.numpy(), rtol=1e-4, atol=1e-4)


def test_deform_conv_zero_offset_equals_grouped_conv():
    """With zero offsets the op is an ordinary grouped 3x3 convolution (torch CPU conv as an independent check)."""
    from waymo_2d_tracking_amd.detnet.nn import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn((1, 512, 16, 24), generator=g)
    weight = torch.randn((512, 16, 3, 3), generator=g) / 12
    exp = torch.nn.functional.conv2d(x.double(), weight.double(), None, 1, 1, 1, 32)
    got = ops.deform_conv3x3(_cl(x), _cl(torch.zeros(1, 18, 16, 24)), ops.deform_pack_weight(weight.cuda(), 32), 32)
    np.testing.assert_allclose(got.cpu().double().numpy(), exp.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('C,stride', [(256, 1), (512, 2), (1024, 1)])
def test_grouped_conv_without_offsets(C, stride):
    """offset=None: plain grouped 3x3 conv (res2 path, 8 channels per group at C=256) vs torch conv in float64."""
    from waymo_2d_tracking_amd.detnet.nn import ops
    g = torch.Generator().manual_seed(C)
    x = torch.randn((2, C, 19, 23), generator=g)
    weight = torch.randn((C, C // 32, 3, 3), generator=g) / (3 * (C // 32) ** 0.5)
    exp = torch.nn.functional.conv2d(x.double(), weight.double(), None, stride, 1, 1, 32)
    got = ops.deform_conv3x3(_cl(x), None, ops.deform_pack_weight(weight.cuda(), 32), 32, stride, 1)
    np.testing.assert_allclose(got.cpu().double().numpy(), exp.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('n', [1, 63, 64, 65, 500, 3000, 4741, 6144, 6500])     # <= 6144: column sweep, above: row sweep
def test_nms_vs_reference(n):
    from oracle import detops_ref as R
    from waymo_2d_tracking_amd.detnet.nn import ops
    g = torch.Generator().manual_seed(n)
    centers = torch.rand((max(1, n // 6), 2), generator=g) * 800
    c = centers[torch.randint(0, len(centers), (n,), generator=g)] + torch.randn((n, 2), generator=g) * 6
    wh = torch.rand((n, 2), generator=g) * 80 + 20
    boxes = torch.cat([c - wh / 2, c + wh / 2], dim=1)
    scores = torch.rand(n, generator=g)
    idxs = torch.randint(0, 3, (n,), generator=g, dtype=torch.int32)
    order = torch.argsort(scores, descending=True, stable=True)
    for use_idx in (False, True):
        exp = R.nms_sorted(boxes[order], idxs[order] if use_idx else None, 0.5)
        got = ops.batched_nms(boxes.cuda(), scores.cuda(), idxs.cuda() if use_idx else None, 0.5)
        assert got.cpu().tolist() == order[exp].tolist()


@pytest.mark.parametrize('M,N,K,relu', [(1000, 1024, 12544, True), (130, 70, 36, False), (2400, 2048, 1024, True),
                                        (64, 64, 4, False), (38400, 256, 512, False)])
def test_gemm_vs_reference(M, N, K, relu):
    from waymo_2d_tracking_amd.detnet.nn import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn((M, K), generator=g)
    bt = torch.randn((N, K), generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    res = torch.randn((M, N), generator=g)
    got = ops.gemm_nt(a.cuda(), bt.cuda(), bias.cuda(), res.cuda(), relu)
    exp = a.cuda().double() @ bt.cuda().double().t() + bias.cuda().double() + res.cuda().double()
    if relu:
        exp = torch.relu(exp)
    err = (got.double() - exp).abs().max().item()
    assert err < 2e-4, err
    got2 = ops.gemm_nt(a.cuda(), bt.cuda())
    exp2 = a.cuda().double() @ bt.cuda().double().t()
    assert (got2.double() - exp2).abs().max().item() < 2e-4


def test_groupnorm_relu_vs_torch():
    from waymo_2d_tracking_amd.detnet.nn import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn((37, 256, 7, 7), generator=g) * 3 + 1
    w = torch.rand(256, generator=g) + 0.5
    b = torch.randn(256, generator=g)
    exp = torch.relu(torch.nn.functional.group_norm(x.double(), 32, w.double(), b.double(), 1e-5))
    got = ops.groupnorm_relu_(_cl(x), w.cuda(), b.cuda(), 32)
    np.testing.assert_allclose(got.cpu().double().numpy(), exp.numpy(), rtol=1e-4, atol=1e-5)


def test_groupnorm_relu_autograd_function_vs_torch():
    """Training form of the box-head norm: forward into a second buffer + one HIP backward launch, against torch's group_norm + relu in
    float64 (dx, dgamma, dbeta), with and without the ReLU."""
    from waymo_2d_tracking_amd.detnet.nn import ops
    g = torch.Generator().manual_seed(11)
    for (r, c, groups, relu) in ((37, 256, 32, True), (5, 128, 32, False), (64, 256, 8, True)):
        x = torch.randn((r, c, 7, 7), generator=g) * 2 + 0.5
        w = torch.rand(c, generator=g) + 0.5
        b = torch.randn(c, generator=g) * 0.3
        gy = torch.randn((r, c, 7, 7), generator=g)
        xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
        yr = torch.nn.functional.group_norm(xr, groups, wr, br, 1e-5)
        yr = torch.relu(yr) if relu else yr
        yr.backward(gy.double())
        xg, wg, bg = _cl(x).requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
        yg = ops.GroupNormReluFn.apply(xg, wg, bg, groups, 1e-5, relu)
        yg.backward(_cl(gy))
        np.testing.assert_allclose(yg.detach().cpu().double().numpy(), yr.detach().numpy(), rtol=1e-4, atol=1e-5)
        for name, a, e in (('dx', xg.grad, xr.grad), ('dgamma', wg.grad, wr.grad), ('dbeta', bg.grad, br.grad)):
            err = (a.cpu().double() - e).abs().max().item() / (e.abs().max().item() + 1e-12)
            assert err < 1e-4, (name, r, c, groups, err)


@pytest.mark.parametrize('fold_epilogue', ['0', '1'])
def test_deform_conv_backward_vs_autograd_reference(fold_epilogue, monkeypatch):
    """dX, dOffset, dW of the HIP backward (im2col / GEMMs / col2im) vs torch autograd through the CPU restatement
    (grid_sample based) in float64."""
    from oracle import detector_ref as R
    from waymo_2d_tracking_amd.detnet.nn import ops
    monkeypatch.setenv('WD_FUSED_DEFORM_EPILOGUE', fold_epilogue)      # 1: the fused kernels mask / scale dY themselves (off by default)
    monkeypatch.setenv('WD_FUSED_DEFORM_S2', fold_epilogue)            # 1: the stride-2 case below takes the fused kernels too (off by default)
    g = torch.Generator().manual_seed(21)
    # stride 1: LDS-accumulating dx kernel (small offsets stay in the patch, the 4.0-scaled case also takes its global
    # path); stride 2: one global atomic per corner value
    for C, stride, H, W, osc in ((512, 1, 11, 13, 1.3), (1024, 2, 12, 10, 1.3), (512, 1, 19, 21, 0.4), (1024, 1, 9, 17, 4.0)):
        x = torch.randn((2, C, H, W), generator=g)
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        offset = torch.randn((2, 18, Ho, Wo), generator=g) * osc + 0.37        # keep samples off integer coordinates
        weight = torch.randn((C, C // 32, 3, 3), generator=g) / (3 * (C // 32) ** 0.5)
        gy = torch.randn((2, C, Ho, Wo), generator=g)
        xr, orf, wr = x.double().requires_grad_(), offset.double().requires_grad_(), weight.double().requires_grad_()
        yr = R.deform_conv3x3(xr, orf, wr, 32, stride, 1)
        yr.backward(gy.double())
        xg = _cl(x).requires_grad_(); og = _cl(offset).requires_grad_(); wg = weight.cuda().requires_grad_()
        yg = ops.DeformConvFn.apply(xg, og, wg, 32, stride, 1)
        yg.backward(_cl(gy))
        for name, a, b in (('dx', xg.grad, xr.grad), ('doffset', og.grad, orf.grad), ('dw', wg.grad, wr.grad)):
            a, b = a.cpu().double(), b
            err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
            assert err < 2e-4, (name, C, stride, err)
        # the block's fused epilogue y = relu(conv * scale + bias): its backward rides on the dY loads of the fused kernels (stride 1) or is one
        # masking pass in front of the column-slab form (stride 2)
        scale = torch.rand(C, generator=g) + 0.5
        bias = torch.randn(C, generator=g) * 0.2
        xr, orf, wr = x.double().requires_grad_(), offset.double().requires_grad_(), weight.double().requires_grad_()
        yr = torch.relu(R.deform_conv3x3(xr, orf, wr, 32, stride, 1) * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1))
        yr.backward(gy.double())
        xg = _cl(x).requires_grad_(); og = _cl(offset).requires_grad_(); wg = weight.cuda().requires_grad_()
        yg = ops.DeformConvFn.apply(xg, og, wg, 32, stride, 1, scale.cuda(), bias.cuda(), True)
        yg.backward(_cl(gy))
        assert (yg.detach().cpu().double() - yr.detach()).abs().max().item() < 2e-4 * yr.abs().max().item()
        for name, a, b in (('dx', xg.grad, xr.grad), ('doffset', og.grad, orf.grad), ('dw', wg.grad, wr.grad)):
            err = (a.cpu().double() - b).abs().max().item() / (b.abs().max().item() + 1e-12)
            assert err < 2e-4, ('fused epilogue', name, C, stride, err)


def test_roi_pool_backward_vs_autograd_reference():
    from oracle import detector_ref as R
    from waymo_2d_tracking_amd.detnet.nn import ops
    g = torch.Generator().manual_seed(5)
    strides = [4, 8, 16, 32]
    feats = [torch.randn((1, 64, 96 // s, 128 // s), generator=g) for s in strides]
    boxes = torch.tensor([[3.0, 5.0, 40.0, 33.0], [10.5, 2.25, 120.0, 90.0], [60.0, 40.0, 75.0, 58.0], [0.0, 0.0, 127.0, 95.0],
                          [20.0, 30.0, 90.0, 44.0]])
    gout = torch.randn((5, 64, 7, 7), generator=g)
    fr = [f.double().requires_grad_() for f in feats]
    out = R.roi_pool_fpn(fr, boxes, [1.0 / s for s in strides])
    out.backward(gout.double())
    fg = [_cl(f).requires_grad_() for f in feats]
    rois = torch.cat((torch.zeros(5, 1), boxes), 1).cuda()
    og = ops.RoiPoolFpnFn.apply(rois, [1.0 / s for s in strides], 7, 2, 4, 224.0, *fg)
    og.backward(_cl(gout))
    for l in range(4):
        a = fg[l].grad.cpu().double()
        b = fr[l].grad if fr[l].grad is not None else torch.zeros_like(a)      # level without ROIs
        assert (a - b).abs().max().item() < 1e-4 * (b.abs().max().item() + 1e-6) + 1e-6, l


# ---------------------------------------------------------------------------------------------------------------
# fused pre-processing kernel (rows a21 / a16): resize + flips + BGR + normalise + pad
from waymo_2d_tracking_amd.detnet.nn import ops  # noqa: E402


def _g7(golden_dir):
    g = np.load(os.path.join(golden_dir, 'tta_g7.npz'))
    return g, len([k for k in g.files if k.endswith('_spec')])


def _parse_spec(spec):
    scale, hf, vf = 1.0, False, False
    for a in spec.split(','):
        if a.startswith('x'):
            scale = float(a[1:])
        hf |= a == 'hflip'
        vf |= a == 'vflip'
    return scale, hf, vf


def test_preprocess_matches_reference_tta_fixture(golden_dir):
    """G7: wd_preprocess_f32 (no swap / normalisation) == the reference's TTA.pre_process output (F.interpolate
    bilinear align_corners=False, torch.flip), tolerance 1e-4 on 0..255 values; the padding is exactly zero."""
    g, n = _g7(golden_dir)
    for ci in range(n):
        spec = str(g['c%d_spec' % ci])
        scale, hf, vf = _parse_spec(spec)
        x = torch.from_numpy(g['c%d_x' % ci]).cuda()
        want = g['c%d_pre' % ci]
        for layout in ('f32', 'u8'):
            src = x if layout == 'f32' else x.permute(0, 2, 3, 1).contiguous().to(torch.uint8)
            out, (ho, wo) = ops.preprocess(src, scale, hf, vf, swap_rb=False, mean=None, std=None, divisor=32)
            assert (ho, wo) == want.shape[2:], spec
            assert out.shape[2] % 32 == 0 and out.shape[3] % 32 == 0 and out.is_contiguous(memory_format=torch.channels_last)
            got = out.cpu().numpy()
            np.testing.assert_allclose(got[:, :, :ho, :wo], want, rtol=0, atol=1e-4, err_msg='%s %s' % (spec, layout))
            assert not got[:, :, ho:, :].any() and not got[:, :, :, wo:].any()


def test_preprocess_normalise_swap_matches_model_preprocess():
    """Fused kernel == the torch restatement used by the training path (CascadeRCNN.preprocess after the RGB->BGR swap),
    bit for bit when there is no resize (same subtraction / division per element)."""
    from waymo_2d_tracking_amd.detnet.nn.cascade_rcnn import PIXEL_MEAN, PIXEL_STD
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 256, (2, 3, 70, 90), generator=g).float().cuda()
    out, (ho, wo) = ops.preprocess(x, 1.0, False, False, True, PIXEL_MEAN, PIXEL_STD, 32)
    bgr = x[:, [2, 1, 0]]
    mean = torch.tensor(PIXEL_MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(PIXEL_STD, device='cuda').view(1, 3, 1, 1)
    want = torch.nn.functional.pad((bgr - mean) / std, (0, 96 - 90, 0, 96 - 70))
    assert (ho, wo) == (70, 90) and out.shape == want.shape
    assert torch.equal(out, want)


def test_preprocess_full_size_properties():
    """1920x1280 (BASELINE.json frame size), --tta x1.5,hflip: size-independent properties.  (i) hflip of the output
    == output of the hflip-free call mirrored; (ii) a constant image stays constant under bilinear resize;
    (iii) u8 and f32 sources agree exactly; (iv) scale 1 without flips is a pure layout change."""
    g = torch.Generator().manual_seed(11)
    u8 = torch.randint(0, 256, (1, 1280, 1920, 3), generator=g, dtype=torch.uint8).cuda()
    f32 = u8.permute(0, 3, 1, 2).float().contiguous()
    a, (ho, wo) = ops.preprocess(f32, 1.5, True, False, False, None, None, 32)
    b, _ = ops.preprocess(f32, 1.5, False, False, False, None, None, 32)
    assert (ho, wo) == (1920, 2880) and a.shape == (1, 3, 1920, 2880)
    assert torch.equal(a, torch.flip(b, [3]))
    c, _ = ops.preprocess(u8, 1.5, True, False, False, None, None, 32)
    assert torch.equal(a, c)
    const = torch.full((1, 3, 1280, 1920), 77.0, device='cuda')
    d, _ = ops.preprocess(const, 1.5, True, True, False, None, None, 32)
    assert float((d - 77.0).abs().max()) <= 1e-4
    e, _ = ops.preprocess(u8, 1.0, False, False, False, None, None, 32)
    assert torch.equal(e, f32)
    ref = torch.flip(torch.nn.functional.interpolate(f32, scale_factor=1.5, mode='bilinear', align_corners=False), [3])
    assert float((a - ref).abs().max()) <= 1e-3          # torch's own GPU kernel, fp32 on 0..255 values


def test_preprocess_rejects_bad_arguments():
    x = torch.zeros((1, 3, 8, 8), device='cuda')
    with pytest.raises(RuntimeError):
        ops.preprocess(x, 0.0)
    with pytest.raises(RuntimeError):
        ops.preprocess(x, 1.0, divisor=6)


@pytest.mark.parametrize('C,H,W,stride', [(64, 13, 17, 1), (512, 40, 60, 1), (128, 16, 20, 2), (32, 1, 1, 1), (32, 5, 1, 1)])
def test_conv3x3_few_vs_conv2d(C, H, W, stride):
    """18-channel offset conv as GEMM + tap shift-add (wd_tap_shift_add_f32) == F.conv2d in float64, 1e-4 relative."""
    g = torch.Generator().manual_seed(C + H)
    x = _cl(torch.randn(2, C, H, W, generator=g))
    w = torch.randn(18, C, 3, 3, generator=g) * 0.05
    b = torch.randn(18, generator=g)
    got = ops.conv3x3_few(x, ops.tap_gemm_weight(w.cuda()), b.cuda(), 18, stride)
    want = torch.nn.functional.conv2d(x.cpu().double(), w.double(), b.double(), stride, 1)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    err = (got.cpu().double() - want).abs().max().item() / max(1.0, want.abs().max().item())
    assert err < 1e-4, err


@pytest.mark.parametrize('variant', ['lds', 'none', 'all'])
@pytest.mark.parametrize('C,off_std', [(512, 0.6), (1024, 0.6), (1024, 3.0)])
def test_deform_conv_every_kernel_variant(monkeypatch, variant, C, off_std):
    """The three stride-1 kernels behind wd_deform_conv3x3_f32 (LDS patch + register fragments [default], L1 gather,
    LDS patch + shared slab) against the float64 restatement: small offsets (all samples inside the patch: the pipelined
    fast path) and large ones (per-sample global fallback), modulated, tile edges that are not multiples of 8."""
    from oracle import detops_ref as R
    monkeypatch.setenv('WD_DEFORM_PATCH', variant)
    name = ops._lib.lib().wd_deform_conv3x3_variant
    name.restype = __import__('ctypes').c_char_p
    expect = {'lds': b'lds_kernel', 'none': b'deform_conv3x3_kernel', 'all': b'patch_kernel'}[variant]
    assert expect in name(C, 32, 1, 1, 1)
    g = torch.Generator().manual_seed(C + int(off_std * 10))
    H, W = 19, 27
    x = torch.randn((1, C, H, W), generator=g)
    offset = torch.randn((1, 18, H, W), generator=g) * off_std
    weight = torch.randn((C, C // 32, 3, 3), generator=g) / (3 * (C // 32) ** 0.5)
    mask = torch.rand((1, 9, H, W), generator=g)
    exp = R.deform_conv3x3(x, offset, weight, 32, 1, 1, mask)
    got = ops.deform_conv3x3(_cl(x), _cl(offset), ops.deform_pack_weight(weight.cuda(), 32), 32, 1, 1, mask=_cl(mask))
    np.testing.assert_allclose(got.cpu().double().numpy(), exp.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('H,W,batch,off_std', [(19, 27, 1, 0.6), (16, 24, 2, 0.0), (40, 33, 1, 3.0), (8, 8, 1, 1.0), (80, 120, 1, 0.6)])
def test_deform_conv_pingpong_kernel(monkeypatch, H, W, batch, off_std):
    """The two-team ping-pong kernel (det_deform_pp.hip; 32 channels per group, stride 1, no mask) against the float64
    restatement: partial tiles, several images, zero / small / large offsets (per-lane global fallback), odd tile counts
    per workgroup (dummy item of team 1), fused FrozenBN affine + ReLU."""
    from oracle import detops_ref as R
    monkeypatch.setenv('WD_DEFORM_PATCH', 'pp')
    C = 1024 if H < 80 else 256
    name = ops._lib.lib().wd_deform_conv3x3_variant
    name.restype = __import__('ctypes').c_char_p
    assert b'pp_kernel' in name(C, C // 32, 1, 1, 1)
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn((batch, C, H, W), generator=g)
    offset = torch.randn((batch, 18, H, W), generator=g) * off_std
    weight = torch.randn((C, 32, 3, 3), generator=g) / (3 * 32 ** 0.5)
    scale = torch.rand(C, generator=g) + 0.5
    bias = torch.randn(C, generator=g)
    exp = R.deform_conv3x3(x, offset, weight, C // 32, 1, 1, None)
    packed = ops.deform_pack_weight(weight.cuda(), C // 32)
    got = ops.deform_conv3x3(_cl(x), _cl(offset), packed, C // 32, 1, 1)
    np.testing.assert_allclose(got.cpu().double().numpy(), exp.numpy(), rtol=1e-4, atol=1e-4)
    got2 = ops.deform_conv3x3(_cl(x), _cl(offset), packed, C // 32, 1, 1, scale=scale.cuda(), bias=bias.cuda(), relu=True)
    exp2 = torch.relu(exp * scale.double()[None, :, None, None] + bias.double()[None, :, None, None])
    np.testing.assert_allclose(got2.cpu().double().numpy(), exp2.numpy(), rtol=1e-4, atol=1e-4)


def test_decode_boxes_is_bit_identical_to_the_torch_sequence():
    """wd_decode_boxes_f32 == apply_deltas + clip_boxes (detectron2 Box2BoxTransform / Boxes.clip restated in
    cascade_rcnn.py), bit for bit, with and without gather index / clipping; huge dw hits the scale clamp."""
    from waymo_2d_tracking_amd.detnet.nn.cascade_rcnn import apply_deltas, clip_boxes
    g = torch.Generator().manual_seed(4)
    m = 5000
    xy = torch.rand(m, 2, generator=g) * 1500
    boxes = torch.cat((xy, xy + torch.rand(m, 2, generator=g) * 400 + 1), 1).cuda()
    deltas = (torch.randn(m, 4, generator=g) * torch.tensor([3.0, 3.0, 40.0, 40.0])).cuda()
    idx = torch.randperm(m, generator=g)[:1777].cuda()
    for w in ((1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0), (30.0, 30.0, 15.0, 15.0)):
        want = apply_deltas(deltas, boxes, w)
        assert torch.equal(ops.decode_boxes(deltas, boxes, w), want)
        assert torch.equal(ops.decode_boxes(deltas, boxes, w, None, (1280, 1920)), clip_boxes(want, 1280, 1920))
        want_i = clip_boxes(apply_deltas(deltas[idx], boxes[idx], w), 886, 1280)
        assert torch.equal(ops.decode_boxes(deltas, boxes, w, idx, (886, 1280)), want_i)
    assert ops.decode_boxes(deltas[:0], boxes[:0], (1.0, 1.0, 1.0, 1.0)).shape == (0, 4)


def test_roi_pool_whole_image_rois_take_the_direct_kernel():
    """ROIs whose footprint exceeds 64 x 64 feature pixels on their level (whole-image boxes on p5) are flagged by the row
    kernel and finished by the direct kernel; mixed with ordinary ROIs in one call."""
    from oracle import detops_ref as R
    g = torch.Generator().manual_seed(2)
    strides = [4, 8, 16, 32]
    H = W = 2560
    C = 8
    feats = [torch.randn((1, C, H // s, W // s), generator=g) for s in strides]
    rois = torch.tensor([[0.0, 0.0, 0.0, 2560.0, 2560.0], [0.0, 100.0, 50.0, 2500.0, 2400.0], [0.0, 300.0, 300.0, 420.0, 380.0],
                         [0.0, 1000.0, 1200.0, 1900.0, 1800.0]], dtype=torch.float32)
    exp, lv = R.roi_pool_fpn(feats, rois, [1.0 / s for s in strides])
    got = ops.roi_pool_fpn([_cl(f) for f in feats], rois.cuda(), [1.0 / s for s in strides])
    np.testing.assert_allclose(got.cpu().double().numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)


def test_deform_far_offset_hint_selects_the_fallback_kernel_with_equal_results():
    """Per-layer calibration (cascade_rcnn.Bottleneck): offsets mostly inside the 2-px halo keep the persistent kernel, a wide
    offset field switches the layer to the per-tile fallback kernel; both kernels agree on the result either way."""
    from waymo_2d_tracking_amd.detnet.nn import ops as O
    g = torch.Generator().manual_seed(9)
    C, H, W = 1024, 40, 56
    x = torch.randn((1, C, H, W), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    packed = O.deform_pack_weight((torch.randn((C, 32, 3, 3), generator=g) * 0.05).cuda(), 32)
    for std, want_far in ((0.5, False), (3.0, True)):
        off = (torch.randn((1, 18, H, W), generator=g) * std).cuda().contiguous(memory_format=torch.channels_last)
        share = O.far_offset_share(off)
        assert (share > O.FAR_OFFSET_SHARE) == want_far, share
        a = O.deform_conv3x3(x, off, packed, 32, 1, 1, far_offsets=False)
        b = O.deform_conv3x3(x, off, packed, 32, 1, 1, far_offsets=True)
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-4)
    O.EVENT_LOG = []
    try:
        O.deform_conv3x3(x, off, packed, 32, 1, 1, far_offsets=True)
        O.deform_conv3x3(x, off, packed, 32, 1, 1, far_offsets=False)
        names = [t[0].split(':')[0] for t in O.EVENT_LOG]
    finally:
        O.EVENT_LOG = None
    assert names == ['deform_conv3x3_lds_kernel<32>', 'deform_conv3x3_pp_kernel<32>'], names


@pytest.mark.parametrize('H,W,batch,off_std', [(80, 120, 1, 0.7), (19, 23, 2, 2.5), (8, 8, 1, 0.0)])
def test_deform_table_prepass_equals_in_kernel_table(H, W, batch, off_std):
    """wd_deform_offsets_table_f32 (offset conv gather + sampling table in one launch) gives the offsets of wd_tap_shift_add_f32
    bit for bit, and the persistent kernel fed with that table the output it computes with its own in-kernel table."""
    from waymo_2d_tracking_amd.detnet.nn import ops as O
    g = torch.Generator().manual_seed(H * W + batch)
    C = 1024
    x = torch.randn((batch, C, H, W), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    packed = O.deform_pack_weight((torch.randn((C, 32, 3, 3), generator=g) * 0.05).cuda(), 32)
    w_off = (torch.randn((18, C, 3, 3), generator=g) * (off_std / 96.0)).cuda()
    b_off = (torch.randn(18, generator=g) * 0.1 * off_std).cuda()
    w2 = O.tap_gemm_weight(w_off)
    off_a = O.conv3x3_few(x, w2, b_off, 18, 1)
    off_b, table = O.conv3x3_few(x, w2, b_off, 18, 1, deform_table=True)
    assert torch.equal(off_a, off_b)
    ya = O.deform_conv3x3(x, off_a, packed, 32, 1, 1)
    yb = O.deform_conv3x3(x, off_b, packed, 32, 1, 1, table=table)
    assert torch.equal(ya, yb)


def test_bf16x3_experiment_mode_stays_close_to_fp32(tmp_path):
    """WD_DEFORM_BF16X3=1 (exploratory, never benchmarked: 2-way bfloat16 split of weights and samples on the bf16 matrix pipe) in a
    fresh process (the switch is read once): the result stays within 1e-4 of the float64 restatement relative to the output's
    scale - the mode is a measured experiment, and this keeps it from rotting."""
    import subprocess
    import sys
    code = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
from oracle import detops_ref as R
from waymo_2d_tracking_amd.detnet.nn import ops
g = torch.Generator().manual_seed(5)
C, H, W = 128, 19, 26
x = torch.randn((1, C, H, W), generator=g)
off = torch.randn((1, 18, H, W), generator=g) * 1.5
w = torch.randn((C, 32, 3, 3), generator=g) / (3 * 32 ** 0.5)
exp = R.deform_conv3x3(x, off, w, 4, 1, 1)
cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)
got = ops.deform_conv3x3(cl(x), cl(off), ops.deform_pack_weight(w.cuda(), 4), 4, 1, 1).cpu().double()
err = float((got - exp).abs().max() / exp.pow(2).mean().sqrt())
print('ERR', err)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(**env):
        p = subprocess.run([sys.executable, '-c', code], env=dict({k: v for k, v in os.environ.items() if k != 'WT_EXPERIMENT'}, **env), capture_output=True,
                           text=True, timeout=600)
        assert p.returncode == 0, (p.stdout[-500:], p.stderr[-2000:])
        return float(p.stdout.split('ERR')[1].split()[0]), p.stderr

    err_x, _ = run(WD_DEFORM_BF16X3='1', WT_EXPERIMENT='1')
    assert err_x < 1e-4, err_x
    # a laboratory switch (results are not fp32; round 6: also the strongest co-residency aggressor measured): without WT_EXPERIMENT=1 it is ignored, loudly
    err, log = run(WD_DEFORM_BF16X3='1')
    assert 'WD_DEFORM_BF16X3=1 ignored' in log, log[-500:]
    assert err < 6e-6 and err_x > 3 * err, (err, err_x)       # ... and the first run really was the split path (the f32 kernel sits at ~3e-6 of the output's scale)


def test_upsample2x_nearest_equals_interpolate():
    """FPN top-down pathway: wd_upsample2x_nhwc_f32 == F.interpolate(scale_factor=2, mode='nearest'), bit for bit, at the three
    pyramid sizes and an odd one"""
    import torch.nn.functional as F
    from waymo_2d_tracking_amd.detnet.nn import ops
    g = torch.Generator().manual_seed(0)
    for (n, c, h, w) in ((1, 256, 40, 60), (1, 256, 80, 120), (1, 256, 160, 240), (2, 12, 5, 7)):
        x = torch.randn((n, c, h, w), generator=g).cuda().contiguous(memory_format=torch.channels_last)
        got = ops.upsample2x_nearest(x)
        exp = F.interpolate(x, scale_factor=2.0, mode='nearest')
        assert got.shape == exp.shape and got.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(got, exp)


def test_autocontrast_kernel_equals_pil_for_every_range_and_on_images():
    """wd_autocontrast_u8 (README.md:37 --auto-contrast=1): every one of the 32 640 (darkest, brightest) ranges gives PIL's table
    (Python float formula), and whole images - odd sizes, flat channels, JPEG-decoded content - equal ImageOps.autocontrast."""
    import io
    from PIL import Image, ImageOps
    from waymo_2d_tracking_amd.detnet.nn import ops
    v = np.arange(256)
    bad = []
    for lo in range(0, 255, 1):
        his = np.arange(lo + 1, 256)
        for k in range(0, len(his), 3):
            hs = [int(his[min(k + c, len(his) - 1)]) for c in range(3)]
            img = np.stack([np.clip(v, lo, h) for h in hs], -1).astype(np.uint8).reshape(1, 256, 3)
            got = ops.autocontrast_(torch.from_numpy(img.copy()).cuda()).cpu().numpy()[0]
            for c, h in enumerate(hs):
                scale = 255.0 / (h - lo)
                offset = -lo * scale
                exp = np.array([min(255, max(0, int(x * scale + offset))) for x in np.clip(v, lo, h)])
                if not np.array_equal(got[:, c], exp):
                    bad.append((lo, h))
    assert not bad, bad[:10]
    rng = np.random.default_rng(2)
    for (h, w, lo, hi) in ((96, 160, 30, 200), (7, 5, 0, 256), (1, 1, 10, 11), (33, 47, 100, 101), (64, 64, 5, 250)):
        arr = rng.integers(lo, hi, (h, w, 3), dtype=np.uint8)
        arr[..., 1] = 77 if h == 64 else arr[..., 1]                       # a flat channel stays as it is
        got = ops.autocontrast_(torch.from_numpy(arr.copy()).cuda()).cpu().numpy()
        assert np.array_equal(got, np.asarray(ImageOps.autocontrast(Image.fromarray(arr)))), (h, w)
    buf = io.BytesIO()
    Image.fromarray(rng.integers(30, 200, (96, 160, 3), dtype=np.uint8)).save(buf, 'JPEG', quality=92)
    img = Image.open(io.BytesIO(buf.getvalue())).convert('RGB')
    got = ops.autocontrast_(torch.from_numpy(np.array(img)).cuda()).cpu().numpy()
    assert np.array_equal(got, np.asarray(ImageOps.autocontrast(img)))


def test_roi_pool_workgroup_kernel_equals_row_kernel_and_ignores_the_processing_order(monkeypatch):
    """Round 4: one workgroup per ROI (weight tables once, sliding 3-bin fold, bucket-ordered processing) against the round-2 kernel
    (one wave per bin row, dense fold) on the roofline tool's ROI distribution plus tiny / border / degenerate boxes: same sums in a
    different association -> 1e-5 relative; the processing order changes nothing at all (bit-equal)."""
    from waymo_2d_tracking_amd.detnet.nn import ops
    g = torch.Generator().manual_seed(5)
    strides = [4, 8, 16, 32]
    H, W, C = 320, 480, 256
    feats = [_cl(torch.randn((1, C, H // s, W // s), generator=g)) for s in strides]
    n = 1500
    size = torch.exp(torch.empty(n).uniform_(0.5, 6.2, generator=g))          # 1.6 .. 490 px: bins far below a pixel up to whole-image boxes
    ar = torch.exp(torch.empty(n).uniform_(-1.0, 1.0, generator=g))
    w, h = size * ar.sqrt(), size / ar.sqrt()
    cx = torch.empty(n).uniform_(-30, W + 30, generator=g)
    cy = torch.empty(n).uniform_(-30, H + 30, generator=g)
    rois = torch.stack([torch.zeros(n), cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    rois[7] = torch.tensor([0.0, -50.0, -50.0, 900.0, 700.0])                 # beyond the 64 x 64 tables
    rois[8] = torch.tensor([0.0, 100.0, 100.0, 100.0, 100.0])                 # empty
    rois[9] = torch.tensor([3.0, 10.0, 10.0, 50.0, 50.0])                     # bad batch index -> zeros
    rois = rois.cuda()
    sc = [1.0 / s for s in strides]
    monkeypatch.setenv('WD_ROI_KERNEL', 'row')
    ref = ops.roi_pool_fpn(feats, rois, sc).clone()
    monkeypatch.setenv('WD_ROI_KERNEL', 'wg')
    monkeypatch.setenv('WD_ROI_ORDER', '0')
    plain = ops.roi_pool_fpn(feats, rois, sc).clone()
    monkeypatch.setenv('WD_ROI_ORDER', '1')
    ordered = ops.roi_pool_fpn(feats, rois, sc).clone()
    assert torch.equal(plain, ordered)
    assert float(plain[9].abs().max()) == 0.0
    err = (plain - ref).abs().max() / ref.abs().max()
    assert float(err) <= 1e-5, float(err)
    # fewer ROIs than the ordering kernel bothers with, and more than one workgroup round
    for m in (1, 63, 64, 1500):
        a = ops.roi_pool_fpn(feats, rois[:m].contiguous(), sc)
        assert torch.equal(a, plain[:m])
