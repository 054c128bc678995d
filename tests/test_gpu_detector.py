"""Whole-graph parity of the HIP-backed Cascade R-CNN X152-FPN against the CPU restatement oracle/detector_ref.py
(same random-init parameters, small image).  float32 on both sides; tolerances are relative to the tensor's scale."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def models():
    from waymo_2d_tracking_amd.detnet.nn.cascade_rcnn import CascadeRCNN
    cpu = CascadeRCNN(seed=3, offset_std=0.02).eval()
    gpu = CascadeRCNN(seed=3, offset_std=0.02).eval().cuda()
    return cpu, gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def test_backbone_heads_and_detections_match_reference(models):
    from oracle import detector_ref as R
    cpu, gpu = models
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (1, 3, 160, 224), generator=g).float()
    (rb, rs, rc), ref = R.forward(cpu, img, return_intermediates=True)
    inter = {}
    gb, gs, gc = gpu(img.cuda(), intermediates=inter)
    # FPN features p2..p6
    for lvl, (a, b) in enumerate(zip(inter['feats'], ref['feats'])):
        assert a.shape == b.shape
        assert _rel(a, b) < 2e-4, (lvl, _rel(a, b))
    # same proposals into both cascades: logits / deltas of every stage
    inter2 = {}
    gpu(img.cuda(), proposals=ref['proposals'].cuda(), intermediates=inter2)
    (_, _, _), ref2 = R.forward(cpu, img, return_intermediates=True, proposals=ref['proposals'])
    for k in range(3):
        assert _rel(inter2['stage_out'][k][0], ref2['stage_out'][k][0]) < 1e-3, k
        assert _rel(inter2['stage_out'][k][1], ref2['stage_out'][k][1]) < 1e-3, k
    np.testing.assert_allclose(inter2['boxes'].cpu().numpy(), ref2['boxes'].numpy(), rtol=0, atol=5e-3)   # pixels
    np.testing.assert_allclose(inter2['scores'].cpu().numpy(), ref2['scores'].numpy(), rtol=0, atol=1e-4)
    # end to end (RPN + NMS decisions included): the same number of detections, matching boxes
    assert gb.shape[0] == rb.shape[0] <= 100
    n_prop = int(inter['n_proposals'].item())                # static-shape proposal list: zero rows behind the real ones
    assert n_prop == ref['proposals'].shape[0] and inter['proposals'].shape[0] == gpu.rpn.post
    assert float(inter['proposals'][n_prop:].abs().sum()) == 0.0
    d = torch.cdist(gb.cpu().double(), rb.double(), p=1).min(dim=1).values
    assert (d < 0.05).float().mean().item() > 0.95


def test_predict_contract(models):
    """Detectron2Det.predict: [per class] (n,5) float32 [score, cx, cy, w, h] normalised (detectron2_det/__init__.py:119-135)."""
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
    m = Detectron2Det(seed=1).eval().cuda()
    x = torch.randint(0, 256, (2, 3, 128, 160)).float()
    out = m.predict(x)
    assert len(out) == 2 and all(len(o) == 4 for o in out)
    for per_class in out:
        for arr in per_class:
            assert arr.ndim == 2 and arr.shape[1] == 5
            if len(arr):
                assert arr.dtype == np.float32
                assert (arr[:, 1:3] >= 0).all() and (arr[:, 1:3] <= 1).all()
        assert sum(len(a) for a in per_class) <= 100


def test_tta_x15_hflip_roundtrip(models):
    """--tta x1.5,hflip (nn/tta.py:228-267): one pass on the enlarged, flipped image; cx is mirrored back."""
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
    from waymo_2d_tracking_amd.detnet.nn.tta import TTA
    from waymo_2d_tracking_amd.detnet.nn import ops
    m = Detectron2Det(seed=1).eval().cuda()
    x = torch.randint(0, 256, (1, 3, 96, 128)).float().cuda()
    direct = m.predict(x, 1.5, True, False)                 # fused pre-processing: resize x1.5 + hflip inside the kernel
    assert m.last_input_size == (144, 192)
    via = TTA(m, ['x1.5', 'hflip']).predict(x)
    # the two passes run the same kernels; the split-K FC (f32 atomics) and the library GEMMs are not run-to-run
    # deterministic, so rows are matched by nearest neighbour instead of by position
    n = matched = 0
    for a, b in zip(direct[0], via[0]):
        assert abs(len(a) - len(b)) <= 1
        n += len(a)
        if len(a) and len(b):
            am = a.copy()
            am[:, 1] = 1 - am[:, 1]                           # HFlipTTA.post_process: cx <- 1 - cx (tta.py:150-155)
            d = np.abs(am[:, None, :] - b[None, :, :]).sum(-1).min(1)
            matched += int((d < 1e-3).sum())
    assert n > 0 and matched >= 0.95 * n, (matched, n)
    # the un-fused route (torch resize + flip, then the detector) sees the same image up to fp32 rounding of the
    # bilinear weights (tests/test_gpu_detops.py pins the fused kernel to the reference's TTA.pre_process output)
    big = torch.flip(torch.nn.functional.interpolate(x, scale_factor=1.5, mode='bilinear', align_corners=False), [3])
    xn, _ = ops.preprocess(x, 1.5, True, False, False, None, None, 32)
    assert float((xn[:, :, :144, :192] - big).abs().max()) <= 1e-3
    unfused = m.predict(big)
    assert abs(sum(len(a) for a in unfused[0]) - n) <= max(2, n // 10)


def _flipped_blocks(names, dec_a, dec_b):
    """Blocks whose DISCRETE decisions differ between two executions (oracle.detector_ref.block_decisions: the three ReLU masks and the bilinear cell of every
    (pixel, tap, axis) of the deformable sampling).  A gradient is a piecewise-smooth function of the forward values: where the decisions agree, two
    executions differ by rounding; a decision that fell the other way (a ReLU input / a sampling position within float32 noise of its boundary) moves the
    tensors of ITS block by a finite step and, through the gradient that flows on, the blocks in front of it by a smaller one."""
    flipped = []
    for name, a, b in zip(names, dec_a, dec_b):
        n = sum(int((a[k] != b[k]).sum()) for k in ('relu1', 'relu2', 'relu3'))
        if a['cells'] is not None:
            n += int((a['cells'] != b['cells']).sum())
        if n:
            flipped.append(name)
    return flipped


def _check_gradient_tiers(rel, names, flipped, tight, max_flipped=3):
    """Evidence-based bounds (round 6; measured with tools/train_flip_check.py, gpurun_out/r06_flip_check.txt): tensors of a block with a demonstrated flip
    <= 3e-2 (seen 8.9e-3), tensors of blocks in FRONT of a flipped block (their incoming gradient passed through it) <= 1e-2 (seen 6e-3 - 8e-3 in round
    5), EVERY other tensor <= `tight` (seen without a flip: 2.3e-3 against float64); at most `max_flipped` flipped blocks (float32 against float64: res4.13 in every run, res4.27 in
    about half; two float32 graphs whose epilogues round differently - fused multiply-add against multiply, then add - disagree in more places: 2 - 5
    blocks seen)."""
    assert len(flipped) <= max_flipped, flipped
    order = {n: i for i, n in enumerate(names)}
    last = max([order[f] for f in flipped], default=-1)
    for n, e in rel.items():
        blk = next((b for b in names if n.startswith('backbone.' + b + '.')), None)
        if blk in flipped:
            bound = 3e-2
        elif blk is not None and order[blk] < last:
            bound = 1e-2
        else:
            bound = tight
        assert e <= bound, (n, e, bound, flipped)
    ordered = sorted(rel.values())
    assert ordered[len(ordered) // 2] <= 1e-3, ordered[len(ordered) // 2]


def _block_names(model):
    import re
    return [n for n, _ in model.backbone.named_modules() if re.fullmatch(r'res\d\.\d+', n)]


def test_training_losses_and_gradients_vs_f64_restatement():
    """Row a23 / config 5 parity: the 8 detectron2 losses and parameter gradients of one training step, HIP-backed graph
    (float32; DeformConv / ROIAlign forward + backward kernels) vs the float64 CPU restatement with autograd
    (oracle/detector_ref.losses - written independently of detnet/nn/training.py).  detectron2's random fg / bg subsampling is
    replaced on BOTH sides by "lowest indices" (training.first_choice), so the sampled anchors / proposals are comparable.
    Tolerances: losses 1e-4 relative (north_star); gradients: every trainable tensor within 3e-3 of its largest entry, the median tensor within 1e-3 -
    EXCEPT the tensors of blocks where a discrete decision of the float32 forward DEMONSTRABLY fell the other way than in float64 (both sides export
    their ReLU masks and bilinear cells; _flipped_blocks compares them exactly) and of the blocks in front of such a block: _check_gradient_tiers.
    Why flips exist (measured, rounds 5 / 6: tools/train_flip_check.py): the float32 forward is not run-to-run identical (library split-K kernels add
    with atomics: every res4 activation moves by ~7e-7 relative); a ReLU input of res4.13's deformable conv sits within that noise of zero in every
    run (its block: 3.7e-3), one of res4.27 in about half of the runs (8.9e-3; 2.3e-3 without it).  On the 10 x 14 res4 map of this test one pixel
    is 1 / 140 of every sum."""
    import copy
    from oracle import detector_ref as R
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
    from waymo_2d_tracking_amd.detnet.nn import training
    m = Detectron2Det(seed=4).cuda().train()
    training.set_trainable(m.model)
    cpu = copy.deepcopy(m.model).cpu()
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (1, 3, 160, 224), generator=g).float()          # BGR 0..255
    gt = torch.tensor([[20., 30., 120., 150.], [100., 40., 215., 155.], [5., 5., 60., 60.], [130., 8., 200., 70.]])
    cls = torch.tensor([0, 1, 3, 0])
    cfg = dict(pre_nms=300, post_nms=200, rpn_batch=64, rpn_pos=0.5, roi_batch=128, roi_pos=0.25)
    # reference first: its proposals feed both sides (an NMS near-tie between float32 and float64 objectness would otherwise
    # change WHICH proposals are sampled and turn a rounding difference into a different experiment)
    ref, inter = R.losses(cpu, img, gt, cls, torch.float64, cfg['rpn_batch'], cfg['rpn_pos'], cfg['pre_nms'], cfg['post_nms'],
                          cfg['roi_batch'], cfg['roi_pos'], return_intermediates=True)
    sum(ref.values()).backward()
    from waymo_2d_tracking_amd.detnet.nn import cascade_rcnn
    block_names = _block_names(m.model)
    cascade_rcnn.DECISION_LOG = log = []
    try:
        got = training.losses(m.model, img.cuda(), gt.cuda(), cls.cuda(), choose=training.first_choice, config=cfg,
                              proposals=inter['proposals'].float().cuda())
    finally:
        cascade_rcnn.DECISION_LOG = None
    assert len(log) == len(block_names) == len(inter['blocks'])
    assert set(got) == set(ref) and len(ref) == 8
    for k in sorted(ref):
        assert abs(float(got[k]) - float(ref[k])) <= 1e-4 * max(1.0, abs(float(ref[k]))), (k, float(got[k]), float(ref[k]))
    sum(got.values()).backward()
    # the HIP path's own RPN proposals agree with the restatement's up to near-ties
    own = training.losses(m.model, img.cuda(), gt.cuda(), cls.cuda(), choose=training.first_choice, config=cfg)
    assert abs(float(own['loss_rpn_cls']) - float(ref['loss_rpn_cls'])) <= 1e-4 * max(1.0, float(ref['loss_rpn_cls']))
    names = ['backbone.res4.5.conv2_weight', 'backbone.res4.5.conv2_offset.weight', 'backbone.res3.0.conv2_weight',
             'backbone.res5.2.conv2_offset.bias', 'backbone.res4.20.conv1.weight', 'backbone.res3.0.shortcut.weight',
             'backbone.lateral.1.weight', 'backbone.output.0.weight', 'rpn.conv.weight', 'rpn.objectness.weight', 'rpn.deltas.bias',
             'heads.0.fc1_weight', 'heads.1.convs.0.weight', 'heads.2.norms.3.weight', 'heads.2.box_weight', 'heads.0.cls_bias']
    gp, rp = dict(m.model.named_parameters()), dict(cpu.named_parameters())
    import re
    rel = {}
    for n, p in gp.items():
        if p.requires_grad and p.grad is not None and float(rp[n].grad.abs().max()) > 0:
            rel[n] = float((p.grad.double().cpu() - rp[n].grad.double()).abs().max()) / float(rp[n].grad.abs().max())
    for n in names:
        assert gp[n].grad is not None and rp[n].grad is not None and n in rel, n
    flipped = _flipped_blocks(block_names, [R.block_decisions(*t) for t in log], inter['blocks'])
    _check_gradient_tiers(rel, block_names, flipped, tight=3e-3)
    assert len(rel) > 300
    # every trainable tensor: gradient direction agrees (cosine) - catches a wrong layout / missing term anywhere
    for n, p in gp.items():
        if p.requires_grad:
            a, b = p.grad.double().cpu().flatten(), rp[n].grad.double().flatten()
            if float(b.norm()) == 0.0:                   # e.g. the p5 / p6 convs: "lowest indices" samples only p2 anchors and no
                assert float(a.norm()) == 0.0, n         # ROI of this small image is pooled from p5 - zero on both sides
                continue
            cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
            assert cos > 0.9999, (n, cos)


def test_fused_training_epilogues_equal_the_plain_autograd_graph(monkeypatch):
    """Round 4: FrozenBN affine / bias / residual / ReLU fused into the producing op for training (ops.LinearActFn, the epilogue of the
    deformable kernel + wd_act_bwd_f32) against the round-3 graph of separate autograd-visible elementwise passes: same losses, same
    gradients up to the float atomics of the scatter kernels."""
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
    from waymo_2d_tracking_amd.detnet.nn import training, cascade_rcnn
    m = Detectron2Det(seed=4).cuda().train()
    training.set_trainable(m.model)
    g = torch.Generator().manual_seed(12)
    img = torch.randint(0, 256, (1, 3, 160, 224), generator=g).float().cuda()
    gt = torch.tensor([[20., 30., 120., 150.], [100., 40., 215., 155.], [5., 5., 60., 60.]]).cuda()
    cls = torch.tensor([0, 1, 3]).cuda()
    cfg = dict(pre_nms=300, post_nms=200, rpn_batch=64, rpn_pos=0.5, roi_batch=128, roi_pos=0.25)
    from oracle import detector_ref as R
    block_names = _block_names(m.model)
    out = {}
    for fused in (False, True):
        monkeypatch.setattr(cascade_rcnn, 'FUSED_TRAINING_EPILOGUES', fused)
        for p in m.model.parameters():
            p.grad = None
        cascade_rcnn.DECISION_LOG = log = []
        try:
            losses = training.losses(m.model, img, gt, cls, choose=training.first_choice, config=cfg)
        finally:
            cascade_rcnn.DECISION_LOG = None
        sum(losses.values()).backward()
        out[fused] = ({k: float(v) for k, v in losses.items()},
                      {n: p.grad.clone() for n, p in m.model.named_parameters() if p.requires_grad and p.grad is not None},
                      [R.block_decisions(*t) for t in log])
    (l0, g0, d0), (l1, g1, d1) = out[False], out[True]
    assert set(l0) == set(l1) and set(g0) == set(g1) and len(g0) > 300
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 2e-5 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    # gradients: float atomics move every tensor by up to ~2.5e-3 of its largest entry from run to run (tools/train_grad_check.py, "vs run 0"); blocks
    # whose discrete decisions differ between the two executions (exported and compared exactly) and the blocks in front of them get the wider tiers
    rel = {n: float((g0[n] - g1[n]).abs().max() / (g0[n].abs().max() + 1e-30)) for n in g0}
    _check_gradient_tiers(rel, block_names, _flipped_blocks(block_names, d0, d1), tight=4e-3, max_flipped=8)


def test_training_step_full_size_properties():
    """Config 5 at its stated size (886x1280 crop, train.py:37-47): finite losses, every trainable tensor receives a non-zero
    finite gradient, and two executions of the same step agree (up to the library kernels' atomics: 1e-3 relative on the
    loss)."""
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
    from waymo_2d_tracking_amd.detnet.nn import training
    m = Detectron2Det(seed=0).cuda().train()
    params = training.set_trainable(m.model)
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (1, 3, 886, 1280), generator=g).float().cuda()
    wh = torch.rand((30, 2), generator=g) * 280 + 20
    xy = torch.rand((30, 2), generator=g) * torch.tensor([1280 - 300.0, 886 - 300.0])
    boxes = torch.cat((xy, xy + wh), 1).cuda()
    classes = torch.randint(0, 4, (30,), generator=g).cuda()
    vals = []
    for rep in range(2):
        for p in params:
            p.grad = None
        out = training.losses(m.model, img, boxes, classes, choose=training.first_choice)
        assert len(out) == 8 and all(torch.isfinite(v) for v in out.values())
        total = sum(out.values())
        total.backward()
        vals.append(float(total))
        if rep == 0:
            zero = [n for n, p in m.model.named_parameters() if p.requires_grad and (p.grad is None or not torch.isfinite(p.grad).all()
                                                                                     or float(p.grad.abs().sum()) == 0.0)]
            assert not zero, zero[:5]
    assert abs(vals[0] - vals[1]) <= 1e-3 * abs(vals[0]), vals


def test_training_step_losses_and_gradients():
    """Config 5 (fwd+bwd): Detectron2Det.loss returns the detectron2 loss dict; backward reaches every trainable
    parameter (through the HIP deformable-conv / ROIAlign backward kernels); an SGD step changes the loss."""
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
    from waymo_2d_tracking_amd.detnet.nn import training
    torch.manual_seed(0)
    m = Detectron2Det(seed=2).cuda().train()
    params = training.set_trainable(m.model)
    x = torch.randint(0, 256, (1, 3, 192, 256)).float().cuda()
    target = {'boxes': [torch.tensor([[20., 30., 120., 150.], [100., 40., 230., 170.], [5., 5., 60., 60.]])],
              'labels': [torch.tensor([1, 2, 4])]}
    losses = m.loss(x, target)
    assert set(losses) == {'loss_rpn_cls', 'loss_rpn_loc', 'loss_cls_stage0', 'loss_box_reg_stage0', 'loss_cls_stage1',
                           'loss_box_reg_stage1', 'loss_cls_stage2', 'loss_box_reg_stage2'}
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    missing = [n for n, p in m.model.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing[:5]
    assert all(torch.isfinite(p.grad).all() for p in params)
    blk = m.model.backbone.res4[5]
    assert blk.conv2_weight.grad.abs().sum() > 0 and blk.conv2_offset.weight.grad.abs().sum() > 0
    assert m.model.backbone.res2[0].conv1.weight.grad is None            # frozen (FREEZE_AT 2)
    opt = torch.optim.SGD(params, lr=1e-4, momentum=0.9, weight_decay=1e-4)
    torch.nn.utils.clip_grad_norm_(params, 35.0)
    opt.step()
    # inference still works after the step (packed deform weights are refreshed)
    m.eval()
    out = m.predict(x)
    assert len(out[0]) == 4


def test_full_size_frame_properties():
    """Config 2 at full size (1920x1280): size-independent invariants of the detector output + determinism."""
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det, detections_to_wire
    m = Detectron2Det(seed=0).eval().cuda()
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 256, (1, 3, 1280, 1920), generator=g).float().cuda()
    (b1, s1, c1), = m.predict_device(x)
    (b2, s2, c2), = m.predict_device(x)
    # run-to-run: same detections; boxes equal up to the library GEMM / conv kernels' accumulation order (the hand-
    # written kernels are deterministic, hipBLASLt / MIOpen algorithm selection is not bit-stable)
    assert b1.shape == b2.shape and torch.equal(c1, c2)
    assert torch.allclose(b1, b2, atol=2e-2) and torch.allclose(s1, s2, atol=1e-5)
    assert 0 < b1.shape[0] <= 100                                                          # top-100 (TEST.DETECTIONS_PER_IMAGE)
    assert torch.isfinite(b1).all() and torch.isfinite(s1).all()
    assert (s1 > 0.01).all() and (s1 <= 1).all()                                           # SCORE_THRESH_TEST 0.01
    assert torch.all(s1[:-1] >= s1[1:])                                                    # NMS keeps score order
    assert (b1[:, 0] >= 0).all() and (b1[:, 1] >= 0).all() and (b1[:, 2] <= 1920).all() and (b1[:, 3] <= 1280).all()
    assert (b1[:, 2] >= b1[:, 0]).all() and (b1[:, 3] >= b1[:, 1]).all()
    assert set(c1.tolist()) <= {0, 1, 2, 3}
    # per class, no two kept boxes overlap more than the NMS threshold 0.5
    from oracle import detector_ref as R
    for c in range(4):
        bc = b1[c1 == c].cpu()
        if len(bc) > 1:
            keep = R.nms_sorted(bc, None, 0.5 + 1e-5)
            assert keep.all()
    xywh, score, cat = detections_to_wire(b1, s1, c1, 1920, 1280)
    assert torch.equal(xywh, torch.trunc(xywh)) and (cat >= 1).all() and (cat <= 4).all()
    assert torch.allclose(score * 1e5, torch.round(score * 1e5), atol=1e-6)


def test_config4_full_size_tta_x15_hflip_properties():
    """Config 4's detector pass at its real size: a 1920x1280 frame with --tta x1.5,hflip (the reference's documented run, README.md:37;
    nn/tta.py:228-267) = one pass over a 2880x1920 image.  Same size-independent invariants as the plain full-size test, on boxes
    mapped back to the ORIGINAL frame (cx mirrored, coordinates divided by 1.5) - plus: the un-mirrored pass at the same scale finds
    the mirror image of a mirrored input."""
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det, detections_to_wire
    from oracle import detector_ref as R
    m = Detectron2Det(seed=0).eval().cuda()
    g = torch.Generator().manual_seed(2)
    x = torch.randint(0, 256, (1, 3, 1280, 1920), generator=g).float().cuda()
    (b1, s1, c1), = m.predict_device(x, 1.5, True, False)
    assert m.last_input_size == (1920, 2880)                                                  # what the detector really saw
    assert 0 < b1.shape[0] <= 100 and torch.isfinite(b1).all() and torch.isfinite(s1).all()
    assert (s1 > 0.01).all() and (s1 <= 1).all() and torch.all(s1[:-1] >= s1[1:])
    # boxes live in the transformed image (2880 x 1920, mirrored)
    assert (b1[:, 0] >= 0).all() and (b1[:, 1] >= 0).all() and (b1[:, 2] <= 2880).all() and (b1[:, 3] <= 1920).all()
    assert (b1[:, 2] >= b1[:, 0]).all() and (b1[:, 3] >= b1[:, 1]).all() and set(c1.tolist()) <= {0, 1, 2, 3}
    for c in range(4):
        bc = b1[c1 == c].cpu()
        if len(bc) > 1:
            assert R.nms_sorted(bc, None, 0.5 + 1e-5).all()
    # the wire conversion maps them back: integer xywh inside the ORIGINAL 1920 x 1280 frame
    xywh, score, cat = detections_to_wire(b1 / 1.5, s1, c1, 1920, 1280)
    assert torch.equal(xywh, torch.trunc(xywh)) and (xywh[:, 0] + xywh[:, 2] <= 1920 + 1).all() and (xywh[:, 1] + xywh[:, 3] <= 1280 + 1).all()
    # mirror consistency: hflip of the input followed by the un-flipped x1.5 pass sees exactly the image the flipped pass sees
    (b2, s2, c2), = m.predict_device(torch.flip(x, [3]), 1.5, False, False)
    assert abs(b2.shape[0] - b1.shape[0]) <= 2
    k = min(20, b1.shape[0], b2.shape[0])
    d = (b1[:k, None, :] - b2[None, :k, :]).abs().sum(-1).min(1).values
    assert int((d < 0.5).sum()) >= int(0.9 * k), d                                         # same boxes up to library-kernel accumulation order


def _random_model_file(tmp_path, seed=0):
    """A {args, kwargs, state_dict} model file in the reference's format (random weights)."""
    from waymo_2d_tracking_amd.detnet import nn as detnn
    net = detnn.create('detectron2:Misc/cascade_mask_rcnn_X_152_32x8d_FPN_IN5k_gn_dconv.yaml',
                       ['vehicle', 'pedestrian', 'sign', 'cyclist'], pretrained=None, freeze_pretrained=2, frozen_bn=True, seed=seed)
    path = tmp_path / 'random.model'
    net.save(str(path))
    return str(path)


def _rows_match(rows, ref_rows):
    """Wire rows (integer boxes, 5-decimal scores) of two float32 detector runs: every row has a partner within one unit per coordinate / one unit of the 5th
    decimal, at most 3 coordinates differ at all (see test_inference_cli_on_image_folder)."""
    assert abs(len(rows) - len(ref_rows)) <= 1
    matched, coord_diffs = 0, 0
    for r in rows:
        cands = [q for q in ref_rows if q['image_id'] == r['image_id'] and q['category_id'] == r['category_id'] and
                 max(abs(a - b) for a, b in zip(q['bbox'], r['bbox'])) <= 1 and abs(q['score'] - r['score']) <= 1.01e-5]
        if cands:
            matched += 1
            coord_diffs += min(sum(a != b for a, b in zip(q['bbox'], r['bbox'])) for q in cands)
    assert matched >= len(rows) - 1, (matched, len(rows))
    assert coord_diffs <= 3, coord_diffs
    assert list(dict.fromkeys(r['image_id'] for r in rows)) == list(dict.fromkeys(r['image_id'] for r in ref_rows))


def test_inference_cli_with_frames_in_flight_equals_the_per_image_loop(tmp_path):
    """Round 6: inference.py with `--inflight 2` (default: every image size captured once per lane as a hipGraph, two passes in flight, results collected
    one image behind - nn.GraphLanePredictor) writes the same detection JSON as `--inflight 0` (one eager Detectron2Det.predict per image): two image
    sizes (the front / side cameras), with and without the folded TTA plan, more images than lanes, an odd count."""
    import json
    from PIL import Image
    from waymo_2d_tracking_amd.detnet import inference as I
    rng = np.random.default_rng(1)
    root = tmp_path / 'images'
    for i in range(7):
        cam = ('FRONT', 'SIDE_LEFT')[i % 2]
        d = root / 'seg' / str(100 + i)
        d.mkdir(parents=True, exist_ok=True)
        arr = rng.integers(30, 200, ((96, 64)[i % 2], 160, 3), dtype=np.uint8)
        Image.fromarray(arr).save(d / (cam + '.jpg'), quality=92)
    model = _random_model_file(tmp_path)
    for tta in ([], ['--tta', 'x1.5,hflip']):
        out = {}
        for lanes in (0, 2):
            path = tmp_path / ('sub%d.json' % lanes)
            I.main(['-m', model, '-i', str(root), '--export', str(path), '--batch-size=1', '--inflight', str(lanes)] + tta)
            out[lanes] = json.load(open(path))
        assert len(out[0]) > 0
        _rows_match(out[2], out[0])


def test_inference_cli_on_image_folder(tmp_path):
    """inference.py drop-in: image folder -> detection JSON (coco.py:229-252 rows) == predict + load_prediction; -o writes the
    prediction store, --resume skips the tested samples and merges, --auto-contrast changes the input like PIL does."""
    import json
    from PIL import Image, ImageOps
    from waymo_2d_tracking_amd.detnet import inference as I
    from waymo_2d_tracking_amd.detnet.trainer import Predictions
    rng = np.random.default_rng(0)
    root = tmp_path / 'images'
    names = (('segA', 100, 'FRONT'), ('segA', 200, 'FRONT'), ('segB', 100, 'SIDE_LEFT'), ('segB', 200, 'SIDE_LEFT'))
    for seg, ts, cam in names:
        d = root / seg / str(ts)
        d.mkdir(parents=True, exist_ok=True)
        # two camera frames as JPEG (decoded by the HIP decoder inside the loader), two as PNG (PIL, as in the reference)
        arr = rng.integers(30, 200, (96, 160, 3), dtype=np.uint8)
        if seg == 'segA':
            Image.fromarray(arr).save(d / (cam + '.jpg'), quality=92)
        else:
            Image.fromarray(arr).save(d / (cam + '.png'))
    model = _random_model_file(tmp_path)
    out = tmp_path / 'sub.json'
    rows0 = I.main(['-m', model, '-i', str(root), '--export', str(out), '--batch-size=1', '--tta', 'x1.5,hflip', '-o', str(tmp_path / 'o')])
    rows = json.load(open(out))
    assert len(rows) > 0 and len(rows) == len(rows0['image'])
    ids = {r['image_id'] for r in rows}
    assert ids <= {'%s/%d/%s' % n for n in names}
    for r in rows:
        assert set(r) == {'image_id', 'category_id', 'bbox', 'score'}
        assert all(isinstance(v, int) for v in r['bbox']) and 1 <= r['category_id'] <= 4
        assert round(r['score'], 5) == r['score']
    # the exported rows == Detectron2Det.predict + load_prediction (the reference's dict path) on the same files
    from waymo_2d_tracking_amd.detnet import nn as detnn
    from waymo_2d_tracking_amd.detnet.nn.tta import TTA
    net = detnn.load(model).cuda().eval()
    sizes, preds = {}, {}
    for image_id, path in I.list_images(str(root)):
        img = Image.open(path).convert('RGB')
        x = torch.as_tensor(np.asarray(img, dtype=np.float32).transpose(2, 0, 1)).unsqueeze(0).cuda()
        preds[image_id] = TTA(net, ['x1.5', 'hflip']).predict(x)[0]
        sizes[image_id] = (img.width, img.height)
    ref_rows = I.load_prediction(sizes, net.classnames, preds)
    # (two float32 runs of the detector are not bit-identical - library convolutions with atomics, algorithms picked by timing - so a coordinate within
    # 1e-4 px of a rounding boundary or two scores within 1e-7 of each other may come out the other way: rows are matched, not compared by position)
    assert abs(len(rows) - len(ref_rows)) <= 1
    assert [r['image_id'] for r in rows[:1]] == [r['image_id'] for r in ref_rows[:1]]
    # wire rows carry INTEGER boxes and 5-decimal scores: a float coordinate on a rounding boundary moves its integer by exactly 1, a score by exactly
    # 1e-5 - nothing in between exists.  Bounded (round 6): a row matches when every coordinate is within 1 and the score within one unit of the 5th
    # decimal; over ALL matched rows at most 3 coordinates may differ at all (expected ~0.2 of ~1200 with 1e-4 px of float32 noise)
    matched, coord_diffs = 0, 0
    for r in rows:
        cands = [q for q in ref_rows if q['image_id'] == r['image_id'] and q['category_id'] == r['category_id'] and
                 max(abs(a - b) for a, b in zip(q['bbox'], r['bbox'])) <= 1 and abs(q['score'] - r['score']) <= 1.01e-5]
        if cands:
            matched += 1
            coord_diffs += min(sum(a != b for a, b in zip(q['bbox'], r['bbox'])) for q in cands)
    assert matched >= len(rows) - 1, (matched, len(rows))
    assert coord_diffs <= 3, coord_diffs
    assert list(dict.fromkeys(r['image_id'] for r in rows)) == list(dict.fromkeys(r['image_id'] for r in ref_rows))
    # -o wrote the store; --resume on a store that holds two of the four images detects only the others and merges
    store = Predictions.open(tmp_path / 'o')
    assert len(store) == 4
    part = Predictions(store.classnames, store.image_ids)
    for k in list(store.keys())[:2]:
        part[k] = store[k]
    part.save(tmp_path / 'part' / 'detections.pkl')
    out2 = tmp_path / 'sub2.json'
    I.main(['-m', model, '-i', str(root), '--export', str(out2), '--tta', 'x1.5,hflip', '--resume', str(tmp_path / 'part')])
    rows2 = json.load(open(out2))
    kept = set(list(store.keys())[:2])
    assert [r for r in rows2 if r['image_id'] in kept] == [r for r in rows if r['image_id'] in kept]     # stored samples: unchanged
    # re-detected samples: same images in the same (data-set) order; values may differ in the last digits between two runs
    # (cudnn.benchmark / TunableOp pick algorithms by timing)
    assert list(dict.fromkeys(r['image_id'] for r in rows2)) == list(dict.fromkeys(r['image_id'] for r in rows))
    # --auto-contrast = ImageOps.autocontrast, bit for bit
    img = Image.open(I.list_images(str(root))[0][1]).convert('RGB')
    got = I.autocontrast_(torch.from_numpy(np.array(img)).cuda()).cpu().numpy()
    assert np.array_equal(got, np.asarray(ImageOps.autocontrast(img)))
    flat = Image.fromarray(np.full((8, 8, 3), 77, np.uint8))
    assert np.array_equal(I.autocontrast_(torch.from_numpy(np.array(flat)).cuda()).cpu().numpy(), np.asarray(ImageOps.autocontrast(flat)))
    out3 = tmp_path / 'sub3.json'
    I.main(['-m', model, '-i', str(root), '--export', str(out3), '--auto-contrast=1'])
    assert json.load(open(out3)) != json.load(open(out))
    # unsupported flags fail loudly; no output target is an error like in the reference
    with pytest.raises(NotImplementedError):
        I.main(['-m', model, '-i', str(root), '--export', str(out3), '--clahe=1'])
    with pytest.raises(UserWarning):
        I.main(['-m', model, '-i', str(root)])


def test_training_step_through_ddp_over_rccl(tmp_path):
    """The DDP leg of config 5 (bench.py --stage train with N > 1; reference: trainer/optim/__init__.py:116 wraps the model in
    DistributedDataParallel): one rank over RCCL in a fresh process - bucketed gradient all-reduce hooks fire during the HIP
    backward kernels; the loss equals the unwrapped step's, the gradients agree up to the run-to-run spread of the backward
    kernels' float atomics (ROIAlign / deformable col2im scatter: 1e-2 of a tensor's largest entry)."""
    import json
    import subprocess
    import sys
    from waymo_2d_tracking_amd.launcher import free_port
    code = r'''
import json, os, sys
sys.path.insert(0, %r)
import torch, torch.nn as nn, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(%d), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
from waymo_2d_tracking_amd.detnet.nn import training
m = Detectron2Det(seed=4).cuda().train()
params = training.set_trainable(m.model)
class W(nn.Module):
    def __init__(self, model):
        super().__init__(); self.model = model
    def forward(self, img, b, c):
        return sum(training.losses(self.model, img, b, c, choose=training.first_choice).values())
g = torch.Generator().manual_seed(11)
img = torch.randint(0, 256, (1, 3, 160, 224), generator=g).float().cuda()
gt = torch.tensor([[20., 30., 120., 150.], [100., 40., 215., 155.], [5., 5., 60., 60.]]).cuda()
cls = torch.tensor([0, 1, 3]).cuda()
plain = W(m.model)
l0 = plain(img, gt, cls); l0.backward()
g0 = {n: p.grad.clone() for n, p in m.model.named_parameters() if p.requires_grad}
for p in params: p.grad = None
ddp = nn.parallel.DistributedDataParallel(plain, device_ids=[0])
l1 = ddp(img, gt, cls); l1.backward()
worst = max(float((p.grad - g0[n]).abs().max() / (g0[n].abs().max() + 1e-30)) for n, p in m.model.named_parameters() if p.requires_grad)
rep = dict(l0=float(l0), l1=float(l1), worst=worst, n=len(g0), missing=[n for n, p in m.model.named_parameters() if p.requires_grad and p.grad is None])
dist.barrier(); dist.destroy_process_group()
json.dump(rep, open(%r, 'wt'))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), free_port(), str(tmp_path / 'rep.json'))
    p = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'), capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    rep = json.load(open(tmp_path / 'rep.json'))
    assert not rep['missing'] and rep['n'] > 300
    assert abs(rep['l0'] - rep['l1']) <= 1e-4 * abs(rep['l0']) and rep['worst'] <= 1e-2, rep
