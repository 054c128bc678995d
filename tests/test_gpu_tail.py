"""GPU parity tests of the fused, static-shape detector tail (csrc/det_tail.hip) against the torch sequences they replace
(which tests/test_gpu_detector.py pins to oracle/detector_ref.py): bit-exact."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from waymo_2d_tracking_amd.detnet.nn import ops  # noqa: E402


def _boxes(g, n, size=1000.0, device='cuda'):
    b = torch.rand((n, 4), generator=g) * size
    b[:, 2:] = b[:, :2] + torch.rand((n, 2), generator=g) * 300
    return b.to(device)


@pytest.mark.parametrize('sizes,k,ties', [((20000, 9000, 3000, 700, 50), 1000, False),
                                          ((460800, 115200, 28800, 7200, 1800), 1000, False),
                                          ((30000, 8192, 8193, 1), 1000, True),
                                          ((500,), 100, True)])
def test_rpn_topk_decode_matches_topk_and_decode(sizes, k, ties):
    g = torch.Generator().manual_seed(len(sizes) * 7 + k)
    logits, deltas, anchors = [], [], []
    for n in sizes:
        lg = torch.randn(n, generator=g)
        if ties:
            lg = torch.round(lg * 8) / 8                      # many equal logits: lower anchor index first
        logits.append(lg.cuda())
        deltas.append((torch.randn((n, 4), generator=g) * 0.7).cuda())
        anchors.append(_boxes(g, n))
    deltas[0][3, 2] = 9.0                                     # scale clamp
    boxes, scores, group, valid = ops.rpn_topk_decode(logits, deltas, anchors, k, 1280.0, 1920.0)
    row = 0
    for l, n in enumerate(sizes):
        kk = min(k, n)
        top, idx = torch.sort(logits[l], descending=True, stable=True)
        top, idx = top[:kk], idx[:kk]
        if not ties:                                          # without ties torch.topk gives the same list
            t2, i2 = torch.topk(logits[l], kk, sorted=True)
            assert torch.equal(t2, top) and torch.equal(i2, idx)
        want = ops.decode_boxes(deltas[l], anchors[l], (1.0, 1.0, 1.0, 1.0), idx, (1280.0, 1920.0))
        assert torch.equal(scores[row:row + kk], top)
        assert torch.equal(boxes[row:row + kk], want)
        ok = ((want[:, 2] - want[:, 0]) > 0) & ((want[:, 3] - want[:, 1]) > 0)
        assert torch.equal(valid[row:row + kk].bool(), ok)
        assert torch.equal(group[row:row + kk], torch.where(ok, torch.full_like(group[row:row + kk], l), torch.full_like(group[row:row + kk], -1)))
        row += kk
    assert row == boxes.shape[0]


@pytest.mark.parametrize('n', [1, 2, 777, 5000, 8192])
def test_sort_candidates_is_a_stable_descending_sort(n):
    g = torch.Generator().manual_seed(n)
    scores = (torch.round(torch.randn(n, generator=g) * 4) / 4).cuda()      # ties
    boxes = _boxes(g, n)
    group = torch.randint(-1, 5, (n,), generator=g, dtype=torch.int32).cuda()
    valid = (group >= 0).to(torch.uint8)
    sb, ss, sg, sv, order = ops.sort_candidates(boxes, scores, group, valid)
    want = torch.argsort(scores, descending=True, stable=True)
    assert torch.equal(order, want)
    assert torch.equal(ss, scores[want]) and torch.equal(sb, boxes[want])
    assert torch.equal(sg, group[want]) and torch.equal(sv, valid[want])


def _tail(thresh=0.05, nms=0.5, topk=100):
    from waymo_2d_tracking_amd.detnet.nn.cascade_rcnn import CascadeRCNN
    ns = types.SimpleNamespace(score_thresh=thresh, nms_thresh=nms, topk=topk)
    ns._inference_many_classes = types.MethodType(CascadeRCNN._inference_many_classes, ns)
    return CascadeRCNN, ns


@pytest.mark.parametrize('r,nc,n_valid', [(1000, 4, 1000), (1000, 4, 613), (37, 4, 37), (300, 3, 0), (2048, 4, 2048)])
def test_box_inference_fused_equals_torch_sequence(r, nc, n_valid):
    CascadeRCNN, ns = _tail()
    g = torch.Generator().manual_seed(r + nc + n_valid)
    boxes = _boxes(g, r, 1800.0)
    boxes[::3] += torch.randn((boxes[::3].shape[0], 4), generator=g).cuda() * 0.5      # clusters -> suppression
    boxes[1::3] = boxes[0:-1:3][:boxes[1::3].shape[0]] + 3.0
    stages = [torch.softmax(torch.randn((r, nc + 1), generator=g) * 2, dim=1).cuda() for _ in range(3)]
    if r > 20:
        boxes[5, 1] = float('nan'); boxes[6, 2] = float('inf'); stages[1][7, 0] = float('nan'); stages[2][8, nc] = float('inf')
        boxes[9] = torch.tensor([-50.0, -20.0, 5000.0, 4000.0])                        # clipped
    nv = torch.tensor([n_valid], dtype=torch.int32, device='cuda')
    got = CascadeRCNN.inference(ns, boxes, stages, 1280.0, 1920.0, nv)
    want = ns._inference_many_classes(boxes, (stages[0] + stages[1] + stages[2]) * (1.0 / 3), 1280.0, 1920.0, nv)
    assert int(got[3].item()) == int(want[3].item())
    if n_valid:
        assert int(got[3].item()) > 0
    for a, b in zip(got[:3], want[:3]):
        assert a.dtype == b.dtype and torch.equal(a, b)


def test_gather_kept_proposals_are_zero_padded():
    g = torch.Generator().manual_seed(5)
    n = 3000
    boxes = _boxes(g, n)
    keep = (torch.rand(n, generator=g) < 0.2).to(torch.uint8).cuda()
    valid = (torch.rand(n, generator=g) < 0.9).to(torch.uint8).cuda()
    scores = torch.rand(n, generator=g).cuda()
    order = torch.randperm(n, generator=g).cuda()
    for cap in (100, 1000):
        props, cnt = ops.gather_kept(keep, valid, boxes, scores, order, cap)
        sel = torch.nonzero(keep.bool() & valid.bool()).flatten()[:cap]
        assert int(cnt.item()) == sel.numel()
        assert torch.equal(props[:sel.numel()], boxes[sel]) and float(props[sel.numel():].abs().sum()) == 0.0


@pytest.mark.parametrize('hflip', [False, True])
def test_wire_kernel_equals_the_torch_sequence(hflip):
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import detections_to_wire
    g = torch.Generator().manual_seed(11)
    n, wo, ho, W, H = 100, 2880, 1920, 1920, 1280              # x1.5 TTA pass scaled back to the original image
    boxes = _boxes(g, n, 2500.0)
    scores = torch.rand(n, generator=g).cuda()
    scores[3] = 0.123455; scores[4] = 0.5; scores[5] = 0.999995
    classes = torch.randint(0, 4, (n,), generator=g).cuda()
    cnt = torch.tensor([61], dtype=torch.int32, device='cuda')
    xywhs, cat = ops.detections_to_wire(boxes, scores, classes, cnt, wo, ho, W, H, hflip)
    # the torch sequence evaluated on the CPU (true division by 1e5, as Python's round(score, 5) in the reference; torch's GPU
    # tensor / scalar is a multiplication by the reciprocal and differs in the last bit for some scores)
    b = boxes.cpu()
    if hflip:
        b = torch.stack((wo - b[:, 2], b[:, 1], wo - b[:, 0], b[:, 3]), dim=1)
    xywh, score, c = detections_to_wire(b, scores.cpu(), classes.cpu(), wo, ho, W, H)
    assert torch.equal(xywhs[:4].t().cpu(), xywh) and torch.equal(xywhs[4].cpu(), score)
    assert [round(float(v), 5) for v in scores.cpu()] == xywhs[4].cpu().tolist()      # detectron2_det/__init__.py:129
    assert torch.equal(cat[:61].cpu(), c[:61]) and int(cat[61:].abs().sum()) == 0
    # same image size in and out, no count: every slot is real; the module-level function takes the same launch on the GPU
    xywh_g, score_g, c_g = detections_to_wire(boxes, scores, classes, W, H)
    xywh, score, c = detections_to_wire(boxes.cpu(), scores.cpu(), classes.cpu(), W, H)
    assert torch.equal(xywh_g.cpu(), xywh) and torch.equal(score_g.cpu(), score) and torch.equal(c_g.cpu(), c)


@pytest.mark.parametrize('m,n,k', [(9600, 1024, 1024), (38400, 512, 512), (2400, 2048, 2048), (37, 24, 16)])
def test_gemm_lt_residual_bias_relu(m, n, k):
    """wd_gemm_lt_f32: relu(a @ w.T + residual + bias) in one hipBLASLt launch, in place on the residual buffer."""
    g = torch.Generator().manual_seed(m + n)
    a = torch.randn((m, k), generator=g).cuda()
    w = (torch.randn((n, k), generator=g) / k ** 0.5).cuda()
    bias = torch.randn(n, generator=g).cuda()
    res = torch.randn((m, n), generator=g).cuda()
    want = torch.relu(a.double() @ w.double().t() + res.double() + bias.double())
    buf = res.clone()
    got = ops.gemm_lt(a, w, bias, buf, True, out=buf)
    assert got.data_ptr() == buf.data_ptr()
    np.testing.assert_allclose(got.cpu().double().numpy(), want.cpu().numpy(), rtol=2e-4, atol=2e-4)
    got2 = ops.gemm_lt(a, w, bias, res, False)                  # out of place, no ReLU; second call takes the cached plan
    np.testing.assert_allclose(got2.cpu().double().numpy(), (a.double() @ w.double().t() + res.double() + bias.double()).cpu().numpy(),
                               rtol=2e-4, atol=2e-4)
    got3 = ops.gemm_lt(a, w, None, None, False)
    np.testing.assert_allclose(got3.cpu().double().numpy(), (a.double() @ w.double().t()).cpu().numpy(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('sizes', [(1000, 1000, 1000, 1000, 1000), (1000, 37, 1, 640), (64,), (1000, 0, 500)])
def test_segmented_nms_equals_grouped_nms_on_the_globally_sorted_list(sizes):
    """wd_nms_segmented_f32 (one sweep chain per FPN level, in parallel) keeps exactly the boxes that the single-chain NMS with
    group ids keeps, and the RPN selection built on it (sort with valid = kept, first `post`) equals sort -> NMS -> first `post`."""
    g = torch.Generator().manual_seed(sum(sizes) + len(sizes))
    boxes_l, scores_l, lvl_l = [], [], []
    for l, n in enumerate(sizes):
        b = _boxes(g, n, 600.0)
        if n > 3:
            b[1::3] = b[0:-1:3][:b[1::3].shape[0]] + 2.0              # overlapping clusters -> suppression chains
        s = torch.sort(torch.round(torch.randn(n, generator=g) * 16) / 16, descending=True).values.cuda()    # sorted, with ties
        boxes_l.append(b); scores_l.append(s); lvl_l.append(torch.full((n,), l, dtype=torch.int32, device='cuda'))
    boxes, scores, lvls = torch.cat(boxes_l), torch.cat(scores_l), torch.cat(lvl_l)
    n = boxes.shape[0]
    ok = (torch.rand(n, generator=g) < 0.95).to(torch.uint8).cuda()
    lvls = torch.where(ok.bool(), lvls, torch.full_like(lvls, -1))      # "empty" boxes neither suppress nor get selected
    seg = [0]
    for m in sizes:
        seg.append(seg[-1] + m)
    keep = ops.nms_segmented(boxes, lvls, seg, 0.7)
    sb, ss, sg, sv, order = ops.sort_candidates(boxes, scores, lvls, ok, keep)
    got, got_n = ops.gather_kept(sv, sv, sb, ss, order, 300)
    # reference order of operations: global stable sort, one NMS chain with group ids, first 300 kept & valid
    rb, rs, rg, rv, rorder = ops.sort_candidates(boxes, scores, lvls, ok)
    rkeep = ops.nms_sorted_mask(rb, rg, 0.7)
    want, want_n = ops.gather_kept(rkeep, rv, rb, rs, rorder, 300)
    real = rv.bool()                                                  # rows flagged empty share group -1: what they do to each other is irrelevant
    assert torch.equal(keep[rorder][real], rkeep[real])
    assert int(got_n.item()) == int(want_n.item()) and torch.equal(got, want)
