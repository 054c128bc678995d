"""TEST INFRASTRUCTURE ONLY - CPU (PyTorch, float32/float64) restatement of the whole detector graph.

Runs the SAME parameters as waymo_2d_tracking_amd.detnet.nn.cascade_rcnn.CascadeRCNN through plain torch CPU ops
(F.conv2d, F.grid_sample based deformable sampling, vectorised ROIAlign, a Python greedy NMS).  Used
  * by tests/test_gpu_detector.py as the parity reference of the HIP-backed graph (feature maps, head outputs),
  * by bench.py's cpu_baseline leg (timed on the GPU box's host cores on a bounded sample).
The op semantics follow SURVEY.md App. C (detectron2 is not vendored: "parity unpinned").  Never imported by the
product package.
"""
import math

import torch
import torch.nn.functional as F

SCALE_CLAMP = math.log(1000.0 / 16)


def conv1x1(mod, x, relu=False, residual=None, stride=1):
    w = mod.weight.to(x.dtype)
    y = F.conv2d(x, w.view(w.shape[0], w.shape[1], 1, 1), mod.bias.to(x.dtype), stride)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def convbn(mod, x, relu=False):
    y = F.conv2d(x, mod.weight.to(x.dtype), mod.bias.to(x.dtype), mod.stride, mod.pad, 1, mod.groups)
    return F.relu(y) if relu else y


def deform_conv3x3(x, offset, weight, groups, stride, pad):
    """DeformConv via 9 bilinear resamplings (grid_sample, zero padding == per-corner bounds checks)."""
    n, c, h, w = x.shape
    ho = (h + 2 * pad - 3) // stride + 1
    wo = (w + 2 * pad - 3) // stride + 1
    ys = torch.arange(ho, dtype=x.dtype).view(1, ho, 1) * stride - pad
    xs = torch.arange(wo, dtype=x.dtype).view(1, 1, wo) * stride - pad
    taps = []
    for k in range(9):
        kh, kw = k // 3, k % 3
        if offset is None:
            py = (ys + kh).expand(n, ho, wo)
            px = (xs + kw).expand(n, ho, wo)
        else:
            py = ys + kh + offset[:, 2 * k]
            px = xs + kw + offset[:, 2 * k + 1]
        gx = 2 * px / max(w - 1, 1) - 1
        gy = 2 * py / max(h - 1, 1) - 1
        taps.append(F.grid_sample(x, torch.stack((gx, gy), dim=-1), mode='bilinear', padding_mode='zeros', align_corners=True))
    col = torch.stack(taps, dim=2).reshape(n, c * 9, ho, wo)
    wt = weight.to(x.dtype).reshape(weight.shape[0], -1, 1, 1)
    return F.conv2d(col, wt, None, 1, 0, 1, groups)


def block_decisions(relu1, offset, relu2, relu3, stride):
    """The DISCRETE decisions of one bottleneck block, in a form two runs can compare exactly: the three ReLU masks (conv1, conv2, block output; from
    the post-ReLU maps) and, for a deformable block, the bilinear cell (floor of the sampling position) of every (pixel, tap, axis).  A gradient is a
    piecewise-smooth function of the forward values: two runs whose decisions agree differ by rounding only, a run whose decision fell the other way
    (a ReLU input or a sampling position within float32 noise of the boundary) differs by a finite step (tests/test_gpu_detector.py)."""
    d = {'relu1': (relu1.detach() > 0).cpu(), 'relu2': (relu2.detach() > 0).cpu(), 'relu3': (relu3.detach() > 0).cpu(), 'cells': None}
    if offset is not None:
        off = offset.detach().double().cpu()
        n, _, ho, wo = off.shape
        ys = torch.arange(ho, dtype=torch.float64).view(1, ho, 1) * stride - 1
        xs = torch.arange(wo, dtype=torch.float64).view(1, 1, wo) * stride - 1
        cells = []
        for k in range(9):
            cells.append(torch.floor(ys + k // 3 + off[:, 2 * k]))
            cells.append(torch.floor(xs + k % 3 + off[:, 2 * k + 1]))
        d['cells'] = torch.stack(cells, 1).to(torch.int32)
    return d


def bottleneck(b, x, record=None):
    sc = x if b.shortcut is None else conv1x1(b.shortcut, x, stride=b.stride)
    out1 = conv1x1(b.conv1, x, relu=True)
    offset = convbn(b.conv2_offset, out1) if b.deform else None
    out = deform_conv3x3(out1, offset, b.conv2_weight, 32, b.stride, 1)
    out2 = F.relu(out * b.conv2_scale.to(x.dtype).view(1, -1, 1, 1) + b.conv2_bias.to(x.dtype).view(1, -1, 1, 1))
    out3 = conv1x1(b.conv3, out2, relu=True, residual=sc)
    if record is not None:
        record.append(block_decisions(out1, offset, out2, out3, b.stride))
    return out3


def backbone(bb, x, record=None):
    x = convbn(bb.stem, x, relu=True)
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for stage in (bb.res2, bb.res3, bb.res4, bb.res5):
        for blk in stage:
            x = bottleneck(blk, x, record)
        feats.append(x)
    prev = conv1x1(bb.lateral[3], feats[3])
    outs = [convbn(bb.output[3], prev)]
    for i in (2, 1, 0):
        top = F.interpolate(prev, scale_factor=2.0, mode='nearest')
        prev = conv1x1(bb.lateral[i], feats[i], residual=top)
        outs.insert(0, convbn(bb.output[i], prev))
    outs.append(F.max_pool2d(outs[3], 1, 2, 0))
    return outs


def apply_deltas(deltas, boxes, weights):
    wx, wy, ww, wh = weights
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    cx = boxes[:, 0] + 0.5 * widths
    cy = boxes[:, 1] + 0.5 * heights
    dx, dy = deltas[:, 0] / wx, deltas[:, 1] / wy
    dw = torch.clamp(deltas[:, 2] / ww, max=SCALE_CLAMP)
    dh = torch.clamp(deltas[:, 3] / wh, max=SCALE_CLAMP)
    pcx, pcy = dx * widths + cx, dy * heights + cy
    pw, ph = torch.exp(dw) * widths, torch.exp(dh) * heights
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=1)


def clip_boxes(b, h, w):
    return torch.stack((b[:, 0].clamp(0, w), b[:, 1].clamp(0, h), b[:, 2].clamp(0, w), b[:, 3].clamp(0, h)), dim=1)


def nms_sorted(boxes, idxs, thr):
    n = boxes.shape[0]
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    keep = torch.ones(n, dtype=torch.bool)
    for i in range(n - 1):
        if not keep[i]:
            continue
        left = torch.maximum(boxes[i, 0], boxes[i + 1:, 0]); right = torch.minimum(boxes[i, 2], boxes[i + 1:, 2])
        top = torch.maximum(boxes[i, 1], boxes[i + 1:, 1]); bottom = torch.minimum(boxes[i, 3], boxes[i + 1:, 3])
        inter = (right - left).clamp(min=0) * (bottom - top).clamp(min=0)
        sup = inter / (area[i] + area[i + 1:] - inter) > thr
        if idxs is not None:
            sup &= idxs[i + 1:] == idxs[i]
        keep[i + 1:] &= ~sup
    return keep


def batched_nms(boxes, scores, idxs, thr):
    if boxes.shape[0] == 0:
        return torch.empty(0, dtype=torch.int64)
    order = torch.argsort(scores, descending=True, stable=True)
    keep = nms_sorted(boxes[order].float(), None if idxs is None else idxs[order], thr)
    return order[keep]


def rpn(r, feats, img_h, img_w):
    boxes_l, scores_l, lvl_l = [], [], []
    for l, f in enumerate(feats):
        t = convbn(r.conv, f, relu=True)
        logits = conv1x1(r.objectness, t).permute(0, 2, 3, 1).reshape(-1)
        deltas = conv1x1(r.deltas, t).permute(0, 2, 3, 1).reshape(-1, 4)
        k = min(r.pre, logits.numel())
        top, idx = torch.topk(logits, k, sorted=True)
        anchors = r.anchors(l, f.shape[2], f.shape[3], f.device).to(f.dtype)
        boxes_l.append(apply_deltas(deltas[idx], anchors[idx], (1.0, 1.0, 1.0, 1.0)))
        scores_l.append(top)
        lvl_l.append(torch.full((k,), l, dtype=torch.int32))
    boxes = clip_boxes(torch.cat(boxes_l), img_h, img_w)
    scores, lvls = torch.cat(scores_l), torch.cat(lvl_l)
    ok = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
    boxes, scores, lvls = boxes[ok], scores[ok], lvls[ok]
    keep = batched_nms(boxes, scores, lvls, r.thr)[: r.post]
    return boxes[keep]


def roi_pool_fpn(feats, boxes, scales, pooled=7, min_level=2, canonical_level=4, canonical_size=224.0):
    """Vectorised ROIAlign (aligned=True, adaptive grid) over FPN levels; boxes (R,4) for batch 0 -> (R,C,P,P)."""
    R = boxes.shape[0]
    C = feats[0].shape[1]
    dt = feats[0].dtype
    out = torch.zeros((R, C, pooled, pooled), dtype=dt)
    size = torch.sqrt(((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).float())
    lvl = torch.floor(canonical_level + torch.log2(size / canonical_size + 1e-8)).clamp(min_level, min_level + len(feats) - 1).long()
    P = pooled
    for li, feat in enumerate(feats):
        sel = torch.nonzero(lvl == li + min_level).flatten()
        if not len(sel):
            continue
        H, W = feat.shape[2], feat.shape[3]
        fl = feat[0].permute(1, 2, 0).reshape(H * W, C)
        b = boxes[sel].float() * scales[li] - 0.5
        rw, rh = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
        bw, bh = rw / P, rh / P
        gws, ghs = torch.ceil(rw / P).long(), torch.ceil(rh / P).long()
        for gh in torch.unique(ghs).tolist():
            for gw in torch.unique(gws[ghs == gh]).tolist():
                m = torch.nonzero((ghs == gh) & (gws == gw)).flatten()
                if gh <= 0 or gw <= 0:
                    continue
                step = max(1, 256 // (gh * gw))
                for c0 in range(0, len(m), step):
                    mm = m[c0:c0 + step]
                    n = len(mm)
                    iy = (torch.arange(gh, dtype=torch.float32) + 0.5).view(1, 1, gh)
                    ix = (torch.arange(gw, dtype=torch.float32) + 0.5).view(1, 1, gw)
                    ph = torch.arange(P, dtype=torch.float32).view(1, P, 1)
                    y = b[mm, 1].view(n, 1, 1) + ph * bh[mm].view(n, 1, 1) + iy * bh[mm].view(n, 1, 1) / gh   # (n,P,gh)
                    x = b[mm, 0].view(n, 1, 1) + ph * bw[mm].view(n, 1, 1) + ix * bw[mm].view(n, 1, 1) / gw   # (n,P,gw)
                    vy = ~((y < -1.0) | (y > H)); vx = ~((x < -1.0) | (x > W))
                    y = y.clamp(min=0); x = x.clamp(min=0)
                    yl = y.floor().long(); xl = x.floor().long()
                    ycap = yl >= H - 1; xcap = xl >= W - 1
                    yl = torch.where(ycap, torch.full_like(yl, H - 1), yl); xl = torch.where(xcap, torch.full_like(xl, W - 1), xl)
                    yh = torch.where(ycap, yl, yl + 1); xh = torch.where(xcap, xl, xl + 1)
                    y = torch.where(ycap, yl.float(), y); x = torch.where(xcap, xl.float(), x)
                    ly = (y - yl.float()); lx = (x - xl.float())
                    hy = 1 - ly; hx = 1 - lx
                    wy = torch.stack((hy, ly), -1) * vy.unsqueeze(-1)          # (n,P,gh,2)
                    wx = torch.stack((hx, lx), -1) * vx.unsqueeze(-1)          # (n,P,gw,2)
                    yi = torch.stack((yl, yh), -1); xi = torch.stack((xl, xh), -1)
                    # separable accumulation: rows then columns
                    idx = (yi.view(n, P, gh, 2, 1, 1, 1) * W + xi.view(n, 1, 1, 1, P, gw, 2))   # (n,P,gh,2,P,gw,2)
                    wgt = wy.view(n, P, gh, 2, 1, 1, 1) * wx.view(n, 1, 1, 1, P, gw, 2)
                    vals = fl[idx.reshape(-1)].view(n, P, gh * 2, P, gw * 2, C)
                    acc = (vals * wgt.reshape(n, P, gh * 2, P, gw * 2, 1).to(dt)).sum(dim=(2, 4))   # (n,P,P,C)
                    out[sel[mm]] = (acc / max(gh * gw, 1)).permute(0, 3, 1, 2)
    return out


def box_head(h, x):
    for conv, norm in zip(h.convs, h.norms):
        x = F.relu(F.group_norm(F.conv2d(x, conv.weight.to(x.dtype), None, 1, 1), norm.num_groups, norm.weight.to(x.dtype),
                                norm.bias.to(x.dtype), norm.eps))
    flat = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    hid = F.relu(F.linear(flat, h.fc1_weight.to(x.dtype), h.fc1_bias.to(x.dtype)))
    return F.linear(hid, h.cls_weight.to(x.dtype), h.cls_bias.to(x.dtype)), F.linear(hid, h.box_weight.to(x.dtype), h.box_bias.to(x.dtype))


@torch.no_grad()
def forward(model, image_bgr, dtype=torch.float32, return_intermediates=False, proposals=None):
    """model: a CascadeRCNN on the CPU.  image (1,3,H,W) 0..255 BGR.  Returns (boxes, scores, classes)."""
    img_h, img_w = image_bgr.shape[2], image_bgr.shape[3]
    x = ((image_bgr.to(dtype) - model.pixel_mean.to(dtype)) / model.pixel_std.to(dtype))
    ph, pw = (32 - img_h % 32) % 32, (32 - img_w % 32) % 32
    if ph or pw:
        x = F.pad(x, (0, pw, 0, ph))
    feats = backbone(model.backbone, x)
    props = rpn(model.rpn, feats, img_h, img_w) if proposals is None else proposals.to(dtype)
    scales = [1.0 / s for s in (4, 8, 16, 32)]
    stage_scores, stage_out = [], []
    boxes = props
    for k in range(3):
        if k > 0:
            boxes = clip_boxes(boxes, img_h, img_w)
        pooled = roi_pool_fpn(feats[:4], boxes, scales)
        logits, deltas = box_head(model.heads[k], pooled)
        stage_scores.append(F.softmax(logits, dim=-1))
        stage_out.append((logits, deltas))
        boxes = apply_deltas(deltas, boxes, model.CASCADE_WEIGHTS[k])
    scores = (stage_scores[0] + stage_scores[1] + stage_scores[2]) * (1.0 / 3)
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    fb, fs = clip_boxes(boxes[valid], img_h, img_w), scores[valid][:, :-1]
    mask = fs > model.score_thresh
    inds = mask.nonzero()
    b, s = fb[inds[:, 0]], fs[mask]
    keep = batched_nms(b, s, inds[:, 1].to(torch.int32), model.nms_thresh)[: model.topk]
    result = (b[keep], s[keep], inds[keep, 1])
    if return_intermediates:
        return result, dict(feats=feats, proposals=props, stage_out=stage_out, boxes=boxes, scores=scores)
    return result


# ----------------------------------------------------------------------------------------------------------------------------
# Training losses (row a23 / config 5): what detectron2 computes when the reference calls Detectron2Det.loss
# (/root/reference/detnet/nn/detectron2_det/__init__.py:144-186 -> GeneralizedRCNN.forward in training mode), restated on the CPU
# ops above so that autograd yields float64 reference gradients.  Independent of waymo_2d_tracking_amd/detnet/nn/training.py.
# The fg / bg subsampling is random in detectron2 (subsample_labels); the parity tests make BOTH sides take the lowest indices.

def _iou_matrix(gt, boxes):
    """(G, N) pairwise IoU, detectron2.structures.pairwise_iou (0 where the boxes do not intersect)."""
    area_g = (gt[:, 2] - gt[:, 0]) * (gt[:, 3] - gt[:, 1])
    area_b = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    lt = torch.max(gt[:, None, :2], boxes[None, :, :2])
    rb = torch.min(gt[:, None, 2:], boxes[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_g[:, None] + area_b[None, :] - inter
    return torch.where(inter > 0, inter / union, torch.zeros_like(inter))


def _matcher(iou, thresholds, labels, low_quality):
    """detectron2.modeling.matcher.Matcher.__call__: per box the best gt and a label from the threshold bands."""
    vals, idx = iou.max(dim=0)
    lab = torch.full_like(idx, 1)
    bounds = [-float('inf')] + list(thresholds) + [float('inf')]
    for l, lo, hi in zip(labels, bounds[:-1], bounds[1:]):
        lab[(vals >= lo) & (vals < hi)] = l
    if low_quality:                                   # set_low_quality_matches_: boxes that realise a gt's best IoU
        best = iou.max(dim=1, keepdim=True)[0]
        lab[(iou == best).any(dim=0)] = 1
    return idx, lab


def _take_first(labels, num, frac, bg):
    pos = torch.nonzero((labels != -1) & (labels != bg)).flatten()
    neg = torch.nonzero(labels == bg).flatten()
    n_pos = min(int(num * frac), pos.numel())
    n_neg = min(num - n_pos, neg.numel())
    return pos[:n_pos], neg[:n_neg]


def _deltas(src, tgt, weights):
    """Box2BoxTransform.get_deltas."""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    scx, scy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tcx, tcy = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    return torch.stack((weights[0] * (tcx - scx) / sw, weights[1] * (tcy - scy) / sh, weights[2] * torch.log(tw / sw),
                        weights[3] * torch.log(th / sh)), dim=1)


class _GradScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None


def losses(model, image_bgr, gt_boxes, gt_classes, dtype=torch.float64, rpn_batch=256, rpn_pos=0.5, pre_nms=2000, post_nms=2000,
           roi_batch=512, roi_pos=0.25, proposals=None, return_intermediates=False):
    """The 8 detectron2 losses of the Cascade R-CNN for one image, differentiable (call .backward() on their sum).
    RPN: Matcher([0.3, 0.7], [0, -1, 1], low quality on), 256 anchors at 0.5 positive, BCE sum / 256 and L1 sum / 256
    (smooth_l1 beta 0).  Proposals: per-level top `pre_nms`, decode, clip, drop empty, per-level NMS 0.7, `post_nms` best.
    ROI heads: gt boxes appended, stage-1 Matcher([0.5]) + 512 at 0.25 foreground, stages 2 / 3 rematch at 0.6 / 0.7 without
    resampling on the previous stage's decoded boxes (clipped, empty removed), pooled features with gradient scale 1/3,
    mean cross-entropy and class-agnostic L1 on the foreground rows / number of rows."""
    gt_boxes = gt_boxes.to(dtype)
    num_classes = model.num_classes
    img_h, img_w = image_bgr.shape[2], image_bgr.shape[3]
    x = ((image_bgr.to(dtype) - model.pixel_mean.to(dtype)) / model.pixel_std.to(dtype))
    ph, pw = (32 - img_h % 32) % 32, (32 - img_w % 32) % 32
    if ph or pw:
        x = F.pad(x, (0, pw, 0, ph))
    block_record = [] if return_intermediates else None
    feats = backbone(model.backbone, x, block_record)
    r = model.rpn
    logits_l, deltas_l, anchors_l, boxes_l, scores_l, lvl_l = [], [], [], [], [], []
    for l, f in enumerate(feats):
        t = convbn(r.conv, f, relu=True)
        lg = conv1x1(r.objectness, t).permute(0, 2, 3, 1).reshape(-1)
        dl = conv1x1(r.deltas, t).permute(0, 2, 3, 1).reshape(-1, 4)
        an = r.anchors(l, f.shape[2], f.shape[3], f.device).to(dtype)
        logits_l.append(lg); deltas_l.append(dl); anchors_l.append(an)
        with torch.no_grad():
            k = min(pre_nms, lg.numel())
            top, idx = torch.topk(lg, k, sorted=True)
            boxes_l.append(apply_deltas(dl[idx], an[idx], (1.0, 1.0, 1.0, 1.0))); scores_l.append(top)
            lvl_l.append(torch.full((k,), l, dtype=torch.int32))
    logits, deltas, anchors = torch.cat(logits_l), torch.cat(deltas_l), torch.cat(anchors_l)
    out = {}
    with torch.no_grad():
        idx, lab = _matcher(_iou_matrix(gt_boxes, anchors), [0.3, 0.7], [0, -1, 1], True)
        pos, neg = _take_first(lab, rpn_batch, rpn_pos, 0)
        tgt_d = _deltas(anchors[pos], gt_boxes[idx[pos]], (1.0, 1.0, 1.0, 1.0))
        if proposals is None:
            b = clip_boxes(torch.cat(boxes_l), img_h, img_w)
            s, lv = torch.cat(scores_l), torch.cat(lvl_l)
            ok = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
            b, s, lv = b[ok], s[ok], lv[ok]
            proposals = b[batched_nms(b, s, lv, r.thr)[:post_nms]]
        else:
            proposals = proposals.to(dtype)
    sel = torch.cat((pos, neg))
    target = torch.cat((torch.ones(len(pos), dtype=dtype), torch.zeros(len(neg), dtype=dtype)))
    out['loss_rpn_cls'] = F.binary_cross_entropy_with_logits(logits[sel], target, reduction='sum') / rpn_batch
    out['loss_rpn_loc'] = (deltas[pos] - tgt_d).abs().sum() / rpn_batch
    inter = dict(proposals=proposals, rpn_pos=pos, rpn_neg=neg, blocks=block_record)
    with torch.no_grad():
        boxes = torch.cat((proposals, gt_boxes))
        idx, lab = _matcher(_iou_matrix(gt_boxes, boxes), [0.5], [0, 1], False)
        cls = torch.where(lab == 1, gt_classes[idx], torch.full_like(idx, num_classes))
        pos, neg = _take_first(cls, roi_batch, roi_pos, num_classes)
        boxes = boxes[torch.cat((pos, neg))]
    scales = [1.0 / s for s in (4, 8, 16, 32)]
    for k in range(3):
        with torch.no_grad():
            if k > 0:
                boxes = clip_boxes(boxes, img_h, img_w)
                boxes = boxes[((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)]
            idx, lab = _matcher(_iou_matrix(gt_boxes, boxes), [(0.5, 0.6, 0.7)[k]], [0, 1], False)
            cls = torch.where(lab == 1, gt_classes[idx], torch.full_like(idx, num_classes))
            tgt_boxes = gt_boxes[idx]
        inter['stage%d_boxes' % k] = boxes
        inter['stage%d_classes' % k] = cls
        pooled = _GradScale.apply(roi_pool_fpn(feats[:4], boxes, scales), 1.0 / 3)
        lg, dl = box_head(model.heads[k], pooled)
        fg = torch.nonzero(cls < num_classes).flatten()
        out['loss_cls_stage%d' % k] = F.cross_entropy(lg, cls, reduction='mean')
        out['loss_box_reg_stage%d' % k] = (dl[fg] - _deltas(boxes[fg], tgt_boxes[fg], model.CASCADE_WEIGHTS[k])).abs().sum() / max(len(cls), 1)
        boxes = apply_deltas(dl.detach(), boxes, model.CASCADE_WEIGHTS[k])
    return (out, inter) if return_intermediates else out
