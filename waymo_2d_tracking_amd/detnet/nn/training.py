"""Training fwd+bwd of the Cascade R-CNN graph - the losses detectron2 computes when the
reference calls ``Detectron2Det.loss`` (/root/reference/detnet/nn/detectron2_det/__init__.py:144-186) restated in
PyTorch (RPN objectness/box losses, three cascade stages with IoU 0.5/0.6/0.7, class-agnostic box regression,
1/3 gradient scaling of the pooled features).  The custom ops run through their HIP forward and backward kernels
(ops.DeformConvFn, ops.RoiPoolFpnFn); dense convs / GEMMs use the library kernels through autograd.
Trainable = everything except stem + res2 (FREEZE_AT 2) and the FrozenBN statistics (folded constants).
"""
import torch
import torch.nn.functional as F

from . import ops
from .cascade_rcnn import apply_deltas, clip_boxes


def pairwise_iou(a, b):
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[None, :, 2:]) - torch.max(a[:, None, :2], b[None, :, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return torch.where(inter > 0, inter / (area_a[:, None] + area_b[None, :] - inter), torch.zeros_like(inter))


def match(iou, thresholds, labels, allow_low_quality):
    """detectron2 Matcher: iou (M gt, N boxes) -> (matched gt index (N), label (N))."""
    n = iou.shape[1]
    if iou.shape[0] == 0:
        return iou.new_zeros(n, dtype=torch.int64), iou.new_full((n,), labels[0], dtype=torch.int8)
    vals, idx = iou.max(dim=0)
    out = vals.new_full((n,), 1, dtype=torch.int8)
    th = [-float('inf')] + list(thresholds) + [float('inf')]
    for l, lo, hi in zip(labels, th[:-1], th[1:]):
        out[(vals >= lo) & (vals < hi)] = l
    if allow_low_quality:
        best, _ = iou.max(dim=1)
        out[(iou == best[:, None]).nonzero()[:, 1]] = 1
    return idx, out


# Per-arch training configuration.  The reference resolves the arch name first against its local configs
# (detectron2_det/__init__.py:20-27) and falls through to detectron2's model-zoo yaml when there is none - which is the case
# for the X-152 cascade config: its values are Base-RCNN-FPN.yaml (PRE_NMS_TOPK_TRAIN 2000 per level, POST_NMS_TOPK_TRAIN 1000,
# RPN batch 256 / 0.5, ROI batch 512 / 0.25) overridden by the zoo file's own `RPN: POST_NMS_TOPK_TRAIN: 2000` (every cascade
# config of the zoo carries that override; the yaml is not in the reference tree, so this value is recalled, not read).
TRAIN_CONFIG = {
    'Misc/cascade_mask_rcnn_X_152_32x8d_FPN_IN5k_gn_dconv.yaml': dict(pre_nms=2000, post_nms=2000, rpn_batch=256, rpn_pos=0.5,
                                                                     roi_batch=512, roi_pos=0.25),
}
DEFAULT_ARCH = 'Misc/cascade_mask_rcnn_X_152_32x8d_FPN_IN5k_gn_dconv.yaml'


def random_choice(candidates, n):
    """detectron2 subsample_labels: a random subset (torch.randperm on the candidates' device)."""
    return candidates[torch.randperm(candidates.numel(), device=candidates.device)[:n]]


def first_choice(candidates, n):
    """Deterministic stand-in for the parity tests: the n lowest indices (candidates come out of nonzero() sorted)."""
    return candidates[:n]


def subsample(labels, num, pos_frac, bg, choose=random_choice):
    pos = ((labels != -1) & (labels != bg)).nonzero().flatten()
    neg = (labels == bg).nonzero().flatten()
    n_pos = min(int(num * pos_frac), pos.numel())
    n_neg = min(num - n_pos, neg.numel())
    return choose(pos, n_pos), choose(neg, n_neg)


def get_deltas(src, tgt, weights):
    wx, wy, ww, wh = weights
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=1)


class _ScaleGradient(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


def rpn_losses(rpn, feats, gt_boxes, img_h, img_w, batch_per_image=256, pos_frac=0.5, pre_nms=2000, post_nms=2000,
               choose=random_choice):
    """RPN head on p2..p6 -> (losses, proposals for the ROI heads (detached)).  Sizes: TRAIN_CONFIG."""
    logits_l, deltas_l, anchors_l = [], [], []
    boxes_l, scores_l, lvl_l = [], [], []
    for l, f in enumerate(feats):
        t = rpn.conv(f, relu=True)
        lg = rpn.objectness(t).permute(0, 2, 3, 1).reshape(-1)
        dl = rpn.deltas(t).permute(0, 2, 3, 1).reshape(-1, 4)
        an = rpn.anchors(l, f.shape[2], f.shape[3], f.device)
        logits_l.append(lg); deltas_l.append(dl); anchors_l.append(an)
        with torch.no_grad():
            k = min(pre_nms, lg.numel())
            top, idx = torch.topk(lg, k, sorted=True)
            boxes_l.append(apply_deltas(dl[idx], an[idx], (1.0, 1.0, 1.0, 1.0))); scores_l.append(top)
            lvl_l.append(torch.full((k,), l, dtype=torch.int32, device=f.device))
    logits, deltas, anchors = torch.cat(logits_l), torch.cat(deltas_l), torch.cat(anchors_l)
    with torch.no_grad():
        idx, lab = match(pairwise_iou(gt_boxes, anchors), [0.3, 0.7], [0, -1, 1], True)
        pos, neg = subsample(lab, batch_per_image, pos_frac, 0, choose)
        boxes = clip_boxes(torch.cat(boxes_l), img_h, img_w)
        scores, lvls = torch.cat(scores_l), torch.cat(lvl_l)
        ok = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
        # find_top_rpn_proposals: drop empty boxes, batched_nms over the levels, the post_nms best.  As at inference the per-level lists are
        # already sorted (topk), so the levels' suppression chains run as parallel workgroups of the column-sweep kernel (one ~10 000-box chain
        # took the 0.9 ms row sweep); an empty box keeps its slot with group -1: it can neither suppress nor be kept
        seg = [0]
        for b_l in boxes_l:
            seg.append(seg[-1] + b_l.shape[0])
        keep = ops.nms_segmented(boxes, torch.where(ok, lvls, torch.full_like(lvls, -1)), seg, rpn.thr).bool() & ok
        kept = torch.nonzero(keep).squeeze(1)
        order = torch.argsort(scores[kept], descending=True, stable=True)
        proposals = boxes[kept[order[:post_nms]]]
        gt_d = get_deltas(anchors[pos], gt_boxes[idx[pos]], (1.0, 1.0, 1.0, 1.0)) if pos.numel() else deltas.new_zeros((0, 4))
    sel = torch.cat((pos, neg))
    tgt = torch.cat((torch.ones_like(pos, dtype=torch.float32), torch.zeros_like(neg, dtype=torch.float32)))
    norm = 1.0 / batch_per_image
    loss_cls = F.binary_cross_entropy_with_logits(logits[sel], tgt, reduction='sum') * norm
    loss_loc = (deltas[pos] - gt_d).abs().sum() * norm                     # smooth L1 with beta = 0
    return {'loss_rpn_cls': loss_cls, 'loss_rpn_loc': loss_loc}, proposals


def roi_losses(model, feats, proposals, gt_boxes, gt_classes, img_h, img_w, batch_per_image=512, pos_frac=0.25,
               choose=random_choice):
    """CascadeROIHeads._forward_box in training mode."""
    num_classes = model.num_classes
    scales = [1.0 / s for s in (4, 8, 16, 32)]
    losses = {}
    with torch.no_grad():
        boxes = torch.cat((proposals, gt_boxes))                           # proposal_append_gt
        idx, lab = match(pairwise_iou(gt_boxes, boxes), [0.5], [0, 1], False)
        cls = torch.where(lab == 1, gt_classes[idx] if gt_classes.numel() else idx, torch.full_like(idx, num_classes))
        pos, neg = subsample(cls, batch_per_image, pos_frac, num_classes, choose)
        keep = torch.cat((pos, neg))
        boxes = boxes[keep]
    for k in range(3):
        with torch.no_grad():
            if k > 0:
                boxes = clip_boxes(boxes, img_h, img_w)
                boxes = boxes[((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)]
            idx, lab = match(pairwise_iou(gt_boxes, boxes), [(0.5, 0.6, 0.7)[k]], [0, 1], False)
            if gt_classes.numel():
                cls = torch.where(lab == 1, gt_classes[idx], torch.full_like(idx, num_classes))
                tgt_boxes = gt_boxes[idx]
            else:
                cls = torch.full_like(idx, num_classes)
                tgt_boxes = boxes
            rois = torch.cat((torch.zeros((boxes.shape[0], 1), device=boxes.device), boxes), dim=1)
        pooled = ops.RoiPoolFpnFn.apply(rois, scales, 7, 2, 4, 224.0, *feats[:4])
        pooled = _ScaleGradient.apply(pooled, 1.0 / 3)
        logits, deltas = model.heads[k](pooled)
        fg = (cls < num_classes).nonzero().flatten()
        gt_d = get_deltas(boxes[fg], tgt_boxes[fg], model.CASCADE_WEIGHTS[k])
        losses['loss_cls_stage%d' % k] = F.cross_entropy(logits, cls, reduction='mean')
        losses['loss_box_reg_stage%d' % k] = (deltas[fg] - gt_d).abs().sum() / max(cls.numel(), 1)
        boxes = apply_deltas(deltas.detach(), boxes, model.CASCADE_WEIGHTS[k])
    return losses


def losses(model, image_bgr, gt_boxes, gt_classes, arch=DEFAULT_ARCH, choose=random_choice, config=None, proposals=None):
    """One image (1,3,H,W) BGR 0..255, gt_boxes (G,4) xyxy pixels, gt_classes (G) in [0, num_classes).
    `choose(candidates, n)` is the fg / bg sampler (random like detectron2's; the parity tests inject `first_choice`);
    `config` overrides TRAIN_CONFIG[arch] (tests shrink the proposal counts for small images); `proposals` (R, 4) replaces
    the RPN's own proposals as input of the ROI heads (detectron2's precomputed-proposals mode; parity tests)."""
    assert torch.is_grad_enabled()
    cfg = dict(TRAIN_CONFIG[arch])
    cfg.update(config or {})
    img_h, img_w = image_bgr.shape[2], image_bgr.shape[3]
    feats = model.backbone(model.preprocess(image_bgr))
    out, own = rpn_losses(model.rpn, feats, gt_boxes, img_h, img_w, cfg['rpn_batch'], cfg['rpn_pos'], cfg['pre_nms'],
                          cfg['post_nms'], choose)
    proposals = own if proposals is None else proposals
    out.update(roi_losses(model, feats, proposals, gt_boxes, gt_classes, img_h, img_w, cfg['roi_batch'], cfg['roi_pos'], choose))
    return out


def set_trainable(model):
    """requires_grad like the reference's run: backbone frozen up to res2, FrozenBN constants fixed."""
    for p in model.parameters():
        p.requires_grad_(False)
    bb = model.backbone
    for stage in (bb.res3, bb.res4, bb.res5):
        for blk in stage:
            for m in (blk.shortcut, blk.conv1, blk.conv3):
                if m is not None:
                    m.weight.requires_grad_(True)
            blk.conv2_weight.requires_grad_(True)
            blk.conv2_offset.weight.requires_grad_(True)
            blk.conv2_offset.bias.requires_grad_(True)
    for m in list(bb.lateral) + list(bb.output) + [model.rpn.conv, model.rpn.objectness, model.rpn.deltas]:
        m.weight.requires_grad_(True)
        m.bias.requires_grad_(True)
    for h in model.heads:
        for p in h.parameters():
            p.requires_grad_(True)
    return [p for p in model.parameters() if p.requires_grad]
