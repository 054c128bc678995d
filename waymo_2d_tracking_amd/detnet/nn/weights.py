"""detectron2 / reference checkpoints <-> the MI355X-native Cascade R-CNN (cascade_rcnn.CascadeRCNN).

The reference builds its detector through detectron2 (`/root/reference/detnet/nn/detectron2_det/__init__.py:21-60`:
`build_model(cfg)` + `DetectionCheckpointer(model).load(cfg.MODEL.WEIGHTS)` for `pretrained='coco'`) and saves / loads
`{args, kwargs, state_dict}` files whose state dict carries detectron2's parameter names under the `model.` prefix of
`Detectron2Det.model` (`/root/reference/detnet/nn/__init__.py:47-63`).  The native graph stores the same weights in the
layout its kernels want: FrozenBatchNorm folded into the producing convolution (1x1 convs are (C_out, C_in) GEMM matrices),
deformable / grouped 3x3 weights raw with the BN affine kept as the kernel's fused scale / bias, `fc1` permuted from
detectron2's (c, ph, pw) flatten order to the NHWC (ph, pw, c) order.

`detectron2_layout()` lists every detectron2 parameter / buffer name and shape of the X-152-32x8d-FPN dconv model; the
test-suite checks it entry by entry against the module tree the reference printed (tests/golden/x152_modules.json, from
logs/12442/job.log:336-1221) - the structural pin of a detector whose arithmetic cannot be pinned (detectron2 is absent).
"""

import torch

from .cascade_rcnn import BLOCKS, GROUPS, STAGE_CH

BN_EPS = 1e-5
PREFIX = 'model.'


def detectron2_layout(num_classes=4):
    """[(name, shape, 'p' | 'b')] in detectron2 naming (GeneralizedRCNN of cascade_mask_rcnn_X_152_32x8d_FPN_IN5k_gn_dconv
    with MASK_ON False), without the `model.` prefix."""
    out = []

    def conv(name, cin, cout, k, groups=1, bias=False, bn=False):
        out.append((name + '.weight', [cout, cin // groups, k, k], 'p'))
        if bias:
            out.append((name + '.bias', [cout], 'p'))
        if bn:
            for b in ('weight', 'bias', 'running_mean', 'running_var'):
                out.append((name + '.norm.' + b, [cout], 'b'))

    for lvl, c in zip((2, 3, 4, 5), STAGE_CH):
        conv('backbone.fpn_lateral%d' % lvl, c, 256, 1, bias=True)
        conv('backbone.fpn_output%d' % lvl, 256, 256, 3, bias=True)
    conv('backbone.bottom_up.stem.conv1', 3, 64, 7, bn=True)
    cin = 64
    for si, (nb, cout) in enumerate(zip(BLOCKS, STAGE_CH)):
        for b in range(nb):
            base = 'backbone.bottom_up.res%d.%d' % (si + 2, b)
            if cin != cout:
                conv(base + '.shortcut', cin, cout, 1, bn=True)
            conv(base + '.conv1', cin, cout, 1, bn=True)
            if si > 0:
                conv(base + '.conv2_offset', cout, 18, 3, bias=True)
            conv(base + '.conv2', cout, cout, 3, groups=GROUPS, bn=True)
            conv(base + '.conv3', cout, cout, 1, bn=True)
            cin = cout
    conv('proposal_generator.rpn_head.conv', 256, 256, 3, bias=True)
    conv('proposal_generator.rpn_head.objectness_logits', 256, 3, 1, bias=True)
    conv('proposal_generator.rpn_head.anchor_deltas', 256, 12, 1, bias=True)
    for k in range(3):
        for j in range(1, 5):
            name = 'roi_heads.box_head.%d.conv%d' % (k, j)
            out.append((name + '.weight', [256, 256, 3, 3], 'p'))
            out.append((name + '.norm.weight', [256], 'p'))
            out.append((name + '.norm.bias', [256], 'p'))
        out.append(('roi_heads.box_head.%d.fc1.weight' % k, [1024, 256 * 7 * 7], 'p'))
        out.append(('roi_heads.box_head.%d.fc1.bias' % k, [1024], 'p'))
        out.append(('roi_heads.box_predictor.%d.cls_score.weight' % k, [num_classes + 1, 1024], 'p'))
        out.append(('roi_heads.box_predictor.%d.cls_score.bias' % k, [num_classes + 1], 'p'))
        out.append(('roi_heads.box_predictor.%d.bbox_pred.weight' % k, [4, 1024], 'p'))
        out.append(('roi_heads.box_predictor.%d.bbox_pred.bias' % k, [4], 'p'))
    return out


def _strip(sd):
    """Accept `model.`-prefixed (reference Detectron2Det files), bare (detectron2 .pth / .pkl `model` dicts) and DataParallel
    `module.` keys; numpy arrays (model-zoo pickles) become tensors."""
    out = {}
    for k, v in sd.items():
        for pre in ('module.', PREFIX):
            if k.startswith(pre):
                k = k[len(pre):]
        out[k] = v if torch.is_tensor(v) else torch.as_tensor(v)
    return out


def _bn(sd, name):
    """FrozenBatchNorm2d (weight, bias, running_mean, running_var) -> per-channel (scale, shift)."""
    w, b = sd[name + '.weight'].double(), sd[name + '.bias'].double()
    mean, var = sd[name + '.running_mean'].double(), sd[name + '.running_var'].double()
    scale = w / torch.sqrt(var + BN_EPS)
    return scale, b - mean * scale


def load_state_dict_detectron2(net, state_dict, strict=True):
    """Fill a cascade_rcnn.CascadeRCNN from a detectron2-named state dict.  Returns (missing, unexpected) name lists;
    with strict=True anything missing, unexpected (other than detectron2's own anchor / pixel buffers) or of the wrong shape
    raises."""
    sd = _strip(state_dict)
    want = {n: s for n, s, _ in detectron2_layout(net.num_classes)}
    missing = [n for n in want if n not in sd]
    ignorable = ('proposal_generator.anchor_generator.', 'pixel_mean', 'pixel_std')
    unexpected = [n for n in sd if n not in want and not n.startswith(ignorable)]
    bad = [(n, list(sd[n].shape), want[n]) for n in want if n in sd and list(sd[n].shape) != want[n]]
    if bad:
        raise ValueError('detectron2 checkpoint does not fit the X-152-32x8d-FPN dconv graph: %s' % bad[:4])
    if strict and (missing or unexpected):
        raise KeyError('detectron2 checkpoint: missing %s, unexpected %s' % (missing[:6], unexpected[:6]))
    has = lambda *names: all(n in sd for n in names)

    def put(param, value):
        with torch.no_grad():
            param.copy_(value.to(param.dtype).reshape(param.shape))

    def copy(param, name, transform=None):   # one tensor, skipped when absent (strict=False)
        if name in sd:
            put(param, transform(sd[name]) if transform else sd[name])

    bn_names = lambda name: [name + '.' + b for b in ('weight', 'bias', 'running_mean', 'running_var')]

    def fold1x1(mod, name):                  # Conv2d 1x1 + FrozenBN -> GEMM matrix (C_out, C_in) + bias
        if not has(name + '.weight', *bn_names(name + '.norm')):
            return
        scale, shift = _bn(sd, name + '.norm')
        put(mod.weight, sd[name + '.weight'].double().flatten(1) * scale[:, None])
        put(mod.bias, shift)

    def plain(mod, name, flatten=False):     # conv / linear with its own bias, no norm
        copy(mod.weight, name + '.weight', (lambda w: w.flatten(1)) if flatten else None)
        copy(mod.bias, name + '.bias')

    bb = net.backbone
    stem = 'backbone.bottom_up.stem.conv1'
    if has(stem + '.weight', *bn_names(stem + '.norm')):
        scale, shift = _bn(sd, stem + '.norm')
        put(bb.stem.weight, sd[stem + '.weight'].double() * scale[:, None, None, None])
        put(bb.stem.bias, shift)
    for si, stage in enumerate((bb.res2, bb.res3, bb.res4, bb.res5)):
        for b, blk in enumerate(stage):
            base = 'backbone.bottom_up.res%d.%d' % (si + 2, b)
            if blk.shortcut is not None:
                fold1x1(blk.shortcut, base + '.shortcut')
            fold1x1(blk.conv1, base + '.conv1')
            fold1x1(blk.conv3, base + '.conv3')
            copy(blk.conv2_weight, base + '.conv2.weight')     # grouped / deformable 3x3: raw weight ...
            if has(*bn_names(base + '.conv2.norm')):           # ... the BN affine is the kernel's fused epilogue
                scale, shift = _bn(sd, base + '.conv2.norm')
                put(blk.conv2_scale, scale); put(blk.conv2_bias, shift)
            blk._packed = None
            if blk.deform:
                plain(blk.conv2_offset, base + '.conv2_offset')
                blk._off_w2 = None
    for i, lvl in enumerate((2, 3, 4, 5)):
        plain(bb.lateral[i], 'backbone.fpn_lateral%d' % lvl, flatten=True)
        plain(bb.output[i], 'backbone.fpn_output%d' % lvl)
    plain(net.rpn.conv, 'proposal_generator.rpn_head.conv')
    plain(net.rpn.objectness, 'proposal_generator.rpn_head.objectness_logits', flatten=True)
    plain(net.rpn.deltas, 'proposal_generator.rpn_head.anchor_deltas', flatten=True)
    for k, head in enumerate(net.heads):
        for j in range(4):
            name = 'roi_heads.box_head.%d.conv%d' % (k, j + 1)
            copy(head.convs[j].weight, name + '.weight')
            copy(head.norms[j].weight, name + '.norm.weight')
            copy(head.norms[j].bias, name + '.norm.bias')
        name = 'roi_heads.box_head.%d.fc1' % k
        # detectron2 flattens (R, 256, 7, 7) as (c, ph, pw); the NHWC head flattens (ph, pw, c)
        copy(head.fc1_weight, name + '.weight', lambda w: w.reshape(1024, 256, 7, 7).permute(0, 2, 3, 1).reshape(1024, 12544))
        copy(head.fc1_bias, name + '.bias')
        name = 'roi_heads.box_predictor.%d' % k
        copy(head.cls_weight, name + '.cls_score.weight'); copy(head.cls_bias, name + '.cls_score.bias')
        copy(head.box_weight, name + '.bbox_pred.weight'); copy(head.box_bias, name + '.bbox_pred.bias')
    return missing, unexpected


def export_state_dict_detectron2(net, prefix=PREFIX):
    """The inverse mapping (for save() in the reference's file format): folded convolutions are written with an identity
    FrozenBatchNorm (weight 1, mean 0, var 1 - eps, bias = folded shift), so load_state_dict_detectron2 reproduces the same
    tensors."""
    sd = {}
    one = lambda n: torch.ones(n)
    ident_var = lambda n: torch.full((n,), 1.0 - BN_EPS)

    def unfold(name, weight4d, bias):
        n = weight4d.shape[0]
        sd[name + '.weight'] = weight4d.detach().cpu().float().contiguous()
        sd[name + '.norm.weight'] = one(n); sd[name + '.norm.bias'] = bias.detach().cpu().float().clone()
        sd[name + '.norm.running_mean'] = torch.zeros(n); sd[name + '.norm.running_var'] = ident_var(n)

    def plain(name, weight4d, bias):
        sd[name + '.weight'] = weight4d.detach().cpu().float().contiguous()
        sd[name + '.bias'] = bias.detach().cpu().float().clone()

    bb = net.backbone
    unfold('backbone.bottom_up.stem.conv1', bb.stem.weight, bb.stem.bias)
    for si, stage in enumerate((bb.res2, bb.res3, bb.res4, bb.res5)):
        for b, blk in enumerate(stage):
            base = 'backbone.bottom_up.res%d.%d' % (si + 2, b)
            if blk.shortcut is not None:
                unfold(base + '.shortcut', blk.shortcut.weight[:, :, None, None], blk.shortcut.bias)
            unfold(base + '.conv1', blk.conv1.weight[:, :, None, None], blk.conv1.bias)
            unfold(base + '.conv3', blk.conv3.weight[:, :, None, None], blk.conv3.bias)
            n = blk.conv2_weight.shape[0]
            sd[base + '.conv2.weight'] = blk.conv2_weight.detach().cpu().float().contiguous()
            # scale = w / sqrt(var + eps) with var = 1 - eps  ->  w = scale
            sd[base + '.conv2.norm.weight'] = blk.conv2_scale.detach().cpu().float().clone()
            sd[base + '.conv2.norm.bias'] = blk.conv2_bias.detach().cpu().float().clone()
            sd[base + '.conv2.norm.running_mean'] = torch.zeros(n); sd[base + '.conv2.norm.running_var'] = ident_var(n)
            if blk.deform:
                plain(base + '.conv2_offset', blk.conv2_offset.weight, blk.conv2_offset.bias)
    for i, lvl in enumerate((2, 3, 4, 5)):
        plain('backbone.fpn_lateral%d' % lvl, bb.lateral[i].weight[:, :, None, None], bb.lateral[i].bias)
        plain('backbone.fpn_output%d' % lvl, bb.output[i].weight, bb.output[i].bias)
    plain('proposal_generator.rpn_head.conv', net.rpn.conv.weight, net.rpn.conv.bias)
    plain('proposal_generator.rpn_head.objectness_logits', net.rpn.objectness.weight[:, :, None, None], net.rpn.objectness.bias)
    plain('proposal_generator.rpn_head.anchor_deltas', net.rpn.deltas.weight[:, :, None, None], net.rpn.deltas.bias)
    for k, head in enumerate(net.heads):
        for j in range(4):
            name = 'roi_heads.box_head.%d.conv%d' % (k, j + 1)
            sd[name + '.weight'] = head.convs[j].weight.detach().cpu().float().contiguous()
            sd[name + '.norm.weight'] = head.norms[j].weight.detach().cpu().float().clone()
            sd[name + '.norm.bias'] = head.norms[j].bias.detach().cpu().float().clone()
        w = head.fc1_weight.detach().cpu().float().reshape(1024, 7, 7, 256).permute(0, 3, 1, 2).reshape(1024, 12544)
        sd['roi_heads.box_head.%d.fc1.weight' % k] = w.contiguous()
        sd['roi_heads.box_head.%d.fc1.bias' % k] = head.fc1_bias.detach().cpu().float().clone()
        name = 'roi_heads.box_predictor.%d' % k
        sd[name + '.cls_score.weight'] = head.cls_weight.detach().cpu().float().clone()
        sd[name + '.cls_score.bias'] = head.cls_bias.detach().cpu().float().clone()
        sd[name + '.bbox_pred.weight'] = head.box_weight.detach().cpu().float().clone()
        sd[name + '.bbox_pred.bias'] = head.box_bias.detach().cpu().float().clone()
    return {prefix + k: v for k, v in sd.items()}


def is_detectron2_state_dict(sd):
    return any(k.startswith((PREFIX + 'backbone.bottom_up.', 'backbone.bottom_up.', 'module.model.backbone.bottom_up.')) for k in sd)


def load_checkpoint_file(path):
    """detectron2 model-zoo `.pkl` ({'model': {name: ndarray}, ...}), detectron2 / torch `.pth` ({'model': state_dict} or a bare
    state dict) or a reference file ({'args', 'kwargs', 'state_dict'}) -> state dict."""
    path = str(path)
    if path.endswith('.pkl'):
        import pickle
        with open(path, 'rb') as f:
            data = pickle.load(f, encoding='latin1')
    else:
        data = torch.load(path, map_location='cpu', weights_only=False)
    if isinstance(data, dict):
        for key in ('state_dict', 'model'):
            if key in data and isinstance(data[key], dict):
                return data[key]
    return data
