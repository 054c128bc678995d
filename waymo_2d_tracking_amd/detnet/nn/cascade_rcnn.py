"""Cascade R-CNN X-152-32x8d-FPN (GN heads, deformable conv in res3-res5), inference graph for MI355X.

This is the detector the reference builds through detectron2's
``Misc/cascade_mask_rcnn_X_152_32x8d_FPN_IN5k_gn_dconv.yaml`` with the mask head off
(/root/reference/detnet/nn/detectron2_det/__init__.py:21-60); the layer shapes follow the module tree printed in
/root/reference/logs/12442/job.log:336-1221 and the op semantics SURVEY.md App. C (detectron2 is not vendored:
"parity unpinned" for the arithmetic; the STRUCTURE is pinned by tests/golden/x152_modules.json).  Weights: seeded random
init, or a detectron2 / reference checkpoint through weights.load_state_dict_detectron2 (FrozenBN folded, fc1 permuted).

MI355X-first layout: every activation is NHWC (torch.channels_last storage); FrozenBatchNorm is folded into the
producing op; the hot per-frame ops run in hand-written HIP kernels (detnet/nn/ops.py):
  * the box-head FC 12544 -> 1024 = the hand-written f32-MFMA GEMM (wd_gemm_nt_f32, split-K x2); the 1x1 convolutions
    (2/3 of the backbone FLOPs) are GEMMs on the NHWC matrix view through hipBLASLt (Conv1x1.USE_LIBRARY_GEMM: the library
    sustains 100-135 TFLOP/s on these shapes, ahead of the hand-written kernel), residual on the beta term,
  * the 47 deformable 3x3 convolutions = implicit GEMM with fused FrozenBN + ReLU (wd_deform_conv3x3_f32),
  * ROIPooler over the 4 FPN levels (wd_roi_pool_fpn_f32), RPN / box NMS (wd_nms_sorted_f32).
Dense 3x3 / 7x7 convolutions and GroupNorm stay on PyTorch-ROCm (MIOpen) - the "Python host carries the graph".
"""
import math
import os

import torch
from torch import nn
import torch.nn.functional as F

from . import ops

BLOCKS = (3, 8, 36, 3)            # job.log:354,402,528,1074
STAGE_CH = (256, 512, 1024, 2048)
GROUPS = 32
PIXEL_MEAN = (103.530, 116.280, 123.675)     # BGR, detectron2 defaults for the IN5k C2 weights
PIXEL_STD = (57.375, 57.120, 58.395)
SCALE_CLAMP = math.log(1000.0 / 16)


# training graph: FrozenBN affine / bias / residual / ReLU fused into the producing op (ops.LinearActFn, ops.DeformConvFn) instead of
# separate autograd-visible elementwise passes.  False = the round-3 graph (kept for the A/B parity test).
FUSED_TRAINING_EPILOGUES = True


# Inference: 1x1 and dense 3x3 convolutions on the split-operand kernel (csrc/det_gemm_split.hip: f32 operands carried exactly as three bf16
# planes, six cross terms on the bf16 matrix cores, f32 accumulation - error vs float64 below the f32 library GEMM's).  WD_SPLIT_GEMM=0 keeps
# every convolution on the exact-f32 library path (hipBLASLt / MIOpen) - the "exact_f32" line of bench.py and the A/B parity tests.
SPLIT_GEMM = os.environ.get('WD_SPLIT_GEMM', '1') != '0'


_SKIP = set(os.environ.get('WD_SPLIT_SKIP', '').split(','))      # experiments: 'cin-cout-stride-hasresidual' 1x1 shapes kept on the library
_INPLACE = os.environ.get('WD_SPLIT_INPLACE', '1') != '0'      # experiments: 0 = block outputs in fresh buffers instead of the residual's
SPLIT_PARTS = set(os.environ.get('WD_SPLIT_PARTS', 'conv1x1,conv3x3,head,fc,offset').split(','))     # experiments: which layer families use the kernel
# the N-thin GEMM of the 18-channel offset convolution goes to the split kernel from this many rows (res3 at full size; below, the library is as fast)
OFFSET_SPLIT_MIN_ROWS = int(os.environ.get('WD_OFFSET_SPLIT_MIN_ROWS', '30000'))


# Round 6, measured and OFF by default (profiles/r06_split_presplit.txt): inside a stage of >= SPLIT_PLANES_MIN_CH channels the block output travels as
# pre-split activation planes (ops.gemm_split_io) - conv3 writes planes (residual read from planes), the next conv1 pulls them with LDS-DMA; the stage's last
# block writes f32 for the next stage / the FPN.  Results are bit-identical to the f32 path.
SPLIT_PLANES = os.environ.get('WD_SPLIT_PLANES', '0') == '1'
SPLIT_PLANES_MIN_CH = int(os.environ.get('WD_SPLIT_PLANES_MIN_CH', '1024'))


class PlanesAct:
    """A block output that exists as activation planes only: (pixels, channels) planes + the NHWC geometry it stands for."""

    def __init__(self, planes, n, h, w, c):
        self.planes, self.n, self.h, self.w, self.c = planes, n, h, w, c


# Parity tests set this to a list: every Bottleneck.forward then appends (conv1 output, offsets or None, conv2 output, block output, stride) - the
# post-ReLU maps and sampling offsets from which oracle.detector_ref.block_decisions derives the block's discrete decisions (ReLU masks, bilinear cells).
DECISION_LOG = None


def _split_ok(cin, cout, part='conv1x1'):
    return SPLIT_GEMM and part in SPLIT_PARTS and cin % 64 == 0 and cout % 32 == 0


class _PackedSplit:
    """Per-module cache of a weight packed for the split-operand kernel (re-packed when the parameter changes or moves)."""

    def __init__(self):
        self.buf = None
        self.key = None

    def get(self, weight):
        key = (weight.device, weight._version, weight.data_ptr())
        if self.buf is None or self.key != key:
            self.buf = ops.split_pack_weight(weight)
            self.key = key
        return self.buf


def _msra(shape, gen, fan_out=True):
    w = torch.empty(shape)
    fan = shape[0] * shape[2] * shape[3] if fan_out else shape[1] * shape[2] * shape[3]
    return w.normal_(0, math.sqrt(2.0 / fan), generator=gen)


class Conv1x1(nn.Module):
    """1x1 conv (+ folded FrozenBN or bias) [+ residual] [+ ReLU] = one GEMM on the NHWC matrix view.

    Plain library GEMM (hipBLASLt through torch): on MI355X it sustains 100-135 TFLOP/s fp32 on these shapes
    (tools/gemm_bench.py), ahead of the hand-written wd_gemm_nt_f32 which is kept for the box-head FC.  The residual
    rides on the GEMM's beta term (addmm), bias+ReLU on the library epilogue where there is no residual."""
    USE_LIBRARY_GEMM = True
    FUSED_RESIDUAL = True

    def __init__(self, cin, cout, gen, bn_scale=1.0, bias=False):
        super().__init__()
        w = _msra((cout, cin, 1, 1), gen).view(cout, cin)
        # FrozenBN(weight=bn_scale, bias=0, mean=0, var=1, eps=1e-5) folded: w * scale, bias 0
        s = bn_scale / math.sqrt(1.0 + 1e-5) if not bias else 1.0
        self.weight = nn.Parameter(w * s, requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        self._split = _PackedSplit()

    def forward(self, x, relu=False, residual=None, stride=1):
        skip = '%d-%d-%d-%d' % (self.weight.shape[1], self.weight.shape[0], stride, 0 if residual is None else 1) in _SKIP
        if not skip and not torch.is_grad_enabled() and x.is_cuda and _split_ok(self.weight.shape[1], self.weight.shape[0]):
            # split-operand kernel: the strided shortcut reads its pixels in place (no gathered copy), the residual is added in the epilogue
            # and the block output lands in the residual's buffer (dead after this block), as on the library path
            pw = self._split.get(self.weight)
            cout = self.weight.shape[0]
            if stride != 1 or not x.is_contiguous(memory_format=torch.channels_last):
                return ops.conv_split(x, pw, cout, 1, stride, 0, self.bias, residual, relu)
            n, c, h, w = x.shape
            a = x.permute(0, 2, 3, 1).reshape(n * h * w, c)
            r = None
            if residual is not None:
                r = residual if residual.is_contiguous(memory_format=torch.channels_last) else residual.contiguous(memory_format=torch.channels_last)
                r = r.permute(0, 2, 3, 1).reshape(n * h * w, cout)
            if ops.gemm_split_ok(a, cout, self.bias, r):           # (alignment / size preconditions: otherwise the library path below)
                y = ops.gemm_split(a, pw, cout, self.bias, r, relu, out=r if _INPLACE else None)
                return y.view(n, h, w, cout).permute(0, 3, 1, 2)
        if stride != 1:
            x = x[:, :, ::stride, ::stride].contiguous(memory_format=torch.channels_last)
        n, c, h, w = x.shape
        a = x.permute(0, 2, 3, 1).reshape(n * h * w, c)
        r = None if residual is None else residual.permute(0, 2, 3, 1).reshape(n * h * w, -1)
        if torch.is_grad_enabled():                      # training fwd+bwd
            if FUSED_TRAINING_EPILOGUES:
                # one fused library call forward, one masking pass backward (ops.LinearActFn)
                y = ops.LinearActFn.apply(a, self.weight, self.bias, r, relu)
            else:                                        # plain autograd-visible library ops (the round-3 graph; parity tests compare the two)
                y = F.linear(a, self.weight, self.bias)
                if r is not None:
                    y = y + r
                if relu:
                    y = F.relu(y)
            return y.view(n, h, w, -1).permute(0, 3, 1, 2)
        if not self.USE_LIBRARY_GEMM:
            y = ops.gemm_nt(a, self.weight, self.bias, r, relu)
        elif r is None:
            if relu:
                y = torch._addmm_activation(self.bias, a, self.weight.t(), use_gelu=False)
            else:
                y = torch.addmm(self.bias, a, self.weight.t())
        else:
            # inference: the residual buffer is dead after this block (block input or a fresh shortcut output), so the
            # GEMM accumulates into it in place (beta = 1, C == D) - no copy of the residual into a new output
            if self.FUSED_RESIDUAL and a.is_contiguous():
                # residual on the beta term, folded-BN shift + ReLU on the library epilogue: one launch (wd_gemm_lt_f32)
                y = ops.gemm_lt(a, self.weight, self.bias, r if r.is_contiguous() else r.contiguous(), relu,
                                out=r if r.is_contiguous() else None)
            else:
                y = r.addmm_(a, self.weight.t()) if r.is_contiguous() else torch.addmm(r, a, self.weight.t())
                y = ops.bias_relu_(y, self.bias, relu)
        return y.view(n, h, w, -1).permute(0, 3, 1, 2)


class ConvBN(nn.Module):
    """Dense / grouped kxk conv with folded FrozenBN (MIOpen)."""

    def __init__(self, cin, cout, k, stride, pad, groups, gen, bias=False):
        super().__init__()
        w = _msra((cout, cin // groups, k, k), gen)
        s = 1.0 / math.sqrt(1.0 + 1e-5) if not bias else 1.0
        self.weight = nn.Parameter((w * s).contiguous(memory_format=torch.channels_last), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        self.stride, self.pad, self.groups = stride, pad, groups
        self._split = _PackedSplit()

    def forward(self, x, relu=False):
        if (not torch.is_grad_enabled() and x.is_cuda and self.groups == 1 and self.weight.shape[2] in (1, 3)
                and _split_ok(self.weight.shape[1], self.weight.shape[0], 'conv3x3')):
            # dense 3x3 (FPN output convs, RPN conv): implicit GEMM over (tap, channel) on the split-operand kernel, bias + ReLU fused
            return ops.conv_split(x, self._split.get(self.weight), self.weight.shape[0], self.weight.shape[2], self.stride, self.pad,
                                  self.bias, None, relu)
        if not torch.is_grad_enabled() and self.weight.shape[0] % 4 == 0:
            # inference: bias (+ReLU) in one vectorised NHWC pass behind the library conv instead of the framework's
            # broadcast bias add (+ separate ReLU)
            y = F.conv2d(x, self.weight, None, self.stride, self.pad, 1, self.groups)
            if y.is_contiguous(memory_format=torch.channels_last):
                n, c, h, w = y.shape
                ops.bias_relu_(y.permute(0, 2, 3, 1).reshape(n * h * w, c), self.bias, relu)
                return y
            y = y + self.bias.view(1, -1, 1, 1)
            return F.relu_(y) if relu else y
        if (torch.is_grad_enabled() and ops.SPLIT_TRAIN and x.is_cuda and self.groups == 1 and self.weight.shape[2] in (1, 3)
                and _split_ok(self.weight.shape[1], self.weight.shape[0])):
            return ops.ConvSplitFn.apply(x, self.weight, self.bias, self.stride, self.pad, relu)     # training: split-operand forward and backward-data
        y = F.conv2d(x, self.weight, self.bias, self.stride, self.pad, 1, self.groups)
        return F.relu_(y) if relu else y


class Bottleneck(nn.Module):
    """BottleneckBlock / DeformBottleneckBlock, STRIDE_IN_1X1=False (job.log:357-419)."""

    def __init__(self, cin, cout, stride, deform, gen, offset_std):
        super().__init__()
        width = cout                      # 32x8d: bottleneck width == stage output channels (job.log:363-376)
        self.stride, self.deform = stride, deform
        self.shortcut = Conv1x1(cin, cout, gen) if cin != cout else None
        self.conv1 = Conv1x1(cin, width, gen)
        if deform:
            self.conv2_offset = ConvBN(width, 18, 3, stride, 1, 1, gen, bias=True)
            # detectron2 zero-initialises the offset conv; random-init benchmarks scale it so that offsets are a
            # few pixels (a zero offset field would make the gather trivially regular)
            self.conv2_offset.weight.data.normal_(0, offset_std, generator=gen)
            w = _msra((width, width // GROUPS, 3, 3), gen)
            self.conv2_weight = nn.Parameter(w, requires_grad=False)
            self.conv2_scale = nn.Parameter(torch.full((width,), 1.0 / math.sqrt(1.0 + 1e-5)), requires_grad=False)
            self.conv2_bias = nn.Parameter(torch.zeros(width), requires_grad=False)
            self._packed = None
        else:
            # plain grouped 3x3 (res2): the same implicit-GEMM kernel without the sampling offsets
            self.conv2_weight = nn.Parameter(_msra((width, width // GROUPS, 3, 3), gen), requires_grad=False)
            self.conv2_scale = nn.Parameter(torch.full((width,), 1.0 / math.sqrt(1.0 + 1e-5)), requires_grad=False)
            self.conv2_bias = nn.Parameter(torch.zeros(width), requires_grad=False)
            self._packed = None
        self.conv3 = Conv1x1(width, cout, gen, bn_scale=0.25)   # keeps random-init activations bounded over 50 blocks

    def packed_weight(self):
        v = self.conv2_weight._version
        if self._packed is None or self._packed.device != self.conv2_weight.device or getattr(self, '_packed_v', -1) != v:
            self._packed = ops.deform_pack_weight(self.conv2_weight, GROUPS)     # re-packed after an optimizer step
            self._packed_v = v
        return self._packed

    def offset_conv(self, x, deform_table=False):
        w = self.conv2_offset.weight
        if getattr(self, '_off_w2', None) is None or self._off_w2.device != w.device or self._off_v != w._version:
            self._off_w2 = ops.tap_gemm_weight(w)
            self._off_v = w._version
        split = None
        if x.shape[0] * x.shape[2] * x.shape[3] >= OFFSET_SPLIT_MIN_ROWS and x.is_cuda and _split_ok(w.shape[1], 32, 'offset'):
            if getattr(self, '_off_split', None) is None or self._off_split_key != (w.device, w._version):
                w32 = ops.tap_gemm_weight(w, align=32)              # 162 -> 192 rows (zero rows): N % 32 == 0 for the split kernel
                self._off_split, self._off_split_key = (ops.split_pack_weight(w32), w32.shape[0]), (w.device, w._version)
            split = self._off_split
        return ops.conv3x3_few(x, self._off_w2, self.conv2_offset.bias, 18, 1, deform_table, split=split)

    def forward(self, x, emit_planes=False):
        if isinstance(x, PlanesAct) or emit_planes:
            return self._forward_planes(x, emit_planes)
        sc = x if self.shortcut is None else self.shortcut(x, stride=self.stride)
        out = self.conv1(x, relu=True)
        rec = DECISION_LOG
        if rec is not None:                              # parity tests: the block's discrete decisions (ReLU masks, bilinear cells), see DECISION_LOG
            out1, offset = out, None
        if self.deform and torch.is_grad_enabled():      # training: autograd Function around the HIP fwd / bwd kernels
            offset = self.conv2_offset(out)
            if FUSED_TRAINING_EPILOGUES and not (self.conv2_scale.requires_grad or self.conv2_bias.requires_grad):
                # FrozenBN affine + ReLU inside the HIP kernel's epilogue, like at inference; one masking pass in backward
                out = ops.DeformConvFn.apply(out, offset, self.conv2_weight, GROUPS, self.stride, 1, self.conv2_scale, self.conv2_bias, True)
            else:
                out = ops.DeformConvFn.apply(out, offset, self.conv2_weight, GROUPS, self.stride, 1)
                out = F.relu(out * self.conv2_scale.view(1, -1, 1, 1) + self.conv2_bias.view(1, -1, 1, 1))
        elif self.deform:
            # stride 1: library GEMM over the input pixels (162 columns) + tap shift-add kernel instead of a direct
            # 18-channel implicit GEMM (MIOpen pads N 18 -> 32 and adds the bias in a second pass)
            table = None
            if self.stride == 1 and self.conv2_weight.shape[1] in (16, 32):
                # 16 / 32 channels per group (res3 / res4): the persistent kernel's sampling table is built once per layer inside the
                # offset conv's gather launch and DMA'd per tile by the kernel; samples that leave the kernel's 14x14 patch are
                # fetched from global memory by their own lane one tap ahead (round 3: 2-px offset spread 93 us vs 144 us for the
                # per-tile fallback kernel, so the round-2 per-layer calibration + host sync is gone)
                offset, table = self.offset_conv(out, deform_table=True)
            else:
                offset = self.offset_conv(out) if self.stride == 1 else self.conv2_offset(out)
            out = ops.deform_conv3x3(out, offset, self.packed_weight(), GROUPS, self.stride, 1, self.conv2_scale,
                                     self.conv2_bias, relu=True, table=table)
        else:
            out = ops.deform_conv3x3(out, None, self.packed_weight(), GROUPS, self.stride, 1, self.conv2_scale,
                                     self.conv2_bias, relu=True)
        if rec is None:
            return self.conv3(out, relu=True, residual=sc)
        out3 = self.conv3(out, relu=True, residual=sc)
        # (clones: at inference / in frozen blocks the block output lands in the residual's buffer, which the next block overwrites)
        rec.append((out1.detach().clone(), None if offset is None else offset.detach().clone(), out.detach().clone(), out3.detach().clone(), self.stride))
        return out3


def _bottleneck_forward_planes(self, x, emit_planes):
    """Inference, round 6 (SPLIT_PLANES): the same block with the block input / output as activation planes.  x: f32 NCHW view (first block of a stage) or
    PlanesAct; returns PlanesAct when emit_planes else the f32 map.  conv1 and conv3 are wd_gemm_split_io calls; offset conv and deformable conv as ever."""
    in_planes = isinstance(x, PlanesAct)
    cout = self.conv3.weight.shape[0]
    if in_planes:
        assert self.shortcut is None and self.stride == 1
        n, h, w, cin = x.n, x.h, x.w, x.c
        m = n * h * w
        o1, _ = ops.gemm_split_io(m, self.conv1.weight.shape[0], cin, self.conv1._split.get(self.conv1.weight), a_planes=x.planes, bias=self.conv1.bias, relu=True)
        out = o1.view(n, h, w, -1).permute(0, 3, 1, 2)
        sc = None
    else:
        sc = x if self.shortcut is None else self.shortcut(x, stride=self.stride)
        out = self.conv1(x, relu=True)
    table = None
    if self.stride == 1 and self.conv2_weight.shape[1] in (16, 32):
        offset, table = self.offset_conv(out, deform_table=True)
    else:
        offset = self.offset_conv(out) if self.stride == 1 else self.conv2_offset(out)
    out = ops.deform_conv3x3(out, offset, self.packed_weight(), GROUPS, self.stride, 1, self.conv2_scale, self.conv2_bias, relu=True, table=table)
    n, c, h, w = out.shape
    m = n * h * w
    a = out.permute(0, 2, 3, 1).reshape(m, c)
    pw = self.conv3._split.get(self.conv3.weight)
    if in_planes:
        # block output into the residual's planes buffer (dead after this block), plus an f32 copy when the stage ends here
        y, yp = ops.gemm_split_io(m, cout, c, pw, a=a, bias=self.conv3.bias, residual_planes=x.planes, relu=True, want_out=not emit_planes,
                                  out_planes=x.planes if emit_planes else None, want_planes=False)
    else:
        r = sc if sc.is_contiguous(memory_format=torch.channels_last) else sc.contiguous(memory_format=torch.channels_last)
        r = r.permute(0, 2, 3, 1).reshape(m, cout)
        y, yp = ops.gemm_split_io(m, cout, c, pw, a=a, bias=self.conv3.bias, residual=r, relu=True, want_out=not emit_planes, want_planes=emit_planes)
    if emit_planes:
        return PlanesAct(yp, n, h, w, cout)
    return y.view(n, h, w, cout).permute(0, 3, 1, 2)


Bottleneck._forward_planes = _bottleneck_forward_planes


def _run_stage(stage, x):
    """One res stage; with SPLIT_PLANES (inference, split kernel on, wide stage) the block outputs inside the stage travel as activation planes."""
    blocks = list(stage)
    use = (SPLIT_PLANES and SPLIT_GEMM and not torch.is_grad_enabled() and x.is_cuda and DECISION_LOG is None and len(blocks) > 1
           and blocks[0].conv3.weight.shape[0] >= SPLIT_PLANES_MIN_CH and blocks[0].deform and 'conv1x1' in SPLIT_PARTS)
    if not use:
        return stage(x)
    for i, blk in enumerate(blocks):
        x = blk(x, emit_planes=i + 1 < len(blocks))
    return x


class ResNeXt152FPN(nn.Module):
    def __init__(self, gen, offset_std=0.01):
        super().__init__()
        self.stem = ConvBN(3, 64, 7, 2, 3, 1, gen)
        stages = []
        cin = 64
        for si, (nb, cout) in enumerate(zip(BLOCKS, STAGE_CH)):
            blocks = []
            for b in range(nb):
                stride = 2 if (b == 0 and si > 0) else 1
                blocks.append(Bottleneck(cin, cout, stride, si > 0, gen, offset_std))   # DEFORM_ON_PER_STAGE F,T,T,T
                cin = cout
            stages.append(nn.Sequential(*blocks))
        self.res2, self.res3, self.res4, self.res5 = stages
        self.lateral = nn.ModuleList([Conv1x1(c, 256, gen, bias=True) for c in STAGE_CH])
        self.output = nn.ModuleList([ConvBN(256, 256, 3, 1, 1, 1, gen, bias=True) for _ in STAGE_CH])

    def forward(self, x):
        return self.top_down(self.bottom_up(x))

    def bottom_up(self, x):
        """stem .. res5 (the part of a frame that runs the persistent deformable-conv kernels)"""
        with torch.no_grad():                                # FREEZE_AT = 2: stem and res2 never train (job.log:219)
            x = self.stem(x, relu=True)
            x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
            c2 = self.res2(x)
        c3 = _run_stage(self.res3, c2); c4 = _run_stage(self.res4, c3); c5 = _run_stage(self.res5, c4)
        return [c2, c3, c4, c5]

    def top_down(self, feats):
        c5 = feats[3]
        prev = self.lateral[3](c5)
        outs = [self.output[3](prev)]
        for i in (2, 1, 0):                                  # top-down pathway, nearest x2, fuse "sum"
            if prev.is_cuda and not (torch.is_grad_enabled() and prev.requires_grad) and prev.shape[1] % 4 == 0:
                top = ops.upsample2x_nearest(prev)               # same values, one HBM-speed kernel
            else:
                top = F.interpolate(prev, scale_factor=2.0, mode='nearest').contiguous(memory_format=torch.channels_last)
            prev = self.lateral[i](feats[i], residual=top)
            outs.insert(0, self.output[i](prev))
        p6 = F.max_pool2d(outs[3], kernel_size=1, stride=2, padding=0)     # LastLevelMaxPool
        return outs + [p6]                                    # p2..p6


def apply_deltas(deltas, boxes, weights):
    """detectron2 Box2BoxTransform.apply_deltas (class-agnostic (R,4) deltas)."""
    wx, wy, ww, wh = weights
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    dx = deltas[:, 0] / wx
    dy = deltas[:, 1] / wy
    dw = torch.clamp(deltas[:, 2] / ww, max=SCALE_CLAMP)
    dh = torch.clamp(deltas[:, 3] / wh, max=SCALE_CLAMP)
    pcx = dx * widths + ctr_x
    pcy = dy * heights + ctr_y
    pw = torch.exp(dw) * widths
    ph = torch.exp(dh) * heights
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=1)


def clip_boxes(boxes, h, w):
    x1 = boxes[:, 0].clamp(min=0, max=w); y1 = boxes[:, 1].clamp(min=0, max=h)
    x2 = boxes[:, 2].clamp(min=0, max=w); y2 = boxes[:, 3].clamp(min=0, max=h)
    return torch.stack((x1, y1, x2, y2), dim=1)


class RPN(nn.Module):
    """StandardRPNHead + DefaultAnchorGenerator + find_top_rpn_proposals (job.log:1126-1135;
    detectron2_det/configs/Base-RCNN-FPN.yaml:10-20: sizes 32..512, ratios .5/1/2, 1000 pre / 1000 post, NMS 0.7)."""
    SIZES = (32, 64, 128, 256, 512)
    RATIOS = (0.5, 1.0, 2.0)
    STRIDES = (4, 8, 16, 32, 64)

    def __init__(self, gen, pre_nms_topk=1000, post_nms_topk=1000, nms_thresh=0.7):
        super().__init__()
        self.conv = ConvBN(256, 256, 3, 1, 1, 1, gen, bias=True)
        self.conv.weight.data.normal_(0, 0.01, generator=gen)
        self.objectness = Conv1x1(256, 3, gen, bias=True)
        self.objectness.weight.data.normal_(0, 0.01, generator=gen)
        self.deltas = Conv1x1(256, 12, gen, bias=True)
        self.deltas.weight.data.normal_(0, 0.01, generator=gen)
        self.pre, self.post, self.thr = pre_nms_topk, post_nms_topk, nms_thresh
        self._anchors = {}

    def anchors(self, level, h, w, device):
        key = (level, h, w, device)
        if key not in self._anchors:
            size, stride = self.SIZES[level], self.STRIDES[level]
            cell = []
            for r in self.RATIOS:
                aw = math.sqrt(size * size / r)
                ah = r * aw
                cell.append([-aw / 2, -ah / 2, aw / 2, ah / 2])
            cell = torch.tensor(cell, dtype=torch.float32, device=device)
            sx = torch.arange(0, w * stride, stride, dtype=torch.float32, device=device)
            sy = torch.arange(0, h * stride, stride, dtype=torch.float32, device=device)
            yy, xx = torch.meshgrid(sy, sx, indexing='ij')
            shifts = torch.stack((xx, yy, xx, yy), dim=-1).reshape(-1, 1, 4)
            self._anchors[key] = (shifts + cell.view(1, -1, 4)).reshape(-1, 4)     # (H*W*A, 4)
        return self._anchors[key]

    def _level_ids(self, level, k, device):
        key = (level, k, device)
        if key not in self._anchors:
            self._anchors[key] = torch.full((k,), level, dtype=torch.int32, device=device)
        return self._anchors[key]

    def forward(self, feats, img_h, img_w):
        """feats p2..p6 (batch 1) -> (proposals (post_nms_topk, 4) sorted by objectness and zero-padded, count int32[1])."""
        logits_l, deltas_l, anchors_l = [], [], []
        for l, f in enumerate(feats):
            t = self.conv(f, relu=True)
            logits = self.objectness(t)                    # (1,3,H,W) channels_last == (H,W,A) order in memory
            deltas = self.deltas(t)                        # (1,12,H,W)
            logits_l.append(logits.permute(0, 2, 3, 1).reshape(-1))
            deltas_l.append(deltas.permute(0, 2, 3, 1).reshape(-1, 4))
            anchors_l.append(self.anchors(l, f.shape[2], f.shape[3], f.device))
        # per level: torch.topk(logits, pre) + apply_deltas(deltas[idx], anchors[idx]) + clip_boxes for all levels in 2-3 launches
        # (wd_rpn_topk_decode_f32).  find_top_rpn_proposals: drop empty boxes, per-level NMS, keep the `post` best - static
        # shapes, no host round trip: an empty box keeps its slot but can neither suppress (group -1) nor be selected (valid
        # 0); the result is always (post, 4) - unused rows are zero boxes - plus the device-side count of real proposals
        boxes, scores, lvls, ok = ops.rpn_topk_decode(logits_l, deltas_l, anchors_l, self.pre, img_h, img_w)
        # per-level NMS on the per-level sorted lists (the levels' suppression chains run in parallel workgroups), THEN the
        # global stable score sort with valid = non-empty & kept, and the first `post` of those: the same proposals as sorting
        # first and sweeping one 5000-box chain (batched_nms never lets levels interact)
        seg = [0]
        for lg in logits_l:
            seg.append(seg[-1] + min(self.pre, lg.numel()))
        keep = ops.nms_segmented(boxes, lvls, seg, self.thr)
        sb, ss, sg, sv, order = ops.sort_candidates(boxes, scores, lvls, ok, keep)
        props, count = ops.gather_kept(sv, sv, sb, ss, order, self.post)
        return props, count


class BoxHead(nn.Module):
    """FastRCNNConvFCHead (4 x conv3x3+GN32+ReLU, fc1 12544->1024) + FastRCNNOutputLayers (job.log:1146-1218)."""

    def __init__(self, num_classes, gen):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(256, 256, 3, 1, 1, bias=False) for _ in range(4)])
        self.norms = nn.ModuleList([nn.GroupNorm(32, 256) for _ in range(4)])
        for c in self.convs:
            c.weight.data.copy_(_msra((256, 256, 3, 3), gen))
        # fc1 weight is stored for the NHWC flatten order (ph, pw, c); a checkpoint in detectron2's (c, ph, pw) order
        # is permuted once by weights.load_state_dict_detectron2()
        self.fc1_weight = nn.Parameter(torch.empty(1024, 12544).normal_(0, math.sqrt(1.0 / 12544), generator=gen),
                                       requires_grad=False)
        self.fc1_bias = nn.Parameter(torch.zeros(1024), requires_grad=False)
        self.cls_weight = nn.Parameter(torch.empty(num_classes + 1, 1024).normal_(0, 0.01, generator=gen), requires_grad=False)
        self.cls_bias = nn.Parameter(torch.zeros(num_classes + 1), requires_grad=False)
        self.box_weight = nn.Parameter(torch.empty(4, 1024).normal_(0, 0.001, generator=gen), requires_grad=False)
        self.box_bias = nn.Parameter(torch.zeros(4), requires_grad=False)
        self._split = [_PackedSplit() for _ in range(4)]
        self._split_fc = _PackedSplit()

    def forward(self, x):
        train = torch.is_grad_enabled()
        for li, (conv, norm) in enumerate(zip(self.convs, self.norms)):
            if not train and x.is_cuda and _split_ok(256, 256, 'head'):
                x = ops.conv_split(x, self._split[li].get(conv.weight), 256, 3, 1, 1)
            elif train and ops.SPLIT_TRAIN and x.is_cuda and _split_ok(256, 256) and x.shape[0] > 0:
                x = ops.ConvSplitFn.apply(x, conv.weight, None, 1, 1, False)
            else:
                x = conv(x)
            if train:
                if FUSED_TRAINING_EPILOGUES and x.shape[2] * x.shape[3] <= 64 and x.shape[0] > 0:
                    x = ops.GroupNormReluFn.apply(x, norm.weight, norm.bias, norm.num_groups, norm.eps, True)
                else:
                    x = F.relu(norm(x))
                continue
            if not x.is_contiguous(memory_format=torch.channels_last):
                x = x.contiguous(memory_format=torch.channels_last)
            x = ops.groupnorm_relu_(x, norm.weight, norm.bias, norm.num_groups, norm.eps, True)
        r = x.shape[0]
        flat = x.permute(0, 2, 3, 1).reshape(r, -1)           # NHWC flatten, a view
        if train:
            h = F.relu(F.linear(flat, self.fc1_weight, self.fc1_bias))
        elif x.is_cuda and _split_ok(self.fc1_weight.shape[1], self.fc1_weight.shape[0], 'fc'):
            # 12544 -> 1024 on the split-operand kernel, K cut into slices (32 output tiles would leave 7/8 of the chip idle)
            h = ops.gemm_split(flat if flat.is_contiguous() else flat.contiguous(), self._split_fc.get(self.fc1_weight), self.fc1_weight.shape[0],
                               self.fc1_bias, None, True)
        else:
            h = ops.gemm_nt(flat, self.fc1_weight, self.fc1_bias, None, True)
        nc1 = self.cls_weight.shape[0]
        if not train and x.is_cuda and nc1 + 4 <= 32 and _split_ok(1024, 32, 'fc'):
            # class scores and box deltas as ONE split-operand GEMM on the concatenated (zero-padded to 32 rows) predictor weights: one launch
            # (+ the K-slice sum) instead of two skinny library GEMMs whose split-K kernels accumulate with float atomics - run-to-run
            # identical scores also with other kernels in flight (tests/test_gpu_e2e.py: two pipelines on two streams)
            key = (self.cls_weight._version, self.box_weight._version, self.cls_bias._version, self.box_bias._version, self.cls_weight.device)
            if getattr(self, '_pred_key', None) != key:
                wcat = torch.zeros(32, 1024, device=h.device)
                wcat[:nc1] = self.cls_weight.detach()
                wcat[nc1:nc1 + 4] = self.box_weight.detach()
                bcat = torch.zeros(32, device=h.device)
                bcat[:nc1] = self.cls_bias.detach()
                bcat[nc1:nc1 + 4] = self.box_bias.detach()
                self._pred_packed, self._pred_bias, self._pred_key = ops.split_pack_weight(wcat), bcat, key
            out = ops.gemm_split(h, self._pred_packed, 32, self._pred_bias, None, False)
            return out[:, :nc1], out[:, nc1:nc1 + 4]
        logits = F.linear(h, self.cls_weight, self.cls_bias)
        deltas = F.linear(h, self.box_weight, self.box_bias)
        return logits, deltas


class CascadeRCNN(nn.Module):
    CASCADE_WEIGHTS = ((10.0, 10.0, 5.0, 5.0), (20.0, 20.0, 10.0, 10.0), (30.0, 30.0, 15.0, 15.0))

    def __init__(self, num_classes=4, seed=0, score_thresh=0.01, nms_thresh=0.5, topk=100, offset_std=0.01):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.num_classes = num_classes
        self.backbone = ResNeXt152FPN(gen, offset_std)
        self.rpn = RPN(gen)
        self.heads = nn.ModuleList([BoxHead(num_classes, gen) for _ in range(3)])
        self.score_thresh, self.nms_thresh, self.topk = score_thresh, nms_thresh, topk   # detectron2_det/__init__.py:53
        self.register_buffer('pixel_mean', torch.tensor(PIXEL_MEAN).view(1, 3, 1, 1))
        self.register_buffer('pixel_std', torch.tensor(PIXEL_STD).view(1, 3, 1, 1))
        self.to(memory_format=torch.channels_last)
        # the grouped / deformable 3x3 weights are consumed by the HIP kernels only, which read OIHW: keep them contiguous (channels_last strides
        # cost a 1.2 MB copy per layer in forward and again in backward)
        for m in self.modules():
            if isinstance(m, Bottleneck):
                m.conv2_weight.data = m.conv2_weight.data.contiguous()

    def preprocess(self, image_bgr):
        """(1,3,H,W) float 0..255 BGR -> normalised, zero-padded to a multiple of 32 (size_divisibility), NHWC."""
        x = (image_bgr - self.pixel_mean) / self.pixel_std
        h, w = x.shape[2], x.shape[3]
        ph, pw = (32 - h % 32) % 32, (32 - w % 32) % 32
        if ph or pw:
            x = F.pad(x, (0, pw, 0, ph))
        return x.contiguous(memory_format=torch.channels_last)

    def forward(self, image_bgr, proposals=None, intermediates=None):
        with torch.no_grad():
            return self._forward_inference(image_bgr, proposals, intermediates)

    def _forward_inference(self, image_bgr, proposals=None, intermediates=None):
        """One image (1,3,H,W) -> (boxes (K,4) xyxy pixels, scores (K), classes (K) int64), K <= topk.
        `proposals` overrides the RPN output and `intermediates` (a dict) receives feature maps / stage outputs:
        both are test hooks."""
        img_h, img_w = image_bgr.shape[2], image_bgr.shape[3]
        return self.forward_normalized(self.preprocess(image_bgr), img_h, img_w, proposals, intermediates)

    def forward_normalized(self, x, img_h, img_w, proposals=None, intermediates=None):
        """Same, from the normalised / padded NHWC tensor that ops.preprocess (the fused HIP pre-processing kernel)
        or self.preprocess produce; (img_h, img_w) = the valid (unpadded) extent."""
        b, s, c, cnt = self.forward_padded(x, img_h, img_w, proposals, intermediates)
        k = int(cnt.item())                                   # the one host round trip of a frame (API boundary)
        return b[:k], s[:k], c[:k]

    def forward_padded(self, x, img_h, img_w, proposals=None, intermediates=None):
        """The whole graph with static shapes and no host synchronisation (capturable as one hipGraph): returns
        (boxes (topk, 4), scores (topk), classes (topk) int64, count int32[1]); rows >= count are padding."""
        return self.forward_padded_from(self.backbone.bottom_up(x), img_h, img_w, proposals, intermediates)

    def forward_padded_from(self, bottom_up_feats, img_h, img_w, proposals=None, intermediates=None):
        """forward_padded behind the bottom-up pathway: FPN top-down, RPN, cascade heads, static-shape tail."""
        feats = self.backbone.top_down(bottom_up_feats)
        from_rpn = proposals is None
        if from_rpn:
            proposals, n_prop = self.rpn(feats, img_h, img_w)
        else:
            n_prop = torch.full((1,), proposals.shape[0], dtype=torch.int32, device=proposals.device)
        scales = [1.0 / s for s in (4, 8, 16, 32)]
        stage_scores = []
        stage_out = []
        boxes = proposals
        # the RPN's proposal list has a static length (post_nms_topk rows, zero boxes behind the device-side count): the box heads
        # run on exactly those rows (measured: padding 1000 -> 1024 rows costs 1 % of the frame).  Injected proposal lists of
        # arbitrary length (tests) are rounded up to a multiple of 32 so that the library convolutions / GEMMs see few shapes -
        # with cudnn.benchmark every NEW shape costs a solver search
        n_roi = boxes.shape[0]
        n_pad = 0 if (from_rpn or not n_roi) else (-n_roi) % 32
        for k in range(3):
            rois = torch.zeros((n_roi + n_pad, 5), dtype=torch.float32, device=boxes.device)
            rois[:n_roi, 1:] = boxes
            pooled = ops.roi_pool_fpn(feats[:4], rois, scales, 7, 2, 4, 224.0)
            logits, deltas = self.heads[k](pooled)
            logits, deltas = logits[:n_roi], deltas[:n_roi]
            stage_out.append((logits, deltas))
            stage_scores.append(F.softmax(logits, dim=-1))
            # stages 0 / 1: the next stage starts with clip_boxes -> fused into the decode launch; the last stage's
            # boxes stay unclipped for the isfinite filter of inference()
            boxes = ops.decode_boxes(deltas, boxes, self.CASCADE_WEIGHTS[k], None, (img_h, img_w) if k < 2 else None)
        if intermediates is not None:
            scores = (stage_scores[0] + stage_scores[1] + stage_scores[2]) * (1.0 / 3)
            intermediates.update(feats=feats, proposals=proposals, stage_out=stage_out, boxes=boxes, scores=scores, n_proposals=n_prop)
        return self.inference(boxes, stage_scores, img_h, img_w, n_prop)

    def inference(self, boxes, stage_scores, img_h, img_w, n_valid=None):
        """detectron2 fast_rcnn_inference_single_image (class-agnostic boxes) with static shapes, in 4 launches: every (box,
        class) pair is a candidate slot; a slot is real when its box row is a real proposal, box and scores are finite and the
        class score (mean of the three stages' softmax) exceeds the threshold.  Real candidates keep their row-major (box,
        class) order under the stable score sort, exactly like `scores[mask]` / `mask.nonzero()` of the reference; padding
        slots sort behind them (score -1), never suppress (group -1) and are never selected.
        Returns (boxes (topk,4), scores (topk), classes (topk), count int32[1])."""
        s0, s1, s2 = stage_scores
        nc = s0.shape[1] - 1
        if s0.shape[0] * nc > 8192:                       # one-workgroup sort limit (1000 proposals x 4 Waymo classes = 4000)
            return self._inference_many_classes(boxes, (s0 + s1 + s2) * (1.0 / 3), img_h, img_w, n_valid)
        sb, ss, sg, sv, order = ops.box_candidates(boxes, s0, s1, s2, n_valid, self.score_thresh, img_h, img_w)
        keep = ops.nms_sorted_mask(sb, sg, self.nms_thresh)
        return ops.gather_kept(keep, sv, sb, ss, order, self.topk, nc)

    def _inference_many_classes(self, boxes, scores, img_h, img_w, n_valid=None):
        """Same contract for candidate counts beyond the fused kernel's limit (e.g. the 80-class COCO checkpoint): the
        candidates are built with torch ops, sort + NMS + selection through ops.nms_select."""
        r, nc = scores.shape[0], scores.shape[1] - 1
        row_ok = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
        if n_valid is not None:
            row_ok = row_ok & (torch.arange(r, device=boxes.device) < n_valid.to(torch.int64))
        boxes = clip_boxes(boxes, img_h, img_w)
        s = scores[:, :-1]
        real = (s > self.score_thresh) & row_ok.unsqueeze(1)                       # (r, nc)
        flat_s = torch.where(real, s, torch.full_like(s, -1.0)).reshape(-1)
        cls = torch.arange(nc, device=boxes.device, dtype=torch.int32).repeat(r)
        flat_c = torch.where(real.reshape(-1), cls, torch.full_like(cls, -1))
        flat_b = boxes.repeat_interleave(nc, dim=0)
        idx, count = ops.nms_select(flat_b, flat_s, flat_c, self.nms_thresh, self.topk, valid=real.reshape(-1))
        sel = idx.clamp(min=0)
        got = (idx >= 0)
        out_b = flat_b[sel] * got.unsqueeze(1).to(flat_b.dtype)
        out_s = flat_s[sel] * got.to(flat_s.dtype)
        out_c = (sel % nc) * got.to(torch.int64)
        return out_b, out_s, out_c, count
