"""Detector custom ops: torch-tensor front-ends of the HIP kernels declared in include/waymodet.h.

torch is plumbing here (device memory + current stream); every function enqueues a hand-written gfx950 kernel
of libwaymotrack.so on the current stream and raises if the library or a GPU is missing (no fallback).
Feature maps are (N, C, H, W) tensors in ``torch.channels_last`` storage, i.e. NHWC in memory.
"""
import ctypes as C
import os

import torch

from ... import _lib


# bench.py sets this to a list to time every deformable-conv launch with HIP events recorded on the launch stream
# (tag, flops, start_event, end_event); None = no instrumentation.
EVENT_LOG = None


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_RAW_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    """The current HIP stream of the current device as a void*.  The raw accessors skip the torch.cuda.Stream object that
    torch.cuda.current_stream() builds per call (~2 us of host time on each of the ~1500 launches of a training step)."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return C.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _nhwc(x):
    if x.dtype != torch.float32:
        x = x.float()
    return x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)


def roi_pool_fpn(feats, rois, scales, pooled=7, min_level=2, canonical_level=4, canonical_size=224.0):
    """detectron2 ROIPooler (ROIAlign aligned=True, sampling_ratio=0) over FPN levels.
    feats: list of (N,C,H_l,W_l); rois (R,5) [batch,x1,y1,x2,y2] -> (R,C,pooled,pooled) channels_last."""
    feats = [_nhwc(f) for f in feats]
    rois = rois.contiguous().float()
    n, c = feats[0].shape[0], feats[0].shape[1]
    r = rois.shape[0]
    out = torch.empty((r, c, pooled, pooled), dtype=torch.float32, device=rois.device, memory_format=torch.channels_last)
    if r == 0:
        return out
    nl = len(feats)
    ptrs = (C.c_void_p * nl)(*[f.data_ptr() for f in feats])
    hs = (C.c_int32 * nl)(*[f.shape[2] for f in feats])
    ws = (C.c_int32 * nl)(*[f.shape[3] for f in feats])
    sc = (C.c_float * nl)(*[float(s) for s in scales])
    _lib.check(_lib.lib().wd_roi_pool_fpn_f32(ptrs, hs, ws, sc, C.c_int(nl), C.c_int(c), C.c_int(n), _p(rois), C.c_int(r),
                                              C.c_int(pooled), C.c_int(min_level), C.c_int(canonical_level),
                                              C.c_float(canonical_size), _p(out), _stream()), 'wd_roi_pool_fpn_f32')
    return out


_nms_ws = {}


def nms_sorted(boxes, idxs, iou_threshold):
    """boxes (n,4) xyxy already sorted by descending score; idxs (n) int32 group ids or None -> bool keep mask."""
    return nms_sorted_mask(boxes, idxs, iou_threshold).bool()


def nms_sorted_mask(boxes, idxs, iou_threshold):
    """nms_sorted as a uint8 mask (no dtype conversion launch)."""
    n = boxes.shape[0]
    keep = torch.zeros(n, dtype=torch.uint8, device=boxes.device)
    if n == 0:
        return keep
    boxes = boxes.contiguous().float()
    if idxs is not None:
        idxs = idxs.contiguous().to(torch.int32)
    lib = _lib.lib()
    need = int(lib.wd_nms_workspace(C.c_int(n)))
    key = (boxes.device, torch.cuda.current_stream().cuda_stream)
    ws = _nms_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=boxes.device)
        _nms_ws[key] = ws
    cnt = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    _lib.check(lib.wd_nms_sorted_f32(_p(boxes), _p(idxs), C.c_int(n), C.c_float(iou_threshold), _p(keep), _p(cnt),
                                     _p(ws), C.c_size_t(ws.numel()), _stream()), 'wd_nms_sorted_f32')
    return keep


def nms_select(boxes, scores, idxs, iou_threshold, cap, valid=None):
    """Static-shape batched_nms(...)[:cap]: (idx (cap,) int64 padded with -1, count int32[1] on the device).  `valid` (bool, n)
    marks real candidates; padding entries must carry a score below every real one and should get a group id of their own
    (they are kept by the NMS but never selected).  No host synchronisation."""
    n = boxes.shape[0]
    order = torch.argsort(scores, descending=True, stable=True)
    sb = boxes[order]
    si = None if idxs is None else idxs[order]
    keep = nms_sorted_mask(sb, si, iou_threshold)
    out = torch.empty(cap, dtype=torch.int64, device=boxes.device)
    cnt = torch.empty(1, dtype=torch.int32, device=boxes.device)
    v = None if valid is None else valid[order].to(torch.uint8).contiguous()
    _lib.check(_lib.lib().wd_select_kept(_p(keep), _p(v), _p(order), C.c_int(n), C.c_int(cap), _p(out), _p(cnt), _stream()),
               'wd_select_kept')
    return out, cnt


def batched_nms(boxes, scores, idxs, iou_threshold):
    """detectron2.layers.batched_nms / torchvision.ops.nms: indices of the kept boxes, descending score."""
    if boxes.shape[0] == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    order = torch.argsort(scores, descending=True, stable=True)
    keep = nms_sorted(boxes[order], None if idxs is None else idxs[order], iou_threshold)
    return order[keep]


def nms(boxes, scores, iou_threshold):
    return batched_nms(boxes, scores, None, iou_threshold)


def deform_pack_weight(weight, groups):
    """(C_out, C_in/groups, 3, 3) -> packed [group][tap][ci][co] float32 buffer for deform_conv3x3."""
    weight = weight.detach().contiguous().float()
    c_out, cg = weight.shape[0], weight.shape[1]
    c_in = cg * groups
    lib = _lib.lib()
    lib.wd_deform_packed_weight_floats.restype = C.c_size_t
    n = int(lib.wd_deform_packed_weight_floats(C.c_int(c_in), C.c_int(c_out), C.c_int(groups)))
    if n == 0:
        raise ValueError('deform_pack_weight: needs C_in == C_out divisible by groups')
    packed = torch.empty(n, dtype=torch.float32, device=weight.device)
    _lib.check(lib.wd_deform_pack_weight(_p(weight), C.c_int(c_in), C.c_int(c_out), C.c_int(groups), _p(packed), _stream()),
               'wd_deform_pack_weight')
    return packed


def deform_conv3x3(x, offset, packed_weight, groups, stride=1, pad=1, scale=None, bias=None, relu=False, mask=None, far_offsets=False,
                   table=None):
    """detectron2 DeformConv (3x3, dilation 1, deformable_groups 1) + fused FrozenBN affine / ReLU.
    x (N,C,H,W), offset (N,18,Ho,Wo), optional mask (N,9,Ho,Wo); returns (N,C,Ho,Wo) channels_last."""
    x = _nhwc(x)
    if offset is not None:
        offset = _nhwc(offset)
    if mask is not None:
        mask = _nhwc(mask)
    n, c, h, w = x.shape
    ho = (h + 2 * pad - 3) // stride + 1
    wo = (w + 2 * pad - 3) // stride + 1
    assert offset is None or offset.shape == (n, 18, ho, wo), (offset.shape, (n, 18, ho, wo))
    y = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    log = EVENT_LOG
    if log is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().wd_deform_conv3x3_tab_f32(_p(x), _p(offset), _p(mask), _p(packed_weight), _p(scale), _p(bias),
                                                    C.c_int(1 if relu else 0), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(c),
                                                    C.c_int(c), C.c_int(groups), C.c_int(stride), C.c_int(pad),
                                                    C.c_int(1 if far_offsets else 0), _p(table), _p(y), _stream()),
               'wd_deform_conv3x3_f32')
    if log is not None:
        e1.record()
        fn = _lib.lib().wd_deform_conv3x3_variant
        fn.restype = C.c_char_p
        name = fn(C.c_int(c), C.c_int(groups), C.c_int(stride), C.c_int(pad), C.c_int(0 if offset is None else 1)).decode()
        if far_offsets and 'pp_kernel' in name:
            name = 'deform_conv3x3_lds_kernel<32>'
        log.append(('%s: deform_conv3x3 C=%d %dx%d s%d%s' % (name, c, ho, wo, stride, '' if offset is not None else ' (no offsets)'),
                    2.0 * c * (c // groups) * 9 * ho * wo * n, e0, e1))
    return y


FAR_OFFSET_PX = 2.0          # halo of the persistent kernel's 14x14 input patch
FAR_OFFSET_SHARE = 0.2       # round 2: share of samples beyond it from which the per-tile fallback kernel won; since round 3 the
                             # persistent kernel is faster at every spread measured (tools/deform_r3_bench.py) - diagnostics only


def far_offset_share(offset):
    """Share of (pixel, tap) samples whose learned offset exceeds the persistent deform kernel's halo in y or x.
    Synchronises (one number to the host): call once per layer, outside stream capture."""
    n, _, h, w = offset.shape
    return float((offset.abs() > FAR_OFFSET_PX).reshape(n, 9, 2, h, w).any(dim=2).float().mean().item())


def gemm_nt(a, bt, bias=None, residual=None, relu=False, out=None):
    """act(a (M,K) @ bt (N,K)^T + bias [+ residual]) on the f32 matrix cores."""
    a = a.contiguous().float()
    bt = bt.contiguous().float()
    m, k = a.shape
    n = bt.shape[0]
    assert bt.shape[1] == k
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    if residual is not None:
        residual = residual.contiguous()
    lib = _lib.lib()
    need = int(lib.wd_gemm_nt_workspace(C.c_int(m), C.c_int(n), C.c_int(k)))
    if need and os.environ.get('WD_GEMM_V1') != '1':       # large shapes (box-head FC): 128x128 tiles, two-pass split-K
        ws = torch.empty(need, dtype=torch.uint8, device=a.device)
        _lib.check(lib.wd_gemm_nt_ws_f32(_p(a), _p(bt), _p(bias), _p(residual), C.c_int(1 if relu else 0), C.c_int(m), C.c_int(n),
                                         C.c_int(k), _p(out), _p(ws), C.c_size_t(need), _stream()), 'wd_gemm_nt_ws_f32')
        return out
    _lib.check(lib.wd_gemm_nt_f32(_p(a), _p(bt), _p(bias), _p(residual), C.c_int(1 if relu else 0), C.c_int(m),
                                  C.c_int(n), C.c_int(k), _p(out), _stream()), 'wd_gemm_nt_f32')
    return out


def bias_relu_(y, bias, relu=True):
    """In-place y = act(y + bias[col]) on a row-major (M,N) matrix: the epilogue pass behind a library GEMM."""
    m, n = y.shape
    _lib.check(_lib.lib().wd_bias_relu_f32(_p(y), _p(bias), C.c_long(m), C.c_int(n), C.c_int(1 if relu else 0), _stream()),
               'wd_bias_relu_f32')
    return y


def act_bwd(dy, y, scale, relu):
    """g = dy * (y > 0 if relu) * scale[col] on row-major (M, N) matrices: backward of the fused affine + ReLU epilogue (one pass)."""
    m, n = dy.shape
    g = torch.empty_like(dy)
    _lib.check(_lib.lib().wd_act_bwd_f32(_p(dy), _p(y) if relu else None, _p(scale), C.c_long(m), C.c_int(n), C.c_int(1 if relu else 0), _p(g),
                                         _stream()), 'wd_act_bwd_f32')
    return g


def upsample2x_nearest(x):
    """F.interpolate(x, scale_factor=2.0, mode='nearest') on a channels_last float32 map (FPN top-down pathway), one HIP kernel
    at HBM speed; returns a channels_last tensor.  Inference only (no autograd)."""
    assert x.dtype == torch.float32 and x.dim() == 4
    x = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
    n, c, h, w = x.shape
    out = torch.empty((n, c, 2 * h, 2 * w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    _lib.check(_lib.lib().wd_upsample2x_nhwc_f32(_p(x), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(c), _p(out), _stream()),
               'wd_upsample2x_nhwc_f32')
    return out


def groupnorm_relu_(x, weight, bias, groups, eps=1e-5, relu=True):
    """In-place GroupNorm (+ReLU) on an (R,C,H,W) channels_last tensor with H*W <= 64 (the 7x7 box-head maps)."""
    assert x.is_contiguous(memory_format=torch.channels_last) and x.dtype == torch.float32
    r, c, h, w = x.shape
    _lib.check(_lib.lib().wd_groupnorm_relu_nhwc_f32(_p(x), _p(weight), _p(bias), C.c_int(r), C.c_int(h * w), C.c_int(c),
                                                     C.c_int(groups), C.c_float(eps), C.c_int(1 if relu else 0), _stream()),
               'wd_groupnorm_relu_nhwc_f32')
    return x


class GroupNormReluFn(torch.autograd.Function):
    """y = relu(GroupNorm(x)) on (R,C,H,W) channels_last maps with H*W <= 64 (training graph of the box heads): the inference kernel writing to
    a second buffer, and one HIP backward launch (wd_groupnorm_relu_bwd_nhwc_f32) that recomputes the statistics from the saved input."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu):
        x = _nhwc(x)
        r, c, h, w = x.shape
        y = torch.empty_like(x, memory_format=torch.channels_last)
        _lib.check(_lib.lib().wd_groupnorm_relu_out_nhwc_f32(_p(x), _p(y), _p(weight), _p(bias), C.c_int(r), C.c_int(h * w), C.c_int(c),
                                                             C.c_int(groups), C.c_float(eps), C.c_int(1 if relu else 0), _stream()),
                   'wd_groupnorm_relu_out_nhwc_f32')
        ctx.save_for_backward(x, weight, bias)
        ctx.cfg = (groups, eps, bool(relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        groups, eps, relu = ctx.cfg
        dy = _nhwc(dy)
        r, c, h, w = x.shape
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        dg = torch.empty(c, dtype=torch.float32, device=x.device)
        db = torch.empty(c, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().wd_groupnorm_relu_bwd_nhwc_f32(_p(x), _p(dy), _p(weight), _p(bias), C.c_int(r), C.c_int(h * w), C.c_int(c),
                                                             C.c_int(groups), C.c_float(eps), C.c_int(1 if relu else 0), _p(dx), _p(dg), _p(db),
                                                             _stream()), 'wd_groupnorm_relu_bwd_nhwc_f32')
        return dx, dg, db, None, None, None


def tap_gemm_weight(weight, align=16):
    """(n_out, C, 3, 3) conv weight -> (ld, C) GEMM operand with row tap*n_out + n (tap = kh*3 + kw), zero rows up to
    ld = 9*n_out rounded up to `align` (keeps the rows of the GEMM result 16-byte aligned)."""
    n_out, c = weight.shape[0], weight.shape[1]
    ld = (9 * n_out + align - 1) // align * align
    w2 = torch.zeros((ld, c), dtype=torch.float32, device=weight.device)
    w2[:9 * n_out] = weight.detach().float().permute(2, 3, 0, 1).reshape(9 * n_out, c)
    return w2


def conv3x3_few(x, w2, bias, n_out, stride=1, deform_table=False, split=None):
    """3x3 conv, pad 1, with few output channels (the 18-channel deformable-offset conv) = one library GEMM over the
    input pixels (N = 9*n_out columns, full MFMA tiles) + the wd_tap_shift_add_f32 gather.  x (N,C,H,W) channels_last;
    returns (N,n_out,Ho,Wo) channels_last.  deform_table=True (n_out 18, stride 1): the gather launch also emits the sampling
    table of the persistent deformable kernel -> (offsets, table)."""
    x = _nhwc(x)
    n, c, h, w = x.shape
    a = x.permute(0, 2, 3, 1).reshape(n * h * w, c)
    if split is not None:
        # (packed planes of the same weight with ld rounded up to 32, ld): the N-thin GEMM on the split-operand kernel - pays from ~30 000 rows
        # (res3 at 1920 x 1280: 58 vs 78 us; res4's 9600 rows: 40 vs 39 us, left on the library; profiles/r05_offset_gemm_split.txt)
        packed, ld = split
        partial = gemm_split(a, packed, ld)
        ld_partial = ld
    else:
        partial = torch.mm(a, w2.t())
        ld_partial = w2.shape[0]
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    out = torch.empty((n, n_out, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if deform_table:
        assert n_out == 18 and stride == 1
        lib = _lib.lib()
        table = torch.empty(int(lib.wd_deform_table_bytes(C.c_int(n), C.c_int(h), C.c_int(w))), dtype=torch.uint8, device=x.device)
        _lib.check(lib.wd_deform_offsets_table_f32(_p(partial), C.c_int(ld_partial), _p(bias), C.c_int(n), C.c_int(h), C.c_int(w),
                                                   _p(out), _p(table), _stream()), 'wd_deform_offsets_table_f32')
        return out, table
    _lib.check(_lib.lib().wd_tap_shift_add_f32(_p(partial), C.c_int(ld_partial), C.c_int(n_out), _p(bias), C.c_int(n), C.c_int(h),
                                               C.c_int(w), C.c_int(stride), _p(out), _stream()), 'wd_tap_shift_add_f32')
    return out


_LT_WS = {}


def gemm_lt(a, weight, bias=None, residual=None, relu=False, out=None):
    """out = relu?(a @ weight.T + residual + bias) as ONE hipBLASLt GEMM (beta term + RELU_BIAS epilogue, wd_gemm_lt_f32).
    a (M,K), weight (N,K), residual / out (M,N) float32 contiguous; `out` may be the residual buffer (in place)."""
    m, k = a.shape
    n = weight.shape[0]
    assert a.is_contiguous() and weight.is_contiguous() and weight.shape[1] == k
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    assert out.is_contiguous() and (residual is None or residual.is_contiguous())
    key = (a.device, torch.cuda.current_stream().cuda_stream)       # one workspace per stream: frames may be in flight concurrently
    ws = _LT_WS.get(key)
    if ws is None:
        ws = _LT_WS[key] = torch.empty(64 << 20, dtype=torch.uint8, device=a.device)
    _lib.check(_lib.lib().wd_gemm_lt_f32(_p(a), _p(weight), _p(bias), _p(residual), _p(out), C.c_int(m), C.c_int(n), C.c_int(k),
                                         C.c_int(1 if relu else 0), _p(ws), C.c_size_t(ws.numel()), _stream()), 'wd_gemm_lt_f32')
    return out


def _split_log_begin():
    if EVENT_LOG is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _split_log_end(e0, tag, m, n, k):
    """bench.py instrumentation: HIP events around the launch on the launch stream; flops = the bf16 matrix-core flops actually issued
    (six cross terms: 6 x 2 M N K), priced against the dense bf16 MFMA peak."""
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    EVENT_LOG.append((tag, 12.0 * m * n * k, e0, e1))


_SPLIT_WS_BYTES = {}
_DEBUG_FILL = os.environ.get('WD_DEBUG_FILL') == 'nan'        # diagnostics: NaN-fill fresh outputs / workspaces (anything left unwritten shows)


def _split_workspace(m, n, k, device):
    """Scratch for the K-sliced form of a small shape (wd_gemm_split_workspace; None for shapes that run unsliced).  A fresh tensor per call from
    torch's caching allocator: stream-ordered reuse, safe under hipGraph capture."""
    key = (m, n, k)
    nbytes = _SPLIT_WS_BYTES.get(key)
    if nbytes is None:
        nbytes = _SPLIT_WS_BYTES[key] = int(_lib.lib().wd_gemm_split_workspace(C.c_long(m), C.c_int(n), C.c_int(k)))
    if nbytes == 0:
        return None, 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if _DEBUG_FILL:
        ws.view(torch.float32).fill_(float('nan'))
    return ws, nbytes


def split_pack_weight(weight, transpose=False):
    """(N, K) f32 GEMM weight, or an (N, C, 3, 3) / (N, C, 1, 1) convolution weight -> packed bf16 planes (hi, mid, lo) in MFMA fragment order
    for gemm_split / conv_split (wd_gemm_split_pack_weight).  Convolution weights are laid out k = (kh * 3 + kw) * C + c.  transpose=True (2-D
    weights): pack weight.T without materialising it (the backward-data GEMM of training: dA = g @ W is the NT product of g with W.T)."""
    w = weight.detach()
    if w.dtype != torch.float32:
        w = w.float()
    if w.dim() == 4:
        w = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()
    if transpose:
        w = w.t()
    n, k = w.shape
    lib = _lib.lib()
    nbytes = int(lib.wd_gemm_split_packed_bytes(C.c_int(n), C.c_int(k)))
    if nbytes == 0:
        raise ValueError('split_pack_weight: K must be a multiple of 64 (N=%d K=%d)' % (n, k))
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.check(lib.wd_gemm_split_pack_weight_strided(_p(w), C.c_int(n), C.c_int(k), C.c_long(w.stride(0)), C.c_long(w.stride(1)), _p(packed), _stream()),
               'wd_gemm_split_pack_weight_strided')
    return packed


class _PackEntry:
    __slots__ = ('ref', 'kind', 'key', 'packed', 'version', 'desc', 'blocks', 'used')


_PACK_REGISTRY = []          # every cached pack of the process (the entry holds a weak reference to its weight)
_PACK_DESC_CACHE = [None, None]      # (ids of the entries of the last batch, its descriptor array on the device)
_PACK_DESC_FIELDS = ('src', 'dst', 's_n', 's_c', 's_kh', 's_kw', 'first_block', 'N', 'K', 'C', 'ksize', 'flip', 'reserved')     # WdSplitPackDesc


def _pack_desc_dtype():
    import numpy as np
    return np.dtype([(f, '<u8' if f in ('src', 'dst') else ('<i8' if f in ('s_n', 's_c', 's_kh', 's_kw', 'first_block') else '<i4')) for f in _PACK_DESC_FIELDS])


def _new_pack_entry(weight, kind):
    import weakref
    w = weight
    if w.dtype != torch.float32 or not w.is_cuda:
        raise ValueError('split_pack_cached: float32 weights on the GPU only')
    if w.dim() == 2:
        if kind == 'fwd':
            n, k, sn, sc = w.shape[0], w.shape[1], w.stride(0), w.stride(1)
        elif kind == 'T':
            n, k, sn, sc = w.shape[1], w.shape[0], w.stride(1), w.stride(0)
        else:
            raise ValueError('split_pack_cached: kind %r needs a convolution weight' % kind)
        c, ks, skh, skw, flip = k, 1, 0, 0, 0
    elif w.dim() == 4 and w.shape[2] == w.shape[3]:
        ks = w.shape[2]
        skh, skw = w.stride(2), w.stride(3)
        if kind == 'fwd':
            n, c, sn, sc, flip = w.shape[0], w.shape[1], w.stride(0), w.stride(1), 0
        elif kind == 'dx':                  # backward-data convolution: taps flipped, channel roles swapped
            n, c, sn, sc, flip = w.shape[1], w.shape[0], w.stride(1), w.stride(0), 1
        else:
            raise ValueError('split_pack_cached: kind %r needs a 2-D weight' % kind)
        k = ks * ks * c
    else:
        raise ValueError('split_pack_cached: 2-D or square-kernel 4-D weights')
    nbytes = int(_lib.lib().wd_gemm_split_packed_bytes(C.c_int(n), C.c_int(k)))
    if nbytes == 0 or c % 8:
        raise ValueError('split_pack_cached: K must be a multiple of 64 (N=%d K=%d)' % (n, k))
    e = _PackEntry()
    e.ref, e.kind = weakref.ref(weight), kind
    e.key = (w.data_ptr(), tuple(w.shape), tuple(w.stride()))
    e.packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    e.version = -1
    e.blocks = ((n + 31) // 32 * 32 * (k // 8) + 255) // 256
    e.desc = (w.data_ptr(), e.packed.data_ptr(), sn, sc, skh, skw, 0, n, k, c, ks, flip, 0)
    e.used = True
    return e


def _pack_entries(entries):
    """One launch (wd_gemm_split_pack_batch) that packs every entry's weight as it is NOW."""
    import numpy as np
    if not entries:
        return
    ids = tuple(id(e) for e in entries)
    dev = entries[0].packed.device
    total = sum(e.blocks for e in entries)
    if _PACK_DESC_CACHE[0] == ids:
        desc = _PACK_DESC_CACHE[1]
    else:
        arr = np.zeros(len(entries), dtype=_pack_desc_dtype())
        first = 0
        for i, e in enumerate(entries):
            arr[i] = e.desc[:6] + (first,) + e.desc[7:]
            first += e.blocks
        desc = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
        _PACK_DESC_CACHE[0], _PACK_DESC_CACHE[1] = ids, desc
    _lib.check(_lib.lib().wd_gemm_split_pack_batch(_p(desc), C.c_int(len(entries)), C.c_long(total), _stream()), 'wd_gemm_split_pack_batch')
    for e in entries:
        w = e.ref()
        e.version = w._version if w is not None else e.version


def split_pack_cached(weight, kind='fwd'):
    """The packed bf16 planes of `weight` for gemm_split / conv_split, cached ON the weight tensor and re-packed when the weight has changed
    (its autograd version counter: every in-place update - an optimizer step - bumps it).  kind: 'fwd' = the weight as it is (2-D (N, K), or a
    convolution weight (N, C, ks, ks) read through its strides, no permute copy); 'T' = the transpose of a 2-D weight (backward-data GEMM);
    'dx' = a convolution weight with flipped taps and swapped channel roles (backward-data convolution).  The first stale weight met after an
    optimizer step re-packs EVERY cached weight that was used since the previous sweep in ONE launch (wd_gemm_split_pack_batch): a training step of
    the X-152 detector packs ~250 weights in two orientations - 2.4 ms of 10-us launches plus the permute / flip copies before (rocprof, round 5)."""
    packs = getattr(weight, '_wd_split_packs', None)
    if packs is None:
        packs = {}
        weight._wd_split_packs = packs               # lives and dies with the tensor object: no stale hit through a recycled address
    e = packs.get(kind)
    if e is None or e.key != (weight.data_ptr(), tuple(weight.shape), tuple(weight.stride())) or e.packed.device != weight.device:
        e = packs[kind] = _new_pack_entry(weight, kind)
        _PACK_REGISTRY.append(e)
        _pack_entries([e])
    elif e.version != weight._version:
        stale, alive = [], []
        for q in _PACK_REGISTRY:
            w = q.ref()
            if w is None or getattr(w, '_wd_split_packs', {}).get(q.kind) is not q:
                continue                                 # weight gone, or its entry was replaced
            alive.append(q)
            if q.packed.device == e.packed.device and (q is e or (q.used and q.version != w._version)):
                stale.append(q)
            q.used = False
        _PACK_REGISTRY[:] = alive
        _pack_entries(stale)
    e.used = True
    return e.packed


def gemm_split_ok(a, n, bias=None, residual=None, out=None):
    """The full precondition of wd_gemm_split_f32 (wd_gemm_split_supported: K % 64, N % 32, row strides % 4, 16-byte aligned pointers, M * lda < 2^31)
    for a 2-D f32 operand `a` (row-strided views allowed) - ONE predicate for every call site that may route a GEMM to the split-operand kernel, so
    that a strided / offset view or a very large batch falls back to the library GEMM instead of raising."""
    if not (a.is_cuda and a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1):
        return False
    for t in (bias, residual, out):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            return False
    m, k = a.shape
    return bool(_lib.lib().wd_gemm_split_supported(_p(a), C.c_long(a.stride(0)), _p(bias), _p(residual), _p(out), C.c_long(n), C.c_long(m), C.c_int(n), C.c_int(k)))


def conv_split_ok(x, c, n, ksize, stride, pad):
    """Precondition of wd_conv_split_f32 (ksize 1 / 3, pad >= 0, C % 64, N % 32, fewer than 2^31 elements)."""
    return (x.is_cuda and x.dtype == torch.float32 and ksize in (1, 3) and stride >= 1 and pad >= 0 and c % 64 == 0 and n % 32 == 0
            and x.numel() < 2 ** 31 and x.data_ptr() % 16 == 0)


def gemm_split(a, packed, n, bias=None, residual=None, relu=False, out=None):
    """out (M, N) = relu?(a (M, K) @ W.T + bias + residual) with W packed by split_pack_weight: f32 operands as exact 3 x bf16 splits, six
    cross terms on the bf16 matrix cores, f32 accumulation (wd_gemm_split_f32).  `a` may be a row-strided view; residual may be `out`."""
    m, k = a.shape
    assert a.dtype == torch.float32 and a.stride(1) == 1
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        if _DEBUG_FILL:
            out.fill_(float('nan'))
    assert out.is_contiguous() and (residual is None or residual.is_contiguous())
    ws, ws_bytes = _split_workspace(m, n, k, a.device)
    ev = _split_log_begin()
    _lib.check(_lib.lib().wd_gemm_split_f32(_p(a), C.c_long(a.stride(0)), _p(packed), _p(bias), _p(residual), _p(out), C.c_long(n), C.c_int(m),
                                            C.c_int(n), C.c_int(k), C.c_int(1 if relu else 0), _p(ws), C.c_size_t(ws_bytes), _stream()), 'wd_gemm_split_f32')
    _split_log_end(ev, 'gemm_split_kernel: 1x1 conv / GEMM M=%d N=%d K=%d' % (m, n, k), m, n, k)
    return out


class _SplitIO(C.Structure):
    """struct WdSplitIO of include/waymodet.h"""
    _fields_ = [('a', C.c_void_p), ('lda', C.c_long), ('a_planes', C.c_void_p), ('residual', C.c_void_p), ('residual_planes', C.c_void_p),
                ('out', C.c_void_p), ('out_planes', C.c_void_p), ('ldc', C.c_long)]


def split_planes_bytes(m, k):
    return int(_lib.lib().wd_split_planes_bytes(C.c_long(m), C.c_int(k)))


def split_planes_empty(m, k, device):
    """Uninitialised activation planes of an (m, k) matrix (uint8 tensor; rows of the last 32-row block beyond m are never read into results)."""
    n = split_planes_bytes(m, k)
    if n == 0:
        raise ValueError('activation planes need K %% 32 == 0 (M=%d K=%d)' % (m, k))
    return torch.empty(n, dtype=torch.uint8, device=device)


def split_planes_pack(a, out=None):
    """f32 (M, K) (row-strided view allowed) -> activation planes (wd_split_planes_pack_f32): three bf16 planes per 32 x 32 block in LDS-image order."""
    m, k = a.shape
    assert a.dtype == torch.float32 and a.stride(1) == 1
    if out is None:
        out = split_planes_empty(m, k, a.device)
    _lib.check(_lib.lib().wd_split_planes_pack_f32(_p(a), C.c_long(a.stride(0)), C.c_long(m), C.c_int(k), _p(out), _stream()), 'wd_split_planes_pack_f32')
    return out


def split_planes_unpack(planes, m, k):
    """activation planes -> f32 (M, K), exact (hi + mid + lo)."""
    out = torch.empty((m, k), dtype=torch.float32, device=planes.device)
    _lib.check(_lib.lib().wd_split_planes_unpack_f32(_p(planes), C.c_long(m), C.c_int(k), _p(out), C.c_long(k), _stream()), 'wd_split_planes_unpack_f32')
    return out


def gemm_split_io(m, n, k, packed, a=None, a_planes=None, bias=None, residual=None, residual_planes=None, relu=False, out=None, out_planes=None,
                  want_out=True, want_planes=False):
    """The split-operand GEMM with every operand as f32 or as activation planes (wd_gemm_split_io).  Returns (out or None, out_planes or None)."""
    dev = packed.device
    if want_out and out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=dev)
    if want_planes and out_planes is None:
        out_planes = split_planes_empty(m, n, dev)
    io = _SplitIO()
    if a is not None:
        assert a.dtype == torch.float32 and a.stride(1) == 1 and tuple(a.shape) == (m, k)
        io.a, io.lda = a.data_ptr(), a.stride(0)
    if a_planes is not None:
        io.a_planes = a_planes.data_ptr()
    if residual is not None:
        assert residual.is_contiguous()
        io.residual = residual.data_ptr()
    if residual_planes is not None:
        io.residual_planes = residual_planes.data_ptr()
    if out is not None:
        assert out.is_contiguous()
        io.out = out.data_ptr()
    if out_planes is not None:
        io.out_planes = out_planes.data_ptr()
    io.ldc = n
    ws, ws_bytes = _split_workspace(m, n, k, dev)
    ev = _split_log_begin()
    _lib.check(_lib.lib().wd_gemm_split_io(C.byref(io), _p(packed), _p(bias), C.c_int(m), C.c_int(n), C.c_int(k), C.c_int(1 if relu else 0), _p(ws),
                                           C.c_size_t(ws_bytes), _stream()), 'wd_gemm_split_io')
    _split_log_end(ev, '%s: 1x1 conv / GEMM M=%d N=%d K=%d' % ('gemm_split_planes_kernel' if a_planes is not None else 'gemm_split_kernel', m, n, k), m, n, k)
    return out, out_planes


def conv_split(x, packed, n_out, ksize, stride=1, pad=0, bias=None, residual=None, relu=False):
    """Dense ksize x ksize convolution (1 or 3) of an (N, C, H, W) channels_last map as an implicit GEMM on the split-operand kernel
    (wd_conv_split_f32); returns (N, n_out, Ho, Wo) channels_last.  residual: (N, n_out, Ho, Wo) channels_last or None."""
    x = _nhwc(x)
    b, c, h, w = x.shape
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    out = torch.empty((b, n_out, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if _DEBUG_FILL:
        out.fill_(float('nan'))
    if residual is not None:
        residual = _nhwc(residual)
    ws, ws_bytes = _split_workspace(b * ho * wo, n_out, ksize * ksize * c, x.device)
    ev = _split_log_begin()
    _lib.check(_lib.lib().wd_conv_split_f32(_p(x), C.c_int(b), C.c_int(h), C.c_int(w), C.c_int(c), _p(packed), C.c_int(ksize), C.c_int(stride),
                                            C.c_int(pad), _p(bias), _p(residual), _p(out), C.c_int(n_out), C.c_int(1 if relu else 0), _p(ws), C.c_size_t(ws_bytes),
                                            _stream()),
               'wd_conv_split_f32')
    k_eff = ksize * ksize * c
    if ksize == 3 and stride == 1 and pad == 1 and h >= 3 and w >= 3 and h * w <= 81 and b >= 256 and os.environ.get('WD_SPLIT_NO_POSMAJOR') != '1':
        # position-major tiles (csrc/det_gemm_split.hip MODE 3) SKIP the taps in the zero padding: only the walked taps are matrix work that was issued
        walked = (h - 2) * (w - 2) * 9 + (2 * (h - 2) + 2 * (w - 2)) * 6 + 4 * 4
        k_eff = k_eff * walked / float(9 * h * w)
    _split_log_end(ev, 'gemm_split_kernel: %dx%d conv s%d M=%d N=%d K=%d' % (ksize, ksize, stride, b * ho * wo, n_out, ksize * ksize * c), b * ho * wo, n_out,
                   k_eff)
    return out


SCALE_CLAMP = 4.135166556742356          # log(1000 / 16), Box2BoxTransform


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def rpn_topk_decode(logits, deltas, anchors, k, img_h, img_w):
    """Per FPN level: the k best objectness logits (sorted) and apply_deltas + clip of those anchors, all levels in 2-3
    launches (wd_rpn_topk_decode_f32).  logits[l] (n_l,), deltas[l] / anchors[l] (n_l, 4) float32.
    Returns (boxes (R,4), scores (R), group (R) int32 = level or -1 for an empty box, valid (R) uint8), R = sum min(k, n_l)."""
    logits = [t.contiguous().float() for t in logits]
    deltas = [t.contiguous().float() for t in deltas]
    anchors = [t.contiguous().float() for t in anchors]
    n = [int(t.numel()) for t in logits]
    dev = logits[0].device
    rows = sum(min(k, v) for v in n)
    counts = (C.c_int * len(n))(*n)
    lib = _lib.lib()
    need = int(lib.wd_rpn_topk_workspace(counts, C.c_int(len(n)), C.c_int(k)))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    boxes = torch.empty((rows, 4), dtype=torch.float32, device=dev)
    scores = torch.empty(rows, dtype=torch.float32, device=dev)
    group = torch.empty(rows, dtype=torch.int32, device=dev)
    valid = torch.empty(rows, dtype=torch.uint8, device=dev)
    _lib.check(lib.wd_rpn_topk_decode_f32(_ptr_array(logits), _ptr_array(deltas), _ptr_array(anchors), counts, C.c_int(len(n)),
                                          C.c_int(k), C.c_float(img_h), C.c_float(img_w), C.c_float(SCALE_CLAMP), _p(boxes),
                                          _p(scores), _p(group), _p(valid), _p(ws), C.c_size_t(need), _stream()),
               'wd_rpn_topk_decode_f32')
    return boxes, scores, group, valid


def _sorted_outputs(n, dev):
    return (torch.empty((n, 4), dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev),
            torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.uint8, device=dev),
            torch.empty(n, dtype=torch.int64, device=dev))


def sort_candidates(boxes, scores, group, valid, valid2=None):
    """Stable descending score sort of <= 8192 candidate rows + gather: (boxes, scores, group, valid [& valid2], order) sorted."""
    n = boxes.shape[0]
    out = _sorted_outputs(n, boxes.device)
    _lib.check(_lib.lib().wd_sort_candidates_f32(_p(boxes), _p(scores), _p(group), _p(valid), _p(valid2), C.c_int(n),
                                                 *[_p(t) for t in out], _stream()), 'wd_sort_candidates_f32')
    return out


def nms_segmented(boxes, idxs, seg_offsets, iou_threshold):
    """Greedy NMS on independent row ranges (each sorted by descending score) in one pair of launches -> uint8 keep mask.
    seg_offsets: python list of n_seg + 1 row offsets starting at 0."""
    n = boxes.shape[0]
    keep = torch.zeros(n, dtype=torch.uint8, device=boxes.device)
    if n == 0:
        return keep
    lib = _lib.lib()
    need = int(lib.wd_nms_workspace(C.c_int(n)))
    key = (boxes.device, torch.cuda.current_stream().cuda_stream)
    ws = _nms_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=boxes.device)
        _nms_ws[key] = ws
    n_seg = len(seg_offsets) - 1
    cnt = torch.zeros(n_seg, dtype=torch.int32, device=boxes.device)
    offs = (C.c_int32 * (n_seg + 1))(*[int(v) for v in seg_offsets])
    _lib.check(lib.wd_nms_segmented_f32(_p(boxes.contiguous()), _p(idxs), offs, C.c_int(n_seg), C.c_float(iou_threshold), _p(keep),
                                        _p(cnt), _p(ws), C.c_size_t(ws.numel()), _stream()), 'wd_nms_segmented_f32')
    return keep


def box_candidates(boxes, s0, s1, s2, n_valid, score_thresh, img_h, img_w):
    """fast_rcnn_inference_single_image candidates (every (row, class) pair; score = mean softmax of the three stages), sorted
    by descending score with the real ones first: (clipped boxes, scores, group = class | -1, valid, order)."""
    r, nc = s0.shape[0], s0.shape[1] - 1
    out = _sorted_outputs(r * nc, boxes.device)
    _lib.check(_lib.lib().wd_box_candidates_f32(_p(boxes.contiguous()), _p(s0.contiguous()), _p(s1.contiguous()), _p(s2.contiguous()),
                                                _p(n_valid), C.c_int(r), C.c_int(nc), C.c_float(score_thresh), C.c_float(img_h),
                                                C.c_float(img_w), *[_p(t) for t in out], _stream()), 'wd_box_candidates_f32')
    return out


def gather_kept(keep, valid, boxes, scores, order, cap, num_classes=0):
    """The first `cap` kept & valid sorted candidates as fixed-size outputs (unused rows zero) and the device-side count.
    num_classes == 0: proposals -> (boxes (cap,4), count); else detections -> (boxes, scores, classes int64, count)."""
    dev = boxes.device
    ob = torch.empty((cap, 4), dtype=torch.float32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    if num_classes:
        osc = torch.empty(cap, dtype=torch.float32, device=dev)
        oc = torch.empty(cap, dtype=torch.int64, device=dev)
    else:
        osc = oc = None
    _lib.check(_lib.lib().wd_gather_kept_f32(_p(keep), _p(valid), _p(boxes), _p(scores), _p(order), C.c_int(boxes.shape[0]),
                                             C.c_int(cap), C.c_int(num_classes), _p(ob), _p(osc), _p(oc), _p(cnt), _stream()),
               'wd_gather_kept_f32')
    return (ob, osc, oc, cnt) if num_classes else (ob, cnt)


def detections_to_wire(boxes, scores, classes, count, in_w, in_h, out_w, out_h, hflip=False):
    """Pixel boxes of the network input -> wire values of the detection JSON in one launch: (xywhs (5, n) float64 = integer
    x, y, w, h and the 5-decimal score, category (n) int32 = class + 1, 0 for slots >= count)."""
    n = boxes.shape[0]
    xywhs = torch.empty((5, n), dtype=torch.float64, device=boxes.device)
    cat = torch.empty(n, dtype=torch.int32, device=boxes.device)
    _lib.check(_lib.lib().wd_detections_to_wire(_p(boxes.contiguous()), _p(scores.contiguous()), _p(classes.contiguous()), _p(count),
                                                C.c_int(n), C.c_int(in_w), C.c_int(in_h), C.c_int(1 if hflip else 0), C.c_int(out_w),
                                                C.c_int(out_h), _p(xywhs), _p(cat), _stream()), 'wd_detections_to_wire')
    return xywhs, cat


def decode_boxes(deltas, boxes, weights, index=None, clip=None, scale_clamp=4.135166556742356):
    """apply_deltas(deltas[index], boxes[index], weights) [+ clip to (h, w)] in one launch; (n,4) float32 xyxy."""
    deltas = deltas.contiguous().float()
    boxes = boxes.contiguous().float()
    n = deltas.shape[0] if index is None else index.shape[0]
    out = torch.empty((n, 4), dtype=torch.float32, device=deltas.device)
    if n == 0:
        return out
    if index is not None:
        index = index.contiguous().to(torch.int64)
    ch, cw = (float(clip[0]), float(clip[1])) if clip is not None else (0.0, 0.0)
    _lib.check(_lib.lib().wd_decode_boxes_f32(_p(deltas), _p(boxes), _p(index), C.c_int(n), C.c_float(weights[0]), C.c_float(weights[1]),
                                              C.c_float(weights[2]), C.c_float(weights[3]), C.c_float(scale_clamp), C.c_float(cw),
                                              C.c_float(ch), _p(out), _stream()), 'wd_decode_boxes_f32')
    return out


def preprocess_out_shape(h, w, scale=1.0, divisor=32):
    """(Ho, Wo, Hp, Wp) of preprocess(): resized extent and the extent padded to a multiple of `divisor`."""
    o = [C.c_int(0) for _ in range(4)]
    _lib.check(_lib.lib().wd_preprocess_out_shape(C.c_int(h), C.c_int(w), C.c_double(scale), C.c_int(divisor),
                                                  *[C.byref(v) for v in o]), 'wd_preprocess_out_shape')
    return tuple(v.value for v in o)


def preprocess(x, scale=1.0, hflip=False, vflip=False, swap_rb=True, mean=None, std=None, divisor=32):
    """Fused TTA pre-process + BGR swap + normalise + pad (one kernel, one HBM pass).
    x: (N,3,H,W) float32 0..255 (contiguous NCHW) or (N,H,W,3) uint8 on the GPU.
    Returns ((N,3,Hp,Wp) float32 tensor in channels_last storage, (Ho, Wo))."""
    if x.dtype == torch.uint8:
        assert x.dim() == 4 and x.shape[3] == 3, 'uint8 input must be (N,H,W,3)'
        layout, (n, h, w) = 1, x.shape[:3]
    else:
        assert x.dim() == 4 and x.shape[1] == 3, 'float input must be (N,3,H,W)'
        x = x.float()
        layout, (n, h, w) = 0, (x.shape[0], x.shape[2], x.shape[3])
    x = x.contiguous()
    ho, wo, hp, wp = preprocess_out_shape(h, w, scale, divisor)
    out = torch.empty((n, 3, hp, wp), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    m3 = (C.c_float * 3)(*[float(v) for v in mean]) if mean is not None else None
    s3 = (C.c_float * 3)(*[float(v) for v in std]) if std is not None else None
    _lib.check(_lib.lib().wd_preprocess_f32(_p(x), C.c_int(layout), C.c_int(n), C.c_int(h), C.c_int(w), C.c_double(scale),
                                            C.c_int(1 if hflip else 0), C.c_int(1 if vflip else 0), C.c_int(1 if swap_rb else 0),
                                            m3, s3, C.c_int(divisor), _p(out), _stream()), 'wd_preprocess_f32')
    return out, (ho, wo)


# ---------------------------------------------------------------------------------------------------------------
# image decode
def autocontrast_(img_u8):
    """PIL's ImageOps.autocontrast (cutoff 0) on an (H, W, 3) uint8 GPU image, IN PLACE, two HIP launches (csrc/det_preprocess.hip:
    wd_autocontrast_u8), bit-exact with PIL.  Returns the image."""
    assert img_u8.is_cuda and img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.shape[2] == 3 and img_u8.is_contiguous()
    ws = torch.empty(6, dtype=torch.int32, device=img_u8.device)
    _lib.check(_lib.lib().wd_autocontrast_u8(_p(img_u8), C.c_int(img_u8.shape[0]), C.c_int(img_u8.shape[1]), _p(ws), _stream()),
               'wd_autocontrast_u8')
    return img_u8


def jpeg_info(data):
    """(width, height, components, h_samp, v_samp, restart_interval) of a baseline JPEG (host-only header parse)."""
    buf = bytes(data)
    o = [C.c_int32(0) for _ in range(6)]
    _lib.check(_lib.lib().wd_jpeg_info(buf, C.c_int64(len(buf)), *[C.byref(v) for v in o]), 'wd_jpeg_info')
    return tuple(v.value for v in o)


def jpeg_decode(data, device=None, return_rounds=False, out=None):
    """bytes of a JPEG file -> (H, W, 3) uint8 RGB tensor on the GPU, bit-exact with
    `PIL.Image.open(...).convert('RGB')` (the reference's decode, detnet/inference.py:170).  Huffman decoding, inverse DCT,
    chroma upsampling and colour conversion all run in HIP kernels (csrc/jpeg_decode.hip); no host decoder is involved.
    Unsupported flavours (progressive, CMYK, ...) raise WaymoTrackError."""
    buf = bytes(data)
    w, h = C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.lib().wd_jpeg_info(buf, C.c_int64(len(buf)), C.byref(w), C.byref(h), None, None, None, None), 'wd_jpeg_info')
    if out is None:
        out = torch.empty((h.value, w.value, 3), dtype=torch.uint8, device=device or 'cuda')
    else:                                                                         # decode into a caller-owned frame slot
        assert out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous() and tuple(out.shape) == (h.value, w.value, 3)
    rounds = C.c_int32(0)
    _lib.check(_lib.lib().wd_jpeg_decode_rgb_u8(buf, C.c_int64(len(buf)), _p(out), C.c_int64(out.numel()), C.byref(w), C.byref(h),
                                                C.byref(rounds), _stream()), 'wd_jpeg_decode_rgb_u8')
    return (out, rounds.value) if return_rounds else out


# ---------------------------------------------------------------------------------------------------------------
# training (fwd + bwd): autograd wrappers around the forward kernels and the backward kernels of det_backward.hip
def deform_im2col(x, offset, stride=1, pad=1, groups=1):
    """col (groups, P, 9, C/groups) float32, P = N*Ho*Wo (detectron2 deformable_im2col; group-major column layout)."""
    x = _nhwc(x); offset = _nhwc(offset)
    n, c, h, w = x.shape
    ho = (h + 2 * pad - 3) // stride + 1
    wo = (w + 2 * pad - 3) // stride + 1
    col = torch.empty((groups, n * ho * wo, 9, c // groups), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().wd_deform_im2col_f32(_p(x), _p(offset), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(c), C.c_int(groups),
                                               C.c_int(stride), C.c_int(pad), _p(col), _stream()), 'wd_deform_im2col_f32')
    return col


def deform_col2im(dcol, x, offset, stride=1, pad=1, groups=1):
    """dcol (groups, P, 9, C/groups) -> (dx (N,C,H,W) channels_last, doffset (N,18,Ho,Wo) channels_last)."""
    x = _nhwc(x); offset = _nhwc(offset)
    dcol = dcol.contiguous()
    n, c, h, w = x.shape
    dx = torch.zeros_like(x, memory_format=torch.channels_last)
    doff = torch.zeros_like(offset, memory_format=torch.channels_last)
    log = EVENT_LOG
    if log is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().wd_deform_col2im_f32(_p(dcol), _p(x), _p(offset), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(c),
                                               C.c_int(groups), C.c_int(stride), C.c_int(pad), _p(dx), _p(doff), _stream()),
               'wd_deform_col2im_f32')
    if log is not None:
        e1.record()
        ho, wo = offset.shape[2], offset.shape[3]
        # algorithmic HBM bytes: the column gradients once (P x 9 x C floats), the offsets in, dx and doffset out, x for doffset
        nbytes = 4.0 * (n * ho * wo * 9 * c + 2 * n * ho * wo * 18 + 2 * n * h * w * c)
        log.append(('deform_col2im (dx gather + doffset): C=%d %dx%d s%d' % (c, ho, wo, stride), nbytes, e0, e1))
    return dx, doff


FUSED_DEFORM_BACKWARD = os.environ.get('WD_FUSED_DEFORM_BWD', '1') != '0'     # A/B switch: 0 = the im2col / GEMM / col2im form for every layer


FUSED_DEFORM_DXOFF = os.environ.get('WD_FUSED_DEFORM_DXOFF', '1') != '0'       # A/B switch for the dX / dOffset half alone


def fused_deform_backward_supported(c, cout, groups, stride, pad):
    """The fused kernels (csrc/det_deform_bwd.hip) cover the stride-1 DeformConvs of res3 / res4: 16 or 32 channels per group.  They also run
    stride 2 (WD_FUSED_DEFORM_S2=1; the 14 x 14 patch then holds the middle of the 17 x 17 footprint and a third of the samples goes through the
    per-sample kernel) - measured no faster than the column-slab form (1231 / 609 us against 1171 / 585 us for the res3 / res4 layer), so off."""
    return (FUSED_DEFORM_BACKWARD and (stride == 1 or (stride == 2 and os.environ.get('WD_FUSED_DEFORM_S2', '0') == '1')) and pad == 1
            and c == cout and c % groups == 0 and c // groups in (16, 32))


def deform_dw(x, offset, dy_nhwc, groups, y_act=None, scale=None, stride=1):
    """dW (C, C/groups, 3, 3) from x, offset and dy (N,H,W,C contiguous): the columns are blended per tile in registers and consumed by the
    MFMAs directly, a second launch sums the workgroups' partial sums into the weight's own layout (wd_deform_dw_f32)."""
    x = _nhwc(x); offset = _nhwc(offset)
    n, c, h, w = x.shape
    cg = c // groups
    dw = torch.empty((c, cg, 3, 3), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    scratch = torch.empty(L.wd_deform_dw_scratch_floats(C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(c), C.c_int(groups), C.c_int(stride)),
                          dtype=torch.float32, device=x.device)
    _lib.check(L.wd_deform_dw_f32(_p(x), _p(offset), _p(dy_nhwc), _p(y_act) if y_act is not None else None, _p(scale) if scale is not None else None,
                                  C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(c), C.c_int(groups), C.c_int(stride), _p(scratch), _p(dw), _stream()),
               'wd_deform_dw_f32')
    return dw


def deform_dxoff(x, offset, dy_nhwc, weight, groups, y_act=None, scale=None, stride=1):
    """(dx (N,C,H,W) channels_last, doffset (N,18,H,W) channels_last) without the dcol slab (wd_deform_dxoff_f32): per tile dcol = dY W on the
    MFMAs, dOffset from the fragment in registers, dX by a gather over the inverted sampling table out of LDS."""
    x = _nhwc(x); offset = _nhwc(offset)
    n, c, h, w = x.shape
    L = _lib.lib()
    tables = torch.empty(L.wd_deform_bwd_tables_bytes(C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(stride)), dtype=torch.uint8, device=x.device)
    packed = torch.empty(weight.numel(), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x, memory_format=torch.channels_last)
    doff = torch.empty_like(offset, memory_format=torch.channels_last)
    log = EVENT_LOG
    if log is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(L.wd_deform_dxoff_f32(_p(x), _p(offset), _p(dy_nhwc), _p(y_act) if y_act is not None else None,
                                     _p(scale) if scale is not None else None, _p(weight.contiguous()), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(c),
                                     C.c_int(groups), C.c_int(stride), _p(tables), _p(packed), _p(dx), _p(doff), _stream()), 'wd_deform_dxoff_f32')
    if log is not None:
        e1.record()
        # algorithmic flops: dcol = dY W per group on the f32 MFMAs (the gather and the dOffset dot products are VALU work on top)
        ho, wo = offset.shape[2], offset.shape[3]
        log.append(('deform_dxoff (tables + dX gather + dOffset + far samples): C=%d %dx%d%s' % (c, ho, wo, ' s2' if stride == 2 else ''),
                    2.0 * n * ho * wo * 9 * (c // groups) * c, e0, e1))
    return dx, doff


class DeformConvFn(torch.autograd.Function):
    """y = DeformConv(x, offset; weight), groups / stride / pad as in the forward kernel (no affine, no ReLU)."""

    @staticmethod
    def forward(ctx, x, offset, weight, groups, stride, pad, scale=None, bias=None, relu=False):
        """scale / bias / relu: the FrozenBN affine + ReLU of the block, fused into the HIP kernel's epilogue like at inference (round 4:
        the training graph spent three elementwise passes per block on them, and two more in backward); constants - no gradient."""
        packed = deform_pack_weight(weight, groups)
        y = deform_conv3x3(x, offset, packed, groups, stride, pad, scale, bias, relu)
        ctx.save_for_backward(x, offset, weight, scale if scale is not None else x.new_empty(0), y if relu else x.new_empty(0))
        ctx.cfg = (groups, stride, pad, scale is not None, bool(relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        """im2col -> two strided-batched library GEMMs per layer -> col2im.  Columns are group-major (G, P, 9*Cg) and dy is
        used through its (G, P, Cout/G) strided view, so no operand or result is permuted / copied in HBM."""
        x, offset, weight, scale, y = ctx.saved_tensors
        groups, stride, pad, has_scale, relu = ctx.cfg
        cout, cg = weight.shape[0], weight.shape[1]
        cog = cout // groups
        fused = fused_deform_backward_supported(x.shape[1], cout, groups, stride, pad)
        # WD_FUSED_DEFORM_EPILOGUE=1: both halves take the epilogue backward on their dY loads instead of the separate masking pass - measured
        # same-box 80.1 / 80.8 ms per step against 79.0 / 80.2 with the pass (the extra loads sit on the kernels' latency-critical prefetch), so off
        all_fused = fused and FUSED_DEFORM_DXOFF and os.environ.get('WD_FUSED_DEFORM_EPILOGUE', '0') == '1'
        y_act = scale_v = None
        if (has_scale or relu) and all_fused:
            y_act = _nhwc(y) if relu else None
            scale_v = scale if has_scale else None
        elif has_scale or relu:                      # epilogue backward in one pass: dy * (y > 0) * scale
            dyc = _nhwc(dy)
            n_, c_, h_, w_ = dyc.shape
            g = act_bwd(dyc.permute(0, 2, 3, 1).reshape(n_ * h_ * w_, c_), _nhwc(y).permute(0, 2, 3, 1).reshape(n_ * h_ * w_, c_) if relu else None,
                        scale if has_scale else None, relu)
            dy = g.view(n_, h_, w_, c_).permute(0, 3, 1, 2)
        dyn = _nhwc(dy).permute(0, 2, 3, 1)
        p = dyn.shape[0] * dyn.shape[1] * dyn.shape[2]
        dyg = dyn.reshape(p, groups, cog).permute(1, 0, 2)                      # (G, P, cog) view, row stride Cout
        dx = doff = dw = None
        if ctx.needs_input_grad[2]:
            if fused:
                dw = deform_dw(x, offset, dyn.contiguous(), groups, y_act, scale_v, stride)
            else:
                col = deform_im2col(x, offset, stride, pad, groups).view(groups, p, 9 * cg)
                dwg = torch.bmm(dyg.transpose(1, 2), col)                       # (G, cog, 9*cg): [g][o][k][i]
                dw = dwg.view(groups, cog, 9, cg).permute(0, 1, 3, 2).reshape(cout, cg, 3, 3)
        if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and fused and FUSED_DEFORM_DXOFF:
            dx, doff = deform_dxoff(x, offset, dyn.contiguous(), weight, groups, y_act, scale_v, stride)
        elif ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            wg = weight.view(groups, cog, cg, 9).permute(0, 1, 3, 2).reshape(groups, cog, 9 * cg)   # [g][o][k][i] (small)
            dcol = torch.bmm(dyg, wg)                                           # (G, P, 9*cg)
            dx, doff = deform_col2im(dcol, x, offset, stride, pad, groups)
        return dx, doff, dw, None, None, None, None, None, None


SPLIT_TRAIN = os.environ.get('WD_SPLIT_TRAIN', '1') != '0'     # A/B switch: 0 keeps the training graph's GEMMs / dense convolutions on the f32 library path


class ConvSplitFn(torch.autograd.Function):
    """y = act(conv2d(x, weight, bias)) for dense 3x3 / 1x1 convolutions of the training graph (FPN output convs, RPN conv, box-head convs) on the
    split-operand kernel: forward = conv_split; backward-data (stride 1) = the same kernel on the gradient with the taps flipped and the channel
    roles swapped; the weight gradient stays on the library (aten convolution_backward, MIOpen wrw): it contracts over the pixel axis, which the
    kernel's K-contiguous operand layout does not offer."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, relu):
        x = _nhwc(x)
        n, c, ks = weight.shape[0], weight.shape[1], weight.shape[2]
        y = conv_split(x, split_pack_cached(weight), n, ks, stride, pad, bias, None, relu)
        ctx.cfg = (stride, pad, bool(relu), bias is not None)
        ctx.save_for_backward(x, weight, y if relu else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        stride, pad, relu, has_bias = ctx.cfg
        dy = _nhwc(dy)
        n, c, ks = weight.shape[0], weight.shape[1], weight.shape[2]
        if relu:
            b, _, ho, wo = dy.shape
            g = act_bwd(dy.permute(0, 2, 3, 1).reshape(b * ho * wo, n), y.permute(0, 2, 3, 1).reshape(b * ho * wo, n), None, True)
            g = g.view(b, ho, wo, n).permute(0, 3, 1, 2)
        else:
            g = dy
        dx = dw = db = None
        want_dx = ctx.needs_input_grad[0]
        if want_dx and stride == 1 and conv_split_ok(g, n, c, ks, 1, ks - 1 - pad):
            # dX[ci](p) = sum over taps, co of g[co](p + pad - tap) W[co, ci, tap]: a convolution of g with the taps flipped, roles of ci / co swapped
            # (the weight is packed in that orientation where it lies: split_pack_cached 'dx')
            dx = conv_split(g, split_pack_cached(weight, 'dx'), c, ks, 1, ks - 1 - pad)
            want_dx = False
        if want_dx or ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            gi, dw, db = torch.ops.aten.convolution_backward(g, x, weight, [n] if has_bias else None, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1,
                                                             [want_dx, ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]])
            if want_dx:
                dx = gi
        return dx, dw, db, None, None, None


class LinearActFn(torch.autograd.Function):
    """y = act(a @ W.T + bias [+ residual]) for the training graph (round 4): the forward is the inference path's ONE fused library call
    (bias + ReLU on the GEMM epilogue, the residual on its beta term) instead of linear + add + relu, the backward masks the incoming
    gradient once (wd_act_bwd_f32) and hands it to both consumers - the two GEMMs and the residual branch - without a copy."""

    @staticmethod
    def forward(ctx, a, weight, bias, residual, relu):
        n, k = weight.shape
        if SPLIT_TRAIN and gemm_split_ok(a, n, bias, None if residual is None or not residual.is_contiguous() else residual):
            # round 5: the split-operand kernel (exact 3 x bf16 operand planes on the bf16 matrix cores), as at inference; a new output buffer -
            # the residual belongs to autograd
            r = None if residual is None else (residual if residual.is_contiguous() else residual.contiguous())
            y = gemm_split(a, split_pack_cached(weight), n, bias, r, relu)
        elif residual is None:
            if relu and hasattr(torch, '_addmm_activation'):
                y = torch._addmm_activation(bias, a, weight.t(), use_gelu=False)
            else:
                y = torch.addmm(bias, a, weight.t())
                if relu:
                    y = y.relu_()
        else:
            r = residual if residual.is_contiguous() else residual.contiguous()
            y = gemm_lt(a if a.is_contiguous() else a.contiguous(), weight, bias, r, relu)
        ctx.relu = bool(relu)
        ctx.has_res = residual is not None
        ctx.save_for_backward(a, weight, y if relu else a.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        a, weight, y = ctx.saved_tensors
        dy = dy if dy.is_contiguous() else dy.contiguous()
        g = act_bwd(dy, y, None, True) if ctx.relu else dy
        n, k = weight.shape
        if not ctx.needs_input_grad[0]:
            da = None
        elif SPLIT_TRAIN and gemm_split_ok(g, k):
            da = gemm_split(g, split_pack_cached(weight, 'T'), k)          # dA (M, K) = g (M, N) . (W^T)^T: W^T packed as a (K, N) weight
        else:
            da = g @ weight
        dw = g.t() @ a if ctx.needs_input_grad[1] else None
        db = g.sum(0) if ctx.needs_input_grad[2] else None
        dr = g if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return da, dw, db, dr, None


class RoiPoolFpnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, scales, pooled, min_level, canonical_level, canonical_size, *feats):
        out = roi_pool_fpn(list(feats), rois, scales, pooled, min_level, canonical_level, canonical_size)
        ctx.save_for_backward(rois)
        ctx.cfg = (list(scales), pooled, min_level, canonical_level, canonical_size, [tuple(f.shape) for f in feats])
        return out

    @staticmethod
    def backward(ctx, gout):
        rois, = ctx.saved_tensors
        scales, pooled, min_level, canonical_level, canonical_size, shapes = ctx.cfg
        gout = _nhwc(gout)
        grads = [torch.zeros(s, dtype=torch.float32, device=gout.device).contiguous(memory_format=torch.channels_last) for s in shapes]
        nl = len(grads)
        ptrs = (C.c_void_p * nl)(*[g.data_ptr() for g in grads])
        hs = (C.c_int32 * nl)(*[s[2] for s in shapes])
        ws = (C.c_int32 * nl)(*[s[3] for s in shapes])
        sc = (C.c_float * nl)(*[float(s) for s in scales])
        r = rois.shape[0]
        if r:
            _lib.check(_lib.lib().wd_roi_pool_fpn_bwd_f32(ptrs, hs, ws, sc, C.c_int(nl), C.c_int(shapes[0][1]), C.c_int(shapes[0][0]),
                                                          _p(rois.contiguous().float()), C.c_int(r), C.c_int(pooled), C.c_int(min_level),
                                                          C.c_int(canonical_level), C.c_float(canonical_size), _p(gout), _stream()),
                       'wd_roi_pool_fpn_bwd_f32')
        return (None, None, None, None, None, None) + tuple(grads)
