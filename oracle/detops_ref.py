"""TEST INFRASTRUCTURE ONLY - PyTorch (CPU, float64 unless stated) restatement of the detector custom ops, after
SURVEY.md App. C.  detectron2 0.1.3 / torchvision 0.6 are absent from /root/reference and from this image, so these
references are "parity unpinned": they restate the published op definitions (ROIAlign aligned=True with adaptive
sampling grid; deformable im2col with per-corner bounds checks; greedy NMS with IoU > thr) and pin the HIP kernels to
them.  Never imported by the product package.
"""
import math

import torch


def assign_levels(rois, min_level=2, max_level=5, canonical_level=4, canonical_size=224.0):
    size = torch.sqrt(((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])).float())
    lvl = torch.floor(canonical_level + torch.log2(size / canonical_size + 1e-8))
    return torch.clamp(lvl, min_level, max_level).long()


def _bilinear(feat, y, x):
    """feat (C,H,W) float64, scalar y, x -> (C,) ; ROIAlign rule."""
    C, H, W = feat.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return torch.zeros(C, dtype=feat.dtype)
    y = max(y, 0.0)
    x = max(x, 0.0)
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = float(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1
        x = float(xl)
    else:
        xh = xl + 1
    ly, lx = y - yl, x - xl
    hy, hx = 1.0 - ly, 1.0 - lx
    return hy * hx * feat[:, yl, xl] + hy * lx * feat[:, yl, xh] + ly * hx * feat[:, yh, xl] + ly * lx * feat[:, yh, xh]


def roi_align(feat, rois, scale, pooled=7):
    """feat (N,C,H,W); rois (R,5); aligned=True, sampling_ratio=0 -> (R,C,P,P).  float32 coordinate arithmetic like
    the kernel (coordinates are float32 in detectron2 too), float64 accumulation."""
    import numpy as np
    f32 = np.float32
    R = rois.shape[0]
    C = feat.shape[1]
    out = torch.zeros((R, C, pooled, pooled), dtype=torch.float64)
    featd = feat.double()
    for r in range(R):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = [f32(v) for v in rois[r, 1:].tolist()]
        s = f32(scale)
        rsw, rsh = f32(x1 * s - f32(0.5)), f32(y1 * s - f32(0.5))
        rew, reh = f32(x2 * s - f32(0.5)), f32(y2 * s - f32(0.5))
        rw, rh = f32(rew - rsw), f32(reh - rsh)
        bh, bw = f32(rh / f32(pooled)), f32(rw / f32(pooled))
        gh, gw = int(math.ceil(float(f32(rh / f32(pooled))))), int(math.ceil(float(f32(rw / f32(pooled)))))
        count = max(gh * gw, 1)
        for ph in range(pooled):
            for pw in range(pooled):
                acc = torch.zeros(C, dtype=torch.float64)
                for iy in range(gh):
                    y = f32(f32(rsh + f32(f32(ph) * bh)) + f32(f32(f32(iy) + f32(0.5)) * bh) / f32(gh))
                    for ix in range(gw):
                        x = f32(f32(rsw + f32(f32(pw) * bw)) + f32(f32(f32(ix) + f32(0.5)) * bw) / f32(gw))
                        acc += _bilinear(featd[b], float(y), float(x))
                out[r, :, ph, pw] = acc / count
    return out


def roi_pool_fpn(feats, rois, scales, pooled=7, min_level=2, canonical_level=4, canonical_size=224.0):
    lv = assign_levels(rois, min_level, min_level + len(feats) - 1, canonical_level, canonical_size)
    out = torch.zeros((rois.shape[0], feats[0].shape[1], pooled, pooled), dtype=torch.float64)
    for l in range(len(feats)):
        sel = torch.nonzero(lv == l + min_level).flatten()
        if len(sel):
            out[sel] = roi_align(feats[l], rois[sel], scales[l], pooled)
    return out, lv


def deform_conv3x3(x, offset, weight, groups, stride=1, pad=1, mask=None):
    """x (N,C,H,W), offset (N,18,Ho,Wo) [2k]=dy [2k+1]=dx, weight (Cout,C/groups,3,3) -> (N,Cout,Ho,Wo), float64."""
    x = x.double(); offset = offset.double(); weight = weight.double()
    N, Cin, H, W = x.shape
    Cout = weight.shape[0]
    Ho = (H + 2 * pad - 3) // stride + 1
    Wo = (W + 2 * pad - 3) // stride + 1
    ho = torch.arange(Ho, dtype=torch.float64).view(1, Ho, 1)
    wo = torch.arange(Wo, dtype=torch.float64).view(1, 1, Wo)
    cols = []
    for k in range(9):
        kh, kw = k // 3, k % 3
        hy = ho * stride - pad + kh + offset[:, 2 * k]          # (N,Ho,Wo)
        wx = wo * stride - pad + kw + offset[:, 2 * k + 1]
        inside = (hy > -1) & (wx > -1) & (hy < H) & (wx < W)
        hl = torch.floor(hy); wl = torch.floor(wx)
        lh = hy - hl; lw = wx - wl
        val = torch.zeros((N, Cin, Ho, Wo), dtype=torch.float64)
        for dh, dw, wgt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            hh = (hl + dh).long(); ww = (wl + dw).long()
            ok = inside & (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)
            hh = hh.clamp(0, H - 1); ww = ww.clamp(0, W - 1)
            idx = (hh * W + ww).view(N, 1, Ho * Wo).expand(N, Cin, Ho * Wo)
            g = torch.gather(x.view(N, Cin, H * W), 2, idx).view(N, Cin, Ho, Wo)
            val = val + g * (wgt * ok).unsqueeze(1)
        if mask is not None:
            val = val * mask[:, k].double().unsqueeze(1)
        cols.append(val)
    col = torch.stack(cols, dim=2)                                # (N,Cin,9,Ho,Wo)
    cg = Cin // groups
    cog = Cout // groups
    out = torch.zeros((N, Cout, Ho, Wo), dtype=torch.float64)
    for g in range(groups):
        c = col[:, g * cg:(g + 1) * cg].reshape(N, cg * 9, Ho * Wo)
        wg = weight[g * cog:(g + 1) * cog].reshape(cog, cg * 9)
        out[:, g * cog:(g + 1) * cog] = torch.matmul(wg, c).view(N, cog, Ho, Wo)
    return out


def nms_sorted(boxes, idxs, thr):
    """boxes (n,4) float32 sorted by descending score -> bool keep (float32 IoU arithmetic like torchvision)."""
    n = boxes.shape[0]
    b = boxes.float()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    keep = torch.ones(n, dtype=torch.bool)
    for i in range(n):
        if not keep[i]:
            continue
        if i + 1 >= n:
            break
        left = torch.maximum(b[i, 0], b[i + 1:, 0]); right = torch.minimum(b[i, 2], b[i + 1:, 2])
        top = torch.maximum(b[i, 1], b[i + 1:, 1]); bottom = torch.minimum(b[i, 3], b[i + 1:, 3])
        inter = (right - left).clamp(min=0) * (bottom - top).clamp(min=0)
        iou = inter / (area[i] + area[i + 1:] - inter)
        sup = iou > thr
        if idxs is not None:
            sup &= idxs[i + 1:] == idxs[i]
        keep[i + 1:] &= ~sup
    return keep
