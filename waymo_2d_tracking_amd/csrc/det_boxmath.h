// Box2BoxTransform.apply_deltas + Boxes.clip (detectron2) as device functions shared by the decode launch
// (det_preprocess.hip) and the fused tail kernels (det_tail.hip).  Units including this header are built with
// -ffp-contract=off: the arithmetic is the torch sequence operation by operation, so results are bit-identical to it.
#pragma once
#include <hip/hip_runtime.h>

namespace wd {

// Boxes.clip: clamp(min=0, max=w|h); NaN stays NaN.  clip_w <= 0: no clipping.
__device__ __forceinline__ float4 clip_box(float4 o, float clip_w, float clip_h) {
    if (clip_w > 0.f) {
        o.x = o.x < 0.f ? 0.f : (o.x > clip_w ? clip_w : o.x); o.z = o.z < 0.f ? 0.f : (o.z > clip_w ? clip_w : o.z);
        o.y = o.y < 0.f ? 0.f : (o.y > clip_h ? clip_h : o.y); o.w = o.w < 0.f ? 0.f : (o.w > clip_h ? clip_h : o.w);
    }
    return o;
}

__device__ __forceinline__ float4 decode_box(const float4 d, const float4 b, float wx, float wy, float ww, float wh,
                                             float scale_clamp, float clip_w, float clip_h) {
    const float widths = b.z - b.x, heights = b.w - b.y;
    const float ctr_x = b.x + 0.5f * widths, ctr_y = b.y + 0.5f * heights;
    // tensor / python_scalar in torch is a multiplication by the float reciprocal of the scalar (BinaryDivTrueKernel)
    const float dx = d.x * (1.0f / wx), dy = d.y * (1.0f / wy);
    float dw = d.z * (1.0f / ww), dh = d.w * (1.0f / wh);
    dw = dw > scale_clamp ? scale_clamp : dw;             // torch.clamp(max=): NaN propagates
    dh = dh > scale_clamp ? scale_clamp : dh;
    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
    const float pw = expf(dw) * widths, ph = expf(dh) * heights;
    return clip_box(make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph), clip_w, clip_h);
}

}  // namespace wd
