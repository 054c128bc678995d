// ROIPooler (FPN level assignment + ROIAlign, aligned=True, adaptive sampling grid) for gfx950.
// Replaces detectron2's ROIPooler/ROIAlign used by the reference's Cascade R-CNN box heads
// (logs/12442/job.log:1137-1143; semantics restated in SURVEY.md App. C).
//
// Layout: features NHWC, so the 64 lanes of a wavefront read 64 consecutive channels (256 B) of one feature
// pixel per load; bilinear weights and sample coordinates are wave-uniform.  One workgroup (4 waves) per ROI:
// wave w owns channels [64w, 64w+64) (+256 strides) and walks the 7x7 bins; each output element is written once,
// coalesced.  Algorithmic bytes per ROI: unique footprint (h_f+1)(w_f+1)*C*4 + 20 B roi + 49*C*4 B out.
#include "common.h"
#include "../../include/waymodet.h"

namespace {

constexpr int kMaxLevels = 8;

struct Levels {
    const float* feat[kMaxLevels];
    int h[kMaxLevels];
    int w[kMaxLevels];
    float scale[kMaxLevels];
};

__global__ __launch_bounds__(256) void roi_pool_fpn_kernel(Levels lv, int n_levels, int C, int batch,
                                                           const float* __restrict__ rois, int n_rois, int P,
                                                           int min_level, int canonical_level, float canonical_size,
                                                           float* __restrict__ out) {
    const int r = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const float* roi = rois + 5 * (size_t)r;
    const int b = (int)roi[0];
    const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
    // level assignment (detectron2 assign_boxes_to_levels)
    const float size = sqrtf((x2 - x1) * (y2 - y1));
    int lvl = (int)floorf((float)canonical_level + log2f(size / canonical_size + 1e-8f));
    lvl = lvl < min_level ? min_level : (lvl > min_level + n_levels - 1 ? min_level + n_levels - 1 : lvl);
    const int li = lvl - min_level;
    const float* __restrict__ feat = lv.feat[li];
    const int H = lv.h[li], W = lv.w[li];
    const float scale = lv.scale[li];
    if (b < 0 || b >= batch) {                       // malformed roi: zeros
        for (int i = threadIdx.x; i < P * P * C; i += 256) out[(size_t)r * P * P * C + i] = 0.f;
        return;
    }
    feat += (size_t)b * H * W * C;
    // ROIAlign forward, aligned=True
    const float rsw = x1 * scale - 0.5f, rsh = y1 * scale - 0.5f;
    const float rew = x2 * scale - 0.5f, reh = y2 * scale - 0.5f;
    const float roi_w = rew - rsw, roi_h = reh - rsh;
    const float bin_h = roi_h / (float)P, bin_w = roi_w / (float)P;
    const int gh = (int)ceilf(roi_h / (float)P), gw = (int)ceilf(roi_w / (float)P);
    const float count = (float)((gh * gw) > 1 ? gh * gw : 1);
    // One bin ROW (7 bins) at a time: for a fixed sample row iy the 7 x gw column samples are independent, so each
    // lane keeps 7 accumulators and has 4 x 7 = 28 coalesced 256-byte loads in flight per ix step (the v1 kernel
    // walked bins one by one with 4 loads in flight and was latency bound at 6 % of the HBM roofline).
    constexpr int PMAX = 7;
    for (int cb = wave * 64; cb < C; cb += 256) {
        const int c = cb + lane;
        const bool cok = c < C;
        const float* __restrict__ fc = feat + (cok ? c : 0);
        for (int ph = 0; ph < P; ++ph) {
            for (int pw0 = 0; pw0 < P; pw0 += PMAX) {
                float acc[PMAX];
#pragma unroll
                for (int j = 0; j < PMAX; ++j) acc[j] = 0.f;
                for (int iy = 0; iy < gh; ++iy) {
                    float yy = rsh + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
                    const bool yok = !(yy < -1.0f || yy > (float)H);
                    if (yy <= 0) yy = 0;
                    int yl = (int)yy, yh;
                    if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
                    const float ly = yy - (float)yl, hy = 1.f - ly;
                    const float* __restrict__ r0 = fc + (size_t)yl * W * C;
                    const float* __restrict__ r1 = fc + (size_t)yh * W * C;
                    for (int ix = 0; ix < gw; ++ix) {
                        float v1[PMAX], v2[PMAX], v3[PMAX], v4[PMAX], wl[PMAX], wh[PMAX];
#pragma unroll
                        for (int j = 0; j < PMAX; ++j) {
                            const int pw = pw0 + j;
                            float x = rsw + (float)pw * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
                            const bool ok = yok && pw < P && !(x < -1.0f || x > (float)W);
                            if (x <= 0) x = 0;
                            int xl = (int)x, xh;
                            if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                            if (!ok) { xl = 0; xh = 0; }
                            const float lx = ok ? x - (float)xl : 0.f, hx = ok ? 1.f - lx : 0.f;
                            wl[j] = hx; wh[j] = lx;
                            v1[j] = r0[(size_t)xl * C]; v2[j] = r0[(size_t)xh * C];
                            v3[j] = r1[(size_t)xl * C]; v4[j] = r1[(size_t)xh * C];
                        }
#pragma unroll
                        for (int j = 0; j < PMAX; ++j) {
                            const float w1 = hy * wl[j], w2 = hy * wh[j], w3 = ly * wl[j], w4 = ly * wh[j];
                            acc[j] += w1 * v1[j] + w2 * v2[j] + w3 * v3[j] + w4 * v4[j];
                        }
                    }
                }
                if (cok) {
#pragma unroll
                    for (int j = 0; j < PMAX; ++j)
                        if (pw0 + j < P) out[(((size_t)r * P + ph) * P + pw0 + j) * C + c] = acc[j] / count;
                }
            }
        }
    }
}

}  // namespace

extern "C" int wd_roi_pool_fpn_f32(const float* const* feats, const int32_t* heights, const int32_t* widths,
                                   const float* scales, int n_levels, int channels, int batch, const float* rois,
                                   int n_rois, int pooled, int min_level, int canonical_level, float canonical_size,
                                   float* out, void* stream) {
    WT_TRY(wt::ensure_device());
    if (n_levels < 1 || n_levels > kMaxLevels || channels < 1 || pooled < 1) {
        wt::set_error("wd_roi_pool_fpn_f32: bad shape");
        return WT_ERR_INVALID;
    }
    if (n_rois <= 0) return WT_OK;
    Levels lv;
    for (int i = 0; i < n_levels; ++i) { lv.feat[i] = feats[i]; lv.h[i] = heights[i]; lv.w[i] = widths[i]; lv.scale[i] = scales[i]; }
    hipLaunchKernelGGL(roi_pool_fpn_kernel, dim3((unsigned)n_rois), dim3(256), 0, (hipStream_t)stream, lv, n_levels,
                       channels, batch, rois, n_rois, pooled, min_level, canonical_level, canonical_size, out);
    WT_HIP(hipGetLastError());
    return WT_OK;
}
