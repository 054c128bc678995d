// ROIPooler (FPN level assignment + ROIAlign, aligned=True, adaptive sampling grid) for gfx950.
// Replaces detectron2's ROIPooler/ROIAlign used by the reference's Cascade R-CNN box heads
// (logs/12442/job.log:1137-1143; semantics restated in SURVEY.md App. C).
//
// Layout: features NHWC, so the 64 lanes of a wavefront read 64 consecutive channels (256 B) of one feature
// pixel per load; bilinear weights and sample coordinates are wave-uniform.  One workgroup (4 waves) per ROI:
// wave w owns channels [64w, 64w+64) (+256 strides); each output element is written once, coalesced.
// Algorithmic bytes per ROI: unique footprint (h_f+1)(w_f+1)*C*4 + 20 B roi + 49*C*4 B out.
//
// Main kernel (roi_pool_sep_kernel): average pooling of bilinear samples is SEPARABLE - a bin's value is
//   sum_r sum_q WY[ph][r] * WX[pw][q] * f[r][q] / count,   WY / WX = per-axis sums of the bilinear weights of the
// bin's samples.  The two small weight tables are built once per ROI in LDS; per bin row the wave makes ONE pass
// over the footprint rows it touches (t[q] = sum_r WY[ph][r] f[r][q], a handful of independent coalesced loads per
// column, 16 in flight per lane) and folds each column sum into the 7 bins.  Every footprint pixel is read ~1.5x instead
// of 4*g*g/(g+1)^2 ... times, with >= 12 loads in flight per lane.  ROIs wider/taller than 64 feature pixels on
// their level fall back to the direct kernel below (roi_pool_fpn_kernel).
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "../../include/waymodet.h"

namespace {

constexpr int kMaxLevels = 8;
#ifndef WD_ROI_QB
#define WD_ROI_QB 8
#endif

struct Levels {
    const float* feat[kMaxLevels];
    int h[kMaxLevels];
    int w[kMaxLevels];
    float scale[kMaxLevels];
};

__global__ __launch_bounds__(256) void roi_pool_fpn_kernel(Levels lv, int n_levels, int C, int batch,
                                                           const float* __restrict__ rois, int n_rois, int P,
                                                           int min_level, int canonical_level, float canonical_size,
                                                           float* __restrict__ out, const int* __restrict__ only_flagged) {
    const int r = blockIdx.x;
    if (only_flagged && !only_flagged[r]) return;         // already done by the separable kernel
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

constexpr int kMaxFoot = 64;      // footprint rows / columns handled by the separable kernel (larger: direct kernel)

__global__ __launch_bounds__(256) void roi_pool_sep_kernel(Levels lv, int n_levels, int C, int batch,
                                                           const float* __restrict__ rois, int n_rois, int P,
                                                           int min_level, int canonical_level, float canonical_size,
                                                           float* __restrict__ out, int* __restrict__ fallback_flags) {
    __shared__ float wy[7][kMaxFoot];
    __shared__ float wx[7][kMaxFoot];
    __shared__ int lo_hi[2][7][2];          // [axis][bin][first, last] non-zero index
    const int r = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const float* roi = rois + 5 * (size_t)r;
    const int b = (int)roi[0];
    const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
    const float size = sqrtf((x2 - x1) * (y2 - y1));
    int lvl = (int)floorf((float)canonical_level + log2f(size / canonical_size + 1e-8f));
    lvl = lvl < min_level ? min_level : (lvl > min_level + n_levels - 1 ? min_level + n_levels - 1 : lvl);
    const int li = lvl - min_level;
    const int H = lv.h[li], W = lv.w[li];
    const float scale = lv.scale[li];
    if (fallback_flags && threadIdx.x == 0) fallback_flags[r] = 0;
    if (b < 0 || b >= batch) {
        for (int i = threadIdx.x; i < P * P * C; i += 256) out[(size_t)r * P * P * C + i] = 0.f;
        return;
    }
    const float* __restrict__ feat = lv.feat[li] + (size_t)b * H * W * C;
    const float rsw = x1 * scale - 0.5f, rsh = y1 * scale - 0.5f;
    const float rew = x2 * scale - 0.5f, reh = y2 * scale - 0.5f;
    const float roi_w = rew - rsw, roi_h = reh - rsh;
    const float bin_h = roi_h / (float)P, bin_w = roi_w / (float)P;
    const int gh = (int)ceilf(roi_h / (float)P), gw = (int)ceilf(roi_w / (float)P);
    const float count = (float)((gh * gw) > 1 ? gh * gw : 1);
    // footprint origin = the low corner of the first sample (after the ROIAlign clamps)
    auto low_index = [](float v, int n) {
        if (v <= 0) v = 0;
        int l = (int)v;
        return l >= n - 1 ? n - 1 : l;
    };
    const int r_lo = low_index(rsh + .5f * bin_h / (float)(gh > 0 ? gh : 1), H);
    const int q_lo = low_index(rsw + .5f * bin_w / (float)(gw > 0 ? gw : 1), W);
    const float y_last = rsh + (float)(P - 1) * bin_h + ((float)(gh > 0 ? gh - 1 : 0) + .5f) * bin_h / (float)(gh > 0 ? gh : 1);
    const float x_last = rsw + (float)(P - 1) * bin_w + ((float)(gw > 0 ? gw - 1 : 0) + .5f) * bin_w / (float)(gw > 0 ? gw : 1);
    const int r_hi = low_index(y_last, H) + 1 < H ? low_index(y_last, H) + 1 : H - 1;
    const int q_hi = low_index(x_last, W) + 1 < W ? low_index(x_last, W) + 1 : W - 1;
    const int nrows = r_hi - r_lo + 1, ncols = q_hi - q_lo + 1;
    if (P != 7 || nrows > kMaxFoot || ncols > kMaxFoot || nrows < 1 || ncols < 1) {      // rare: direct kernel does it
        if (fallback_flags && threadIdx.x == 0) fallback_flags[r] = 1;
        return;
    }
    // ---- per-axis weight tables (thread ph builds row ph sequentially: deterministic sums) ----
    for (int i = threadIdx.x; i < 7 * kMaxFoot; i += 256) { (&wy[0][0])[i] = 0.f; (&wx[0][0])[i] = 0.f; }
    __syncthreads();
    if (threadIdx.x < 14) {
        const int axis = threadIdx.x / 7, p = threadIdx.x % 7;
        const int g = axis ? gw : gh, N = axis ? W : H, lo = axis ? q_lo : r_lo;
        const float start = axis ? rsw : rsh, bin = axis ? bin_w : bin_h;
        float* wrow = axis ? wx[p] : wy[p];
        int first = kMaxFoot, last = -1;
        for (int i = 0; i < g; ++i) {
            float v = start + (float)p * bin + ((float)i + .5f) * bin / (float)g;
            if (v < -1.0f || v > (float)N) continue;
            if (v <= 0) v = 0;
            int l = (int)v, h;
            if (l >= N - 1) { h = l = N - 1; v = (float)l; } else h = l + 1;
            const float fl = v - (float)l;
            wrow[l - lo] += 1.f - fl;
            wrow[h - lo] += fl;
            first = (l - lo) < first ? (l - lo) : first;
            last = (h - lo) > last ? (h - lo) : last;
        }
        lo_hi[axis][p][0] = first;
        lo_hi[axis][p][1] = last;
    }
    __syncthreads();
    for (int cb = wave * 64; cb < C; cb += 256) {
        const int c = cb + lane;
        const bool cok = c < C;
        const float* __restrict__ fc = feat + ((size_t)r_lo * W + q_lo) * C + (cok ? c : 0);
        for (int ph = 0; ph < 7; ++ph) {
            const int ra = lo_hi[0][ph][0], rb = lo_hi[0][ph][1];
            float bins[7];
#pragma unroll
            for (int pw = 0; pw < 7; ++pw) bins[pw] = 0.f;
            // row pass over 8 columns x 2 rows at a time (16 independent coalesced loads in flight per lane); each
            // column sum t = sum_r WY[ph][r] f[r][q] is folded straight into the 7 bins with WX[pw][q]
            for (int q0 = 0; q0 < ncols; q0 += 8) {
                float acc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = 0.f;
                for (int rr = ra; rr <= rb; rr += 2) {
                    const int r1 = (rr + 1 <= rb) ? rr + 1 : rb;
                    const float w0 = wy[ph][rr], w1 = (rr + 1 <= rb) ? wy[ph][r1] : 0.f;
                    float v0[8], v1[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int qq = (q0 + u < ncols) ? q0 + u : ncols - 1;
                        v0[u] = fc[((size_t)rr * W + qq) * C];
                        v1[u] = fc[((size_t)r1 * W + qq) * C];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[u] += w0 * v0[u] + w1 * v1[u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int qq = (q0 + u < ncols) ? q0 + u : ncols - 1;
                    const float tv = (q0 + u < ncols) ? acc[u] : 0.f;
#pragma unroll
                    for (int pw = 0; pw < 7; ++pw) bins[pw] += wx[pw][qq] * tv;
                }
            }
            if (cok) {
#pragma unroll
                for (int pw = 0; pw < 7; ++pw) out[(((size_t)r * 7 + ph) * 7 + pw) * C + c] = bins[pw] / count;
            }
        }
    }
}

// Row-unit kernel (default for pooled == 7, C % 4 == 0): one WAVE per (ROI, bin row ph) - 7000 independent units for
// 1000 ROIs instead of 1000 workgroups of very different sizes - and 16 bytes per lane: a wave-load covers 256
// channels (1 KiB, the whole pixel for C = 256), 4x fewer instructions per byte than the dword version above.  The
// wave builds WY[ph][.] and WX[0..6][.] itself (lanes 0..7, sequential sums: deterministic), walks the footprint rows of
// its bin row once with 16 x 1 KiB loads in flight, and writes the 7 x C outputs of (ROI, ph) as one contiguous run.
// Processing order of the ROIs: by FPN level, then by 16-pixel rows of the level's feature map, then by x.  One workgroup,
// bitonic sort of (key << 32 | index) in LDS (n <= 8192).  Only the ORDER of the work changes - outputs stay in ROI order.
__global__ __launch_bounds__(1024) void roi_order_kernel(Levels lv, int n_levels, const float* __restrict__ rois, int n_rois,
                                                        int min_level, int canonical_level, float canonical_size,
                                                        int* __restrict__ order) {
    extern __shared__ unsigned long long okeys[];
    int p = 2;
    while (p < n_rois) p <<= 1;
    for (int t = threadIdx.x; t < p; t += 1024) {
        unsigned long long key = ~0ull;
        if (t < n_rois) {
            const float* roi = rois + 5 * (size_t)t;
            const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
            const float size = sqrtf((x2 - x1) * (y2 - y1));
            int lvl = (int)floorf((float)canonical_level + log2f(size / canonical_size + 1e-8f));
            lvl = lvl < min_level ? min_level : (lvl > min_level + n_levels - 1 ? min_level + n_levels - 1 : lvl);
            const int li = lvl - min_level;
            const float sc = lv.scale[li];
            float cx = 0.5f * (x1 + x2) * sc, cy = 0.5f * (y1 + y2) * sc;
            cx = cx > 0.f ? (cx < 8191.f ? cx : 8191.f) : 0.f;        // NaN -> 0
            cy = cy > 0.f ? (cy < 8191.f ? cy : 8191.f) : 0.f;
            const unsigned k = ((unsigned)li << 26) | (((unsigned)cy >> 4) << 13) | (unsigned)cx;
            key = ((unsigned long long)k << 32) | (unsigned)t;
        }
        okeys[t] = key;
    }
    for (int size = 2; size <= p; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (p >> 1); t += 1024) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const unsigned long long a = okeys[i], b = okeys[j];
                const bool asc = (i & size) == 0;
                if ((a > b) == asc) { okeys[i] = b; okeys[j] = a; }
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n_rois; t += 1024) order[t] = (int)(okeys[t] & 0xffffffffull);
}

__global__ __launch_bounds__(256) void roi_pool_row_kernel(Levels lv, int n_levels, int C, int batch,
                                                           const float* __restrict__ rois, int n_rois,
                                                           int min_level, int canonical_level, float canonical_size,
                                                           float* __restrict__ out, int* __restrict__ fallback_flags,
                                                           const int* __restrict__ order) {
    constexpr int QB = WD_ROI_QB;                   // footprint columns per pass (2 QB float4 loads in flight per lane)
    __shared__ float tabs[4][8][kMaxFoot];          // per wave: [0] = WY[ph], [1 + pw] = WX[pw]
    __shared__ int lohi[4][8][2];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each with its own L2: give every XCD a CONTIGUOUS eighth
    // of the (spatially sorted) unit list, so that ROIs overlapping in the feature maps meet in one L2 at about the same time
    const int per_xcd = ((int)gridDim.x + 7) >> 3;
    const int block = order ? ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int unit = block * 4 + wave;
    if (unit >= n_rois * 7 || (order && block >= ((n_rois * 7 + 3) >> 2))) return;
    const int slot = unit / 7, ph = unit - 7 * slot;
    const int r = order ? order[slot] : slot;
    float (*tab)[kMaxFoot] = tabs[wave];
    const float* roi = rois + 5 * (size_t)r;
    const int b = (int)roi[0];
    const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
    const float size = sqrtf((x2 - x1) * (y2 - y1));
    int lvl = (int)floorf((float)canonical_level + log2f(size / canonical_size + 1e-8f));
    lvl = lvl < min_level ? min_level : (lvl > min_level + n_levels - 1 ? min_level + n_levels - 1 : lvl);
    const int li = lvl - min_level;
    const int H = lv.h[li], W = lv.w[li];
    const float scale = lv.scale[li];
    float* orow = out + ((size_t)r * 7 + ph) * 7 * C;
    if (b < 0 || b >= batch) {                       // malformed roi: zeros
        for (int i = lane; i < 7 * C; i += 64) orow[i] = 0.f;
        if (lane == 0) fallback_flags[r] = 0;
        return;
    }
    const float* __restrict__ feat = lv.feat[li] + (size_t)b * H * W * C;
    const float rsw = x1 * scale - 0.5f, rsh = y1 * scale - 0.5f;
    const float rew = x2 * scale - 0.5f, reh = y2 * scale - 0.5f;
    const float roi_w = rew - rsw, roi_h = reh - rsh;
    const float bin_h = roi_h / 7.f, bin_w = roi_w / 7.f;
    const int gh = (int)ceilf(roi_h / 7.f), gw = (int)ceilf(roi_w / 7.f);
    const float count = (float)((gh * gw) > 1 ? gh * gw : 1);
    auto low_index = [](float v, int n) {
        if (v <= 0) v = 0;
        int l = (int)v;
        return l >= n - 1 ? n - 1 : l;
    };
    const int r_lo = low_index(rsh + .5f * bin_h / (float)(gh > 0 ? gh : 1), H);
    const int q_lo = low_index(rsw + .5f * bin_w / (float)(gw > 0 ? gw : 1), W);
    const float y_last = rsh + 6.f * bin_h + ((float)(gh > 0 ? gh - 1 : 0) + .5f) * bin_h / (float)(gh > 0 ? gh : 1);
    const float x_last = rsw + 6.f * bin_w + ((float)(gw > 0 ? gw - 1 : 0) + .5f) * bin_w / (float)(gw > 0 ? gw : 1);
    const int r_hi = low_index(y_last, H) + 1 < H ? low_index(y_last, H) + 1 : H - 1;
    const int q_hi = low_index(x_last, W) + 1 < W ? low_index(x_last, W) + 1 : W - 1;
    const int nrows = r_hi - r_lo + 1, ncols = q_hi - q_lo + 1;
    if (lane == 0) fallback_flags[r] = 0;
    if (nrows > kMaxFoot || ncols > kMaxFoot || nrows < 1 || ncols < 1) {
        // rare (whole-image boxes: footprint beyond the 64 x 64 weight tables): this wave does its bin row by direct bilinear
        // sampling, sample by sample (round 3: the separate fallback launch - 1000 mostly idle workgroups, 4.8 - 7.6 us per call -
        // is gone).  Same sample positions, weights and accumulation order as roi_pool_fpn_kernel.
        for (int cb = 0; cb < C; cb += 256) {
            const int c = cb + lane * 4;
            const bool cok = c < C;
            const float* __restrict__ fc = feat + (cok ? c : 0);
            float4 acc[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
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
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        float x = rsw + (float)j * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
                        const bool ok = yok && !(x < -1.0f || x > (float)W);
                        if (x <= 0) x = 0;
                        int xl = (int)x, xh;
                        if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                        if (!ok) { xl = 0; xh = 0; }
                        const float lx = ok ? x - (float)xl : 0.f, hx = ok ? 1.f - lx : 0.f;
                        const float4 v1 = *reinterpret_cast<const float4*>(r0 + (size_t)xl * C), v2 = *reinterpret_cast<const float4*>(r0 + (size_t)xh * C);
                        const float4 v3 = *reinterpret_cast<const float4*>(r1 + (size_t)xl * C), v4 = *reinterpret_cast<const float4*>(r1 + (size_t)xh * C);
                        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                        acc[j].x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x; acc[j].y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
                        acc[j].z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z; acc[j].w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
                    }
                }
            }
            if (cok) {
#pragma unroll
                for (int j = 0; j < 7; ++j)
                    *reinterpret_cast<float4*>(orow + (size_t)j * C + c) =
                        make_float4(acc[j].x / count, acc[j].y / count, acc[j].z / count, acc[j].w / count);
            }
        }
        return;
    }
    for (int i = lane; i < 8 * kMaxFoot; i += 64) (&tab[0][0])[i] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 8) {
        const int axis = lane > 0, p = axis ? lane - 1 : ph;
        const int g = axis ? gw : gh, N = axis ? W : H, lo = axis ? q_lo : r_lo;
        const float start = axis ? rsw : rsh, bin = axis ? bin_w : bin_h;
        float* wrow = tab[lane];
        int first = kMaxFoot, last = -1;
        for (int i = 0; i < g; ++i) {
            float v = start + (float)p * bin + ((float)i + .5f) * bin / (float)g;
            if (v < -1.0f || v > (float)N) continue;
            if (v <= 0) v = 0;
            int l = (int)v, h;
            if (l >= N - 1) { h = l = N - 1; v = (float)l; } else h = l + 1;
            const float fl = v - (float)l;
            wrow[l - lo] += 1.f - fl;
            wrow[h - lo] += fl;
            first = (l - lo) < first ? (l - lo) : first;
            last = (h - lo) > last ? (h - lo) : last;
        }
        lohi[wave][lane][0] = first;
        lohi[wave][lane][1] = last;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int ra = lohi[wave][0][0], rb = lohi[wave][0][1];
    const float* wy = tab[0];
    for (int cb = 0; cb < C; cb += 256) {
        const int c = cb + lane * 4;
        const bool cok = c < C;
        const float* __restrict__ fc = feat + ((size_t)r_lo * W + q_lo) * C + (cok ? c : 0);
        float4 bins[7];
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) bins[pw] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q0 = 0; q0 < ncols; q0 += QB) {
            float4 acc[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int rr = ra; rr <= rb; rr += 2) {
                const int r1 = (rr + 1 <= rb) ? rr + 1 : rb;
                const float w0 = wy[rr], w1 = (rr + 1 <= rb) ? wy[r1] : 0.f;
                float4 v0[QB], v1[QB];
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    const int qq = (q0 + u < ncols) ? q0 + u : ncols - 1;
                    v0[u] = *reinterpret_cast<const float4*>(fc + ((size_t)rr * W + qq) * C);
                    v1[u] = *reinterpret_cast<const float4*>(fc + ((size_t)r1 * W + qq) * C);
                }
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    acc[u].x += w0 * v0[u].x + w1 * v1[u].x; acc[u].y += w0 * v0[u].y + w1 * v1[u].y;
                    acc[u].z += w0 * v0[u].z + w1 * v1[u].z; acc[u].w += w0 * v0[u].w + w1 * v1[u].w;
                }
            }
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                const int qq = (q0 + u < ncols) ? q0 + u : ncols - 1;
                const bool ok = q0 + u < ncols;
#pragma unroll
                for (int pw = 0; pw < 7; ++pw) {
                    const float wxv = ok ? tab[1 + pw][qq] : 0.f;
                    bins[pw].x += wxv * acc[u].x; bins[pw].y += wxv * acc[u].y;
                    bins[pw].z += wxv * acc[u].z; bins[pw].w += wxv * acc[u].w;
                }
            }
        }
        if (cok) {
#pragma unroll
            for (int pw = 0; pw < 7; ++pw)
                *reinterpret_cast<float4*>(orow + (size_t)pw * C + c) =
                    make_float4(bins[pw].x / count, bins[pw].y / count, bins[pw].z / count, bins[pw].w / count);
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
    // The separable kernel handles ROIs whose footprint fits 64 x 64 feature pixels (all but degenerate whole-image
    // boxes); it flags the rest, which the direct kernel then processes (it exits immediately for unflagged ROIs).
    // scratch ([fallback flags | processing order]) per stream: launches on different streams may be in flight concurrently
    // (two frames' graphs), and a buffer baked into a captured graph must stay valid -> buffers are never freed or moved;
    // a stream that later needs more rows gets an additional, larger buffer
    struct Scratch { int* p; int rows; };                        // 4 * rows ints: flags [0, 2 rows), order [2 rows, 3 rows)
    static std::mutex mu;
    static std::unordered_map<void*, std::vector<Scratch>> scratch;
    int* flags = nullptr;
    int flags_cap = 0;
    if (pooled == 7) {
        std::lock_guard<std::mutex> lock(mu);
        auto& list = scratch[stream];
        int rows = 0;
        for (const Scratch& sc : list)
            if (sc.rows >= n_rois) { flags = sc.p; rows = sc.rows; break; }
        if (!flags) {
            WT_HIP(hipMalloc(&flags, sizeof(int) * (size_t)n_rois * 4));        // (not inside a stream capture: warm up first)
            rows = n_rois;
            list.push_back({flags, rows});
        }
        flags_cap = 2 * rows;
        int* order = nullptr;
        // WD_ROI_ORDER=1: spatially sorted processing order + one contiguous eighth of it per XCD.  Measured on MI355X (1000 ROIs,
        // profiles/r02_hbm_rooflines_roi_ordered.json): L2->fabric fetch traffic 375 -> 184 MB (the unique footprint is 136 MB), but
        // 105 -> 124 us: at this size the kernel is bound by load latency / occupancy (27 waves per CU in total), not by HBM
        // bytes, and the sort adds a launch.  Off by default.
        const char* om = getenv("WD_ROI_ORDER");
        if (n_rois >= 64 && n_rois <= 8192 && om && om[0] == '1') {
            order = flags + flags_cap;
            hipLaunchKernelGGL(roi_order_kernel, dim3(1), dim3(1024), (size_t)8192 * 8, (hipStream_t)stream, lv, n_levels, rois,
                               n_rois, min_level, canonical_level, canonical_size, order);
        }
        const char* mode = getenv("WD_ROI_KERNEL");             // experiments: "sep" = one workgroup per ROI
        const bool row_path = (channels & 3) == 0 && ((uintptr_t)out & 15) == 0 && !(mode && strcmp(mode, "sep") == 0);
        if (row_path)
            hipLaunchKernelGGL(roi_pool_row_kernel, dim3((unsigned)(order ? (((n_rois * 7 + 3) / 4 + 7) / 8 * 8) : (n_rois * 7 + 3) / 4)),
                               dim3(256), 0, (hipStream_t)stream, lv, n_levels, channels, batch, rois, n_rois, min_level,
                               canonical_level, canonical_size, out, flags, (const int*)order);
        else
            hipLaunchKernelGGL(roi_pool_sep_kernel, dim3((unsigned)n_rois), dim3(256), 0, (hipStream_t)stream, lv, n_levels,
                               channels, batch, rois, n_rois, pooled, min_level, canonical_level, canonical_size, out, flags);
        if (!row_path)      // the one-workgroup-per-ROI kernel still flags its large ROIs for the direct kernel
            hipLaunchKernelGGL(roi_pool_fpn_kernel, dim3((unsigned)n_rois), dim3(256), 0, (hipStream_t)stream, lv, n_levels,
                               channels, batch, rois, n_rois, pooled, min_level, canonical_level, canonical_size, out,
                               (const int*)flags);
    } else {
        hipLaunchKernelGGL(roi_pool_fpn_kernel, dim3((unsigned)n_rois), dim3(256), 0, (hipStream_t)stream, lv, n_levels,
                           channels, batch, rois, n_rois, pooled, min_level, canonical_level, canonical_size, out,
                           (const int*)nullptr);
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}
