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
#include <type_traits>
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


// ---- round 4: one workgroup per ROI --------------------------------------------------------------------------------------
// What bounded the row kernel above (tools/roi_probe.py: 41 of its 69 us remain when every byte is a cache hit) is on-chip work:
// 7 waves per ROI each repeat the ROI arithmetic and build their own weight tables, the column tail and odd row counts are padded
// with loads whose weight is 0 (1.5x the useful 1-KiB loads through the 64 B/clk vector-memory path), and every column sum is folded
// into all 7 bins although at most 3 of them have a non-zero x weight (28 of the 46 FMAs per column).  Here:
//   * the workgroup (4 waves) computes the ROI geometry once (uniform: scalar loads) and builds WY[7][.], WX[7][.] once;
//   * wave w walks bin rows w and w + 4; loads are issued for exactly the rows / columns that carry weight (uniform predicates);
//   * a column sum goes into the 3 bins starting at the first one with a non-zero weight (wave-uniform switch); ROIs whose bins
//     are narrower than a pixel (a column can feed more than 3 bins) take the dense fold;
//   * ROIs beyond the 64 x 64 tables are sampled directly, bin row by bin row, as before.
// Processing order (optional `order`): ROIs bucketed by (level, 16-row band, 8-column cell) with one LDS counting sort (the bitonic
// sort of round 2 cost 14.7 us); every XCD gets a contiguous eighth of that list, so overlapping footprints meet in one L2.
__global__ __launch_bounds__(1024) void roi_bucket_order_kernel(Levels lv, int n_levels, const float* __restrict__ rois, int n_rois,
                                                               int min_level, int canonical_level, float canonical_size,
                                                               int* __restrict__ order) {
    constexpr int NB = 2048;                       // 2 bits level | 6 bits row band | 3 bits column cell: 2 buckets per thread
    __shared__ int hist[NB];
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    hist[tid] = 0; hist[tid + 1024] = 0;
    __syncthreads();
    constexpr int PER = 8;                         // n_rois <= 8192
    int key[PER], rank[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int t = tid + 1024 * j;
        key[j] = -1;
        if (t < n_rois) {
            const float* roi = rois + 5 * (size_t)t;
            const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
            const float size = sqrtf((x2 - x1) * (y2 - y1));
            int lvl = (int)floorf((float)canonical_level + log2f(size / canonical_size + 1e-8f));
            lvl = lvl < min_level ? min_level : (lvl > min_level + n_levels - 1 ? min_level + n_levels - 1 : lvl);
            const int li = lvl - min_level;
            const float sc = lv.scale[li];
            float cx = 0.5f * (x1 + x2) * sc, cy = 0.5f * (y1 + y2) * sc;
            cx = cx > 0.f ? cx : 0.f;              // NaN -> 0
            cy = cy > 0.f ? cy : 0.f;
            // bands / cells relative to the level's own size: 64 bands x 8 cells whatever the resolution
            int band = (int)(cy * 64.f / (float)(lv.h[li] > 0 ? lv.h[li] : 1)), cell = (int)(cx * 8.f / (float)(lv.w[li] > 0 ? lv.w[li] : 1));
            band = band > 63 ? 63 : band;
            cell = cell > 7 ? 7 : cell;
            // serpentine: odd bands run right to left, so consecutive ROIs stay neighbours at the band ends
            key[j] = ((li & 3) << 9) | (band << 3) | ((band & 1) ? 7 - cell : cell);
#ifdef ROI_KEY_LPT      // experiment: largest footprint first
            { const float fp = ((x2 - x1) * sc + 2.f) * ((y2 - y1) * sc + 2.f);
              int k2 = (int)fp; k2 = k2 > NB - 1 ? NB - 1 : (k2 < 0 ? 0 : k2); key[j] = NB - 1 - k2; }
#endif
            rank[j] = atomicAdd(&hist[key[j]], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the counts: 2 per thread, wave scan, then the 16 wave totals
    const int c0 = hist[2 * tid], c1 = hist[2 * tid + 1];
    const int sum = c0 + c1;
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int base = inc - sum;
    for (int w2 = 0; w2 < wave; ++w2) base += wave_tot[w2];
    hist[2 * tid] = base; hist[2 * tid + 1] = base + c0;           // (every thread rewrites only the two counts it has read itself)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (key[j] >= 0) order[hist[key[j]] + rank[j]] = tid + 1024 * j;
}

// a pointer into global memory every lane holds the same value of -> scalar registers, global address space (the loads then take
// the `global_load v, v_offset, s[base]` form; a plain integer -> pointer cast would make them flat loads)
using roi_gptr = const __attribute__((address_space(1))) char*;
__device__ __forceinline__ roi_gptr roi_uniform_ptr(const void* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (roi_gptr)(((unsigned long long)hi << 32) | lo);
}
using roi_f4v = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ float4 roi_ld16(roi_gptr p) {          // (HIP's float4 class cannot be read through an address-space pointer)
    const roi_f4v v = *(const __attribute__((address_space(1))) roi_f4v*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ void roi_fma4(float4& a, float w, const float4& v) {
    a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; a.w += w * v.w;
}

#ifndef ROI_NW
#define ROI_NW 7                  // waves per workgroup: one per bin row (4: waves take rows w and w + 4, measured slower on scattered ROIs)
#endif
#ifndef ROI_QB
#define ROI_QB 8                  // footprint columns per block: 2 QB loads of 1 KiB in flight per wave
#endif
#ifndef ROI_MINW
#define ROI_MINW 2
#endif
__global__ __launch_bounds__(64 * ROI_NW, ROI_MINW) void roi_pool_wg_kernel(Levels lv, int n_levels, int C, int batch,
                                                          const float* __restrict__ rois, int n_rois,
                                                          int min_level, int canonical_level, float canonical_size,
                                                          float* __restrict__ out, const int* __restrict__ order) {
    constexpr int QB = ROI_QB, NW = ROI_NW, NT = 64 * NW;
    __shared__ float wy[7][kMaxFoot];
    __shared__ float wx[8][kMaxFoot];                // row 7 stays zero: the sliding window may look one bin past the last
    __shared__ int lohi_y[7][2];
    __shared__ int col_pa[kMaxFoot];
    __shared__ int any_wide;
    __shared__ __attribute__((aligned(16))) float done_lds[ROI_NW * 7 * 64 * 4];     // per wave: the 7 finished bin sums of the current bin row
#ifdef ROI_LDS_PAD
    __shared__ int lds_pad[ROI_LDS_PAD / 4];         // experiments: fewer resident workgroups per CU
    if (n_rois < 0) lds_pad[threadIdx.x] = n_rois;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // with an order list: XCD x (blockIdx % 8) takes the x-th contiguous eighth of it
    const int per_xcd = ((int)gridDim.x + 7) >> 3;
#ifdef ROI_KEY_LPT
    const int slot = (int)blockIdx.x; (void)per_xcd;
#else
    const int slot = order ? ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
#endif
    if (slot >= n_rois) return;
    const int r = order ? order[slot] : slot;
    const float* roi = rois + 5 * (size_t)r;
    const int b = (int)roi[0];
    const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
    const float size = sqrtf((x2 - x1) * (y2 - y1));
    int lvl = (int)floorf((float)canonical_level + log2f(size / canonical_size + 1e-8f));
    lvl = lvl < min_level ? min_level : (lvl > min_level + n_levels - 1 ? min_level + n_levels - 1 : lvl);
    // everything below is the same for all threads of the workgroup: told to the compiler (scalar registers, scalar loads)
    const int li = __builtin_amdgcn_readfirstlane(lvl - min_level);
    const int H = lv.h[li], W = lv.w[li];
    const float scale = lv.scale[li];
    float* obase = out + (size_t)r * 49 * C;
    if (b < 0 || b >= batch) {                       // malformed roi: zeros
        for (int i = tid; i < 49 * C; i += NT) obase[i] = 0.f;
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
    const int r_lo = __builtin_amdgcn_readfirstlane(low_index(rsh + .5f * bin_h / (float)(gh > 0 ? gh : 1), H));
    const int q_lo = __builtin_amdgcn_readfirstlane(low_index(rsw + .5f * bin_w / (float)(gw > 0 ? gw : 1), W));
    const float y_last = rsh + 6.f * bin_h + ((float)(gh > 0 ? gh - 1 : 0) + .5f) * bin_h / (float)(gh > 0 ? gh : 1);
    const float x_last = rsw + 6.f * bin_w + ((float)(gw > 0 ? gw - 1 : 0) + .5f) * bin_w / (float)(gw > 0 ? gw : 1);
    const int r_hi = low_index(y_last, H) + 1 < H ? low_index(y_last, H) + 1 : H - 1;
    const int q_hi = low_index(x_last, W) + 1 < W ? low_index(x_last, W) + 1 : W - 1;
    const int nrows = __builtin_amdgcn_readfirstlane(r_hi - r_lo + 1), ncols = __builtin_amdgcn_readfirstlane(q_hi - q_lo + 1);
    bool direct = nrows > kMaxFoot || ncols > kMaxFoot || nrows < 1 || ncols < 1;
    if (!direct) {
        // ---- weight tables, once per ROI (thread p / 7 + p builds one row sequentially: deterministic sums) ----
        for (int i = tid; i < 8 * kMaxFoot; i += NT) {
            (&wx[0][0])[i] = 0.f;
            if (i < 7 * kMaxFoot) (&wy[0][0])[i] = 0.f;
        }
        if (tid == 0) any_wide = 0;
        __syncthreads();
        if (tid < 14) {
            const int axis = tid / 7, p = tid - 7 * axis;
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
            if (!axis) { lohi_y[p][0] = first; lohi_y[p][1] = last; }
        }
        __syncthreads();
        if (tid < ncols) {
            int pa = 7, pb = -1;
#pragma unroll
            for (int pw = 0; pw < 7; ++pw)
                if (wx[pw][tid] != 0.f) { pa = pw < pa ? pw : pa; pb = pw; }
            if (pb - pa > 2) any_wide = 1;             // bins narrower than a pixel: a column feeds more than 3 of them
            col_pa[tid] = pa;                          // 7 = no weight at all (column between the samples of an out-of-image stretch)
        }
        __syncthreads();
        direct = any_wide != 0;
    }
    if (direct) {
        // rare: footprints beyond the 64 x 64 weight tables (whole-image boxes) and ROIs a few pixels wide: direct bilinear sampling,
        // wave w does bin rows w and w + 4, one bin at a time (this path must not set the register budget of the kernel).  Same sample
        // positions, weights and accumulation order as roi_pool_fpn_kernel.
        for (int ph = wave; ph < 7; ph += NW) {
            float* orow = obase + (size_t)ph * 7 * C;
            for (int cb = 0; cb < C; cb += 256) {
                const int c = cb + lane * 4;
                const bool cok = c < C;
                const float* __restrict__ fc = feat + (cok ? c : 0);
#pragma unroll 1
                for (int j = 0; j < 7; ++j) {
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int iy = 0; iy < gh; ++iy) {
                        float yy = rsh + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
                        const bool yok = !(yy < -1.0f || yy > (float)H);
                        if (yy <= 0) yy = 0;
                        int yl = (int)yy, yh;
                        if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
                        const float ly = yy - (float)yl, hy = 1.f - ly;
                        const float* __restrict__ r0 = fc + (size_t)yl * W * C;
                        const float* __restrict__ r1 = fc + (size_t)yh * W * C;
#pragma unroll 2
                        for (int ix = 0; ix < gw; ++ix) {
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
                            acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x; acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
                            acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z; acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
                        }
                    }
                    if (cok)
                        *reinterpret_cast<float4*>(orow + (size_t)j * C + c) = make_float4(acc.x / count, acc.y / count, acc.z / count, acc.w / count);
                }
            }
        }
        return;
    }
    const unsigned row_b32 = (unsigned)W * (unsigned)C * 4u, col_b32 = (unsigned)C * 4u;
    // uniform base (scalar registers) + one 32-bit lane offset per load: `global_load_dwordx4 v, v_off, s[base]`
    const roi_gptr ubase = roi_uniform_ptr(feat + ((size_t)r_lo * W + q_lo) * C);
    float4* done = reinterpret_cast<float4*>(done_lds) + (size_t)wave * 7 * 64 + lane;     // [bin][lane] of this wave
    for (int ph = wave; ph < 7; ph += NW) {
        const int ra = __builtin_amdgcn_readfirstlane(lohi_y[ph][0]), rb = __builtin_amdgcn_readfirstlane(lohi_y[ph][1]);
        const float* wyr = wy[ph];
        float* orow = obase + (size_t)ph * 7 * C;
        for (int cb = 0; cb < C; cb += 256) {
            const int c = cb + lane * 4;
            const bool cok = c < C;
            const unsigned lane_off = (unsigned)(cok ? c : 0) * 4u;
            // the three bins that can still receive weight: bins `cur`, cur + 1, cur + 2 (the first bin with weight in a column never
            // decreases from left to right); a bin that falls out of the window is complete and is written at once
            float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0, b2 = b0;
            int cur = 0;
            auto retire_until = [&](int pa) {           // wave-uniform trip count; the sums wait in LDS for the one division at the end
                while (cur < pa && cur < 7) {
                    done[cur * 64] = b0;
                    b0 = b1; b1 = b2; b2 = make_float4(0.f, 0.f, 0.f, 0.f);
                    ++cur;
                }
            };
            auto fold = [&](int q, const float4& t) {
                const int pa = __builtin_amdgcn_readfirstlane(col_pa[q]);
                if (pa >= 7) return;                    // no sample of this ROI touches the column
                retire_until(pa);
                roi_fma4(b0, wx[cur][q], t); roi_fma4(b1, wx[cur + 1][q], t); roi_fma4(b2, wx[cur + 2 < 8 ? cur + 2 : 7][q], t);
            };
            // N columns starting at q0: column sums over the rows [ra, rb] (two rows = 2 N loads of 1 KiB in flight), then the fold
            auto block = [&](auto NC, int q0, int nvalid) {
                constexpr int N = decltype(NC)::value;
                float4 acc[N];
#pragma unroll
                for (int u = 0; u < N; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                // 32-bit byte offsets from the footprint origin (the 64 x 64 pixel tables bound them far below 4 GB): one address
                // register per load in flight, `global_load_dwordx4 v, v_offset, s[base]`
                unsigned coff[N];
#pragma unroll
                for (int u = 0; u < N; ++u) coff[u] = (unsigned)(q0 + (u < nvalid ? u : nvalid - 1)) * col_b32 + lane_off;     // tail block: repeats its last column
                int rr = ra;
                for (; rr + 1 <= rb; rr += 2) {
                    const float w0 = wyr[rr], w1 = wyr[rr + 1];
                    const unsigned r0 = (unsigned)rr * row_b32, r1 = r0 + row_b32;
                    float4 v0[N], v1[N];
#pragma unroll
                    for (int u = 0; u < N; ++u) {
                        v0[u] = roi_ld16(ubase + (r0 + coff[u]));
                        v1[u] = roi_ld16(ubase + (r1 + coff[u]));
                    }
#pragma unroll
                    for (int u = 0; u < N; ++u) {
                        acc[u].x += w0 * v0[u].x + w1 * v1[u].x; acc[u].y += w0 * v0[u].y + w1 * v1[u].y;
                        acc[u].z += w0 * v0[u].z + w1 * v1[u].z; acc[u].w += w0 * v0[u].w + w1 * v1[u].w;
                    }
                }
                if (rr == rb) {                                             // odd row count: the last row alone
                    const float w0 = wyr[rr];
                    const unsigned r0 = (unsigned)rr * row_b32;
                    float4 v0[N];
#pragma unroll
                    for (int u = 0; u < N; ++u) v0[u] = roi_ld16(ubase + (r0 + coff[u]));
#pragma unroll
                    for (int u = 0; u < N; ++u) roi_fma4(acc[u], w0, v0[u]);
                }
#pragma unroll
                for (int u = 0; u < N; ++u)
                    if (u < nvalid) fold(q0 + u, acc[u]);
            };
            int q0 = 0;
            for (; q0 + QB <= ncols; q0 += QB) block(std::integral_constant<int, QB>{}, q0, QB);
            if (ncols - q0 > QB / 2) block(std::integral_constant<int, QB>{}, q0, ncols - q0);
            else if (ncols - q0 > 0) block(std::integral_constant<int, QB / 2>{}, q0, ncols - q0);
            retire_until(7);
            if (cok) {
#pragma unroll 1
                for (int pw = 0; pw < 7; ++pw) {
                    const float4 v = done[pw * 64];
                    *reinterpret_cast<float4*>(orow + (size_t)pw * C + c) = make_float4(v.x / count, v.y / count, v.z / count, v.w / count);
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
        const char* mode0 = getenv("WD_ROI_KERNEL");
        if (n_rois >= 64 && n_rois <= 8192 && om && om[0] == '1' && mode0 && strcmp(mode0, "row") == 0) {
            order = flags + flags_cap;
            hipLaunchKernelGGL(roi_order_kernel, dim3(1), dim3(1024), (size_t)8192 * 8, (hipStream_t)stream, lv, n_levels, rois,
                               n_rois, min_level, canonical_level, canonical_size, order);
        }
        const char* mode = getenv("WD_ROI_KERNEL");             // experiments: "row" = one wave per (ROI, bin row), "sep" = round-1 kernel
        const bool aligned = (channels & 3) == 0 && ((uintptr_t)out & 15) == 0;
        const bool wg_path = aligned && !(mode && (strcmp(mode, "sep") == 0 || strcmp(mode, "row") == 0));
        const bool row_path = aligned && mode && strcmp(mode, "row") == 0;
        if (wg_path) {
            // round 4 default: one workgroup per ROI (roi_pool_wg_kernel); WD_ROI_ORDER=0 switches the bucket order off
            int* worder = nullptr;
            if (n_rois >= 64 && n_rois <= 8192 && !(om && om[0] == '0')) {
                worder = flags + flags_cap;
                hipLaunchKernelGGL(roi_bucket_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, lv, n_levels, rois, n_rois,
                                   min_level, canonical_level, canonical_size, worder);
            }
            hipLaunchKernelGGL(roi_pool_wg_kernel, dim3((unsigned)(worder ? (n_rois + 7) / 8 * 8 : n_rois)), dim3(64 * ROI_NW), 0,
                               (hipStream_t)stream, lv, n_levels, channels, batch, rois, n_rois, min_level, canonical_level,
                               canonical_size, out, (const int*)worder);
        } else if (row_path)
            hipLaunchKernelGGL(roi_pool_row_kernel, dim3((unsigned)(order ? (((n_rois * 7 + 3) / 4 + 7) / 8 * 8) : (n_rois * 7 + 3) / 4)),
                               dim3(256), 0, (hipStream_t)stream, lv, n_levels, channels, batch, rois, n_rois, min_level,
                               canonical_level, canonical_size, out, flags, (const int*)order);
        else
            hipLaunchKernelGGL(roi_pool_sep_kernel, dim3((unsigned)n_rois), dim3(256), 0, (hipStream_t)stream, lv, n_levels,
                               channels, batch, rois, n_rois, pooled, min_level, canonical_level, canonical_size, out, flags);
        if (!wg_path && !row_path)      // the one-workgroup-per-ROI kernel of round 1 still flags its large ROIs for the direct kernel
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
