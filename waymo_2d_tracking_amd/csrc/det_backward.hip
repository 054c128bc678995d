// Backward kernels of the detector custom ops (training fwd+bwd), gfx950.
//   * ROIAlign backward (detectron2 ROIAlign_backward restated): bilinear scatter of the bin gradients, lane = channel
//     so every atomic batch touches 64 consecutive floats of one feature pixel;
//   * deformable conv backward in the classic three-step form of the reference's CUDA op
//     (deformable_im2col / deformable_col2im / deformable_col2im_coord): the im2col slab is materialised in HBM
//     ([pixel][tap][channel], fully coalesced float4 stores), the two per-group GEMMs (dW, dcol) run on the library,
//     and ONE fused scatter kernel turns dcol into dX (float atomics) and dOffset (wave-reduced, one atomic per wave).
// Forward-speed kernels stay in det_deform.hip / det_roialign.hip; these favour simplicity and exactness.
#include "common.h"
#include <cstdlib>
#include <cstring>
#include "../../include/waymodet.h"

namespace {

constexpr int kMaxLevels = 8;
struct LevelsW {
    float* grad[kMaxLevels];
    int h[kMaxLevels];
    int w[kMaxLevels];
    float scale[kMaxLevels];
};

// bilinear footprint of one ROIAlign sample coordinate (detectron2 bilinear_interpolate_gradient along one axis): rows (lo, hi), weights
// (1 - l, l); false = the sample lies outside (-1, size) and contributes nothing
__device__ __forceinline__ bool roi_axis_sample(float y, int size, int& lo, int& hi, float& wlo, float& whi) {
    if (y < -1.0f || y > (float)size) return false;
    if (y <= 0) y = 0;
    lo = (int)y;
    if (lo >= size - 1) { hi = lo = size - 1; y = (float)lo; } else hi = lo + 1;
    whi = y - (float)lo;
    wlo = 1.f - whi;
    return true;
}

// The per-sample form (4 atomics per sample and channel): ROIs whose footprint exceeds the separable kernel's tables, and WD_ROI_BWD=sample.
__device__ void roi_bwd_samples(float* __restrict__ grad, int H, int W, int C, const float* __restrict__ g0, int P, float rsh, float rsw,
                                float bin_h, float bin_w, int gh, int gw, float count, int lane, int wave) {
    for (int cb = wave * 64; cb < C; cb += 256) {
        const int c = cb + lane;
        if (c >= C) continue;
        for (int ph = 0; ph < P; ++ph)
            for (int pw = 0; pw < P; ++pw) {
                const float g = g0[((size_t)ph * P + pw) * C + c] / count;
                for (int iy = 0; iy < gh; ++iy) {
                    const float y = rsh + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
                    int yl, yh; float hy, ly;
                    if (!roi_axis_sample(y, H, yl, yh, hy, ly)) continue;
                    for (int ix = 0; ix < gw; ++ix) {
                        const float x = rsw + (float)pw * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
                        int xl, xh; float hx, lx;
                        if (!roi_axis_sample(x, W, xl, xh, hx, lx)) continue;
                        atomicAdd(&grad[((size_t)yl * W + xl) * C + c], g * hy * hx);
                        atomicAdd(&grad[((size_t)yl * W + xh) * C + c], g * hy * lx);
                        atomicAdd(&grad[((size_t)yh * W + xl) * C + c], g * ly * hx);
                        atomicAdd(&grad[((size_t)yh * W + xh) * C + c], g * ly * lx);
                    }
                }
            }
    }
}

// ROIAlign backward, separable (round 4).  The gradient a ROI sends to feature pixel (y, x) is sum_{ph, pw} Wy[ph][y] Wx[pw][x] g[ph][pw] / count
// with Wy[ph][y] = the summed bilinear weights of bin row ph's samples on row y (the out-of-range test of a sample is per axis as well): one
// workgroup per ROI builds the two small tables in LDS, keeps the 7 x 7 bin gradients of its channels in registers, contracts the columns first
// (T[ph] = sum_pw Wx[pw][x] g[ph][pw]) and issues ONE float atomic per footprint pixel and channel - the per-sample kernel issued 4 per sample
// (16 - 64 samples per bin for the ROI sizes of a level): 975 us -> see DESIGN.md for a training batch of 512 ROIs.
constexpr int RB_MAXP = 7, RB_EXT = 192;
template <int PP>
__global__ __launch_bounds__(256) void roi_pool_fpn_bwd_kernel(LevelsW lv, int n_levels, int C, int batch, const float* __restrict__ rois, int P,
                                                               int min_level, int canonical_level, float canonical_size,
                                                               const float* __restrict__ gout, int split) {
    __shared__ float wy[RB_MAXP][RB_EXT], wx[RB_MAXP][RB_EXT];
    // `split` workgroups per ROI, each takes every split-th footprint column: the footprints of a training batch differ 50 x in area and the
    // launch ends with its largest ROI
    const int r = blockIdx.x / split, part = blockIdx.x - r * split;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* roi = rois + 5 * (size_t)r;
    const int b = (int)roi[0];
    if (b < 0 || b >= batch) return;
    const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
    const float size = sqrtf((x2 - x1) * (y2 - y1));
    int lvl = (int)floorf((float)canonical_level + log2f(size / canonical_size + 1e-8f));
    lvl = lvl < min_level ? min_level : (lvl > min_level + n_levels - 1 ? min_level + n_levels - 1 : lvl);
    const int li = lvl - min_level;
    const int H = lv.h[li], W = lv.w[li];
    const float scale = lv.scale[li];
    float* __restrict__ grad = lv.grad[li] + (size_t)b * H * W * C;
    const float rsw = x1 * scale - 0.5f, rsh = y1 * scale - 0.5f;
    const float roi_w = (x2 * scale - 0.5f) - rsw, roi_h = (y2 * scale - 0.5f) - rsh;
    const float bin_h = roi_h / (float)P, bin_w = roi_w / (float)P;
    const int gh = (int)ceilf(roi_h / (float)P), gw = (int)ceilf(roi_w / (float)P);
    const float count = (float)((gh * gw) > 1 ? gh * gw : 1);
    const float* g0 = gout + (size_t)r * P * P * C;
    // footprint: rows r0 .. r0 + nr - 1, columns c0 .. c0 + nc - 1 (every sample's two rows / columns lie inside)
    const int r0 = max(0, min(H - 1, (int)floorf(rsh))), c0 = max(0, min(W - 1, (int)floorf(rsw)));
    const int r1 = max(r0, min(H - 1, (int)floorf(rsh + roi_h) + 1)), c1 = max(c0, min(W - 1, (int)floorf(rsw + roi_w) + 1));
    const int nr = r1 - r0 + 1, nc = c1 - c0 + 1;
    if (PP == 0 || P != PP || nr > RB_EXT || nc > RB_EXT) {
        if (part == 0) roi_bwd_samples(grad, H, W, C, g0, P, rsh, rsw, bin_h, bin_w, gh, gw, count, lane, wave);
        return;
    }
    for (int i = threadIdx.x; i < RB_MAXP * RB_EXT; i += 256) { (&wy[0][0])[i] = 0.f; (&wx[0][0])[i] = 0.f; }
    __syncthreads();
    if (threadIdx.x < PP) {                        // one thread per bin row: its gh samples in turn (plain LDS adds, no two threads share a row)
        const int ph = threadIdx.x;
        for (int iy = 0; iy < gh; ++iy) {
            const float y = rsh + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
            int lo, hi; float wl, wh;
            if (!roi_axis_sample(y, H, lo, hi, wl, wh)) continue;
            // the footprint bound and the sample coordinate are two float expressions: a sample that rounds across the last row can only carry
            // weight 0 there (its y is within an ulp of the integer), but the index is still kept inside the table
            if ((unsigned)(lo - r0) < (unsigned)nr) wy[ph][lo - r0] += wl;
            if ((unsigned)(hi - r0) < (unsigned)nr) wy[ph][hi - r0] += wh;
        }
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + PP) {
        const int pw = threadIdx.x - 64;
        for (int ix = 0; ix < gw; ++ix) {
            const float x = rsw + (float)pw * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
            int lo, hi; float wl, wh;
            if (!roi_axis_sample(x, W, lo, hi, wl, wh)) continue;
            if ((unsigned)(lo - c0) < (unsigned)nc) wx[pw][lo - c0] += wl;
            if ((unsigned)(hi - c0) < (unsigned)nc) wx[pw][hi - c0] += wh;
        }
    }
    __syncthreads();
    const float inv_count = 1.f / count;
    for (int cb = wave * 64; cb < C; cb += 256) {
        const int c = cb + lane;
        if (c >= C) continue;
        float g[PP > 0 ? PP : 1][PP > 0 ? PP : 1];
#pragma unroll
        for (int ph = 0; ph < PP; ++ph)
#pragma unroll
            for (int pw = 0; pw < PP; ++pw) g[ph][pw] = g0[((size_t)ph * PP + pw) * C + c] * inv_count;
        for (int x = part; x < nc; x += split) {
            float t[PP > 0 ? PP : 1];
#pragma unroll
            for (int ph = 0; ph < PP; ++ph) t[ph] = 0.f;
#pragma unroll
            for (int pw = 0; pw < PP; ++pw) {
                const float w = wx[pw][x];
#pragma unroll
                for (int ph = 0; ph < PP; ++ph) t[ph] += w * g[ph][pw];
            }
            float* __restrict__ gp = grad + ((size_t)r0 * W + c0 + x) * C + c;
            for (int y = 0; y < nr; ++y) {
                float v = 0.f;
#pragma unroll
                for (int ph = 0; ph < PP; ++ph) v += wy[ph][y] * t[ph];
                if (v != 0.f) atomicAdd(gp + (size_t)y * W * C, v);
            }
        }
    }
}

struct Tap {
    int idx[4];
    float wgt[4];
    float lh, lw;
    int ok;          // sample inside (-1,H) x (-1,W)
};

__device__ __forceinline__ Tap make_tap(const float* __restrict__ offset, long gp, int k, int Ho, int Wo, int H, int W,
                                        int stride, int pad) {
    Tap t;
#pragma unroll
    for (int q = 0; q < 4; ++q) { t.idx[q] = 0; t.wgt[q] = 0.f; }
    t.lh = 0.f; t.lw = 0.f; t.ok = 0;
    const int n = (int)(gp / ((long)Ho * Wo));
    const int rem = (int)(gp - (long)n * Ho * Wo);
    const int ho = rem / Wo, wo = rem - ho * Wo;
    const int kh = k / 3, kw = k - 3 * kh;
    const float h_im = (float)(ho * stride - pad + kh) + offset[(size_t)gp * 18 + 2 * k];
    const float w_im = (float)(wo * stride - pad + kw) + offset[(size_t)gp * 18 + 2 * k + 1];
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
        const int hh = hl + 1, wh = wl + 1;
        const float lh = h_im - (float)hl, lw = w_im - (float)wl;
        const int base = n * H * W;
        t.lh = lh; t.lw = lw; t.ok = 1;
        if (hl >= 0 && wl >= 0) { t.idx[0] = base + hl * W + wl; t.wgt[0] = (1.f - lh) * (1.f - lw); }
        if (hl >= 0 && wh <= W - 1) { t.idx[1] = base + hl * W + wh; t.wgt[1] = (1.f - lh) * lw; }
        if (hh <= H - 1 && wl >= 0) { t.idx[2] = base + hh * W + wl; t.wgt[2] = lh * (1.f - lw); }
        if (hh <= H - 1 && wh <= W - 1) { t.idx[3] = base + hh * W + wh; t.wgt[3] = lh * lw; }
        // validity flags for the coordinate gradient are encoded as idx >= 0 with wgt possibly 0: keep a bit set
        t.ok |= (hl >= 0 && wl >= 0) ? 2 : 0;
        t.ok |= (hl >= 0 && wh <= W - 1) ? 4 : 0;
        t.ok |= (hh <= H - 1 && wl >= 0) ? 8 : 0;
        t.ok |= (hh <= H - 1 && wh <= W - 1) ? 16 : 0;
    }
    return t;
}

// col[p][k][c] = bilinear(x, p, k, c)   (c fastest: float4 per thread, C/4 consecutive threads per (p, k))
// column layout (both directions): [group][pixel p][tap k][ci], group-major, so that the per-group GEMMs between im2col and
// col2im are strided-batched library GEMMs on these buffers without any permute copy (groups = 1: the plain [p][k][c])
__device__ __forceinline__ size_t col_index(long gp, int k, int c, long npix, int cg) {
    const int g = c / cg, ci = c - g * cg;
    return (((size_t)g * npix + gp) * 9 + k) * cg + ci;
}

__global__ __launch_bounds__(256) void deform_im2col_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                            long npix, int Ho, int Wo, int H, int W, int C, int cg, int stride,
                                                            int pad, float* __restrict__ col) {
    const int c4n = C >> 2;
    const long total = npix * 9 * c4n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c4 = (int)(e % c4n);
        const long pk = e / c4n;
        const int k = (int)(pk % 9);
        const long gp = pk / 9;
        const Tap t = make_tap(offset, gp, k, Ho, Wo, H, W, stride, pad);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 s = *reinterpret_cast<const float4*>(x + (size_t)t.idx[q] * C + c4 * 4);
            v.x += t.wgt[q] * s.x; v.y += t.wgt[q] * s.y; v.z += t.wgt[q] * s.z; v.w += t.wgt[q] * s.w;
        }
        *reinterpret_cast<float4*>(col + col_index(gp, k, c4 * 4, npix, cg)) = v;
    }
}

// dcol[p][k][c] -> dx (atomics) and doffset[p][2k], [2k+1] (reduced over channels: wave shuffle + one atomic per wave)
template <bool WITH_DX>
__global__ __launch_bounds__(256) void deform_col2im_kernel(const float* __restrict__ dcol, const float* __restrict__ x,
                                                            const float* __restrict__ offset, long npix, int Ho, int Wo,
                                                            int H, int W, int C, int cg, int stride, int pad,
                                                            float* __restrict__ dx, float* __restrict__ doffset) {
    const int c4n = C >> 2;                       // a multiple of 32 (C % 128 == 0): a wave never straddles two (p, k)
    const long total = npix * 9 * c4n;
    for (long e0 = (long)blockIdx.x * 256; e0 < total; e0 += (long)gridDim.x * 256) {
        const long e = e0 + threadIdx.x;
        const bool act = e < total;
        const long ee = act ? e : total - 1;
        const int c4 = (int)(ee % c4n);
        const long pk = ee / c4n;
        const int k = (int)(pk % 9);
        const long gp = pk / 9;
        const Tap t = make_tap(offset, gp, k, Ho, Wo, H, W, stride, pad);
        float4 g = act ? *reinterpret_cast<const float4*>(dcol + col_index(gp, k, c4 * 4, npix, cg)) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = *reinterpret_cast<const float4*>(x + (size_t)t.idx[q] * C + c4 * 4);
            if (!((t.ok >> (q + 1)) & 1)) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (WITH_DX && act && t.wgt[q] != 0.f) {
                float* d = dx + (size_t)t.idx[q] * C + c4 * 4;
                atomicAdd(d + 0, t.wgt[q] * g.x); atomicAdd(d + 1, t.wgt[q] * g.y);
                atomicAdd(d + 2, t.wgt[q] * g.z); atomicAdd(d + 3, t.wgt[q] * g.w);
            }
        }
        // d val / d h = (v3 - v1)(1 - lw) + (v4 - v2) lw ;  d val / d w = (v2 - v1)(1 - lh) + (v4 - v3) lh
        float dh = 0.f, dw = 0.f;
        if (t.ok & 1) {
            const float a = 1.f - t.lw, b = t.lw, cc = 1.f - t.lh, d2 = t.lh;
            dh = g.x * ((v[2].x - v[0].x) * a + (v[3].x - v[1].x) * b) + g.y * ((v[2].y - v[0].y) * a + (v[3].y - v[1].y) * b) +
                 g.z * ((v[2].z - v[0].z) * a + (v[3].z - v[1].z) * b) + g.w * ((v[2].w - v[0].w) * a + (v[3].w - v[1].w) * b);
            dw = g.x * ((v[1].x - v[0].x) * cc + (v[3].x - v[2].x) * d2) + g.y * ((v[1].y - v[0].y) * cc + (v[3].y - v[2].y) * d2) +
                 g.z * ((v[1].z - v[0].z) * cc + (v[3].z - v[2].z) * d2) + g.w * ((v[1].w - v[0].w) * cc + (v[3].w - v[2].w) * d2);
        }
        // reduce over the 32 lanes that share (p, k): c4n is a multiple of 32
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { dh += __shfl_xor(dh, o, 64); dw += __shfl_xor(dw, o, 64); }
        if (act && (threadIdx.x & 31) == 0) {
            atomicAdd(&doffset[(size_t)gp * 18 + 2 * k], dh);
            atomicAdd(&doffset[(size_t)gp * 18 + 2 * k + 1], dw);
        }
    }
}

// dx of stride-1 layers as a GATHER: a workgroup owns an 8 x 8 output tile x 64 channels and inverts its sampling table
// in LDS - for every pixel of the 14 x 14 input patch the list of (dcol row, bilinear weight) pairs that touch it
// (count with integer LDS atomics, exclusive scan, fill) - then each wave sums the rows of one patch pixel in
// registers (lane = channel, 16 independent 256-byte loads in flight) and issues ONE global atomic per (patch pixel,
// channel): 196 x 64 per workgroup instead of 2304 x 64.  (LDS float atomics were measured at ~250 cycles per wave
// instruction on gfx950 - an LDS accumulator version of this kernel ran 1.1 ms against 2.2 ms for plain global atomics.)
// Samples whose corners leave the patch go to global memory directly.  doffset comes from deform_col2im_kernel<false>.
constexpr int BT = 8, BPR = 2, BCH = 64, BNE = BT * BT * 9;
template <int S>                             // S = stride (1 / 2): the input patch of a tile is ((BT-1)*S + 3 + 2*BPR)^2 pixels
__global__ __launch_bounds__(256) void deform_col2im_dx_gather_kernel(const float* __restrict__ dcol, const float* __restrict__ offset,
                                                                      int batch, int Ho, int Wo, int H, int W, int C, int cg,
                                                                      float* __restrict__ dx) {
    constexpr int BPS = (BT - 1) * S + 3 + 2 * BPR;      // 14 / 21
    constexpr int NPP = BPS * BPS;
    constexpr int PER = (NPP + 63) / 64;                 // counts scanned per lane
    __shared__ float4 tw[BNE];
    __shared__ int toff[BNE];                  // patch pixel of corner (hl, wl); < 0: packed image coordinates (global path)
    __shared__ int trow[BNE];                  // dcol row (gp * 9 + k) of the entry, -1 = output pixel outside the image
    __shared__ int cnt[NPP], start[NPP + 1], cursor[NPP];
    __shared__ int ent_row[BNE * 4];
    __shared__ float ent_w[BNE * 4];
    __shared__ int n_global;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (Wo + BT - 1) / BT, tiles_y = (Ho + BT - 1) / BT;
    const int ntiles = batch * tiles_y * tiles_x;
    const int nchunks = C / BCH;
    const int chunk = blockIdx.x % nchunks, tile = blockIdx.x / nchunks;
    if (tile >= ntiles) return;
    const int tn = tile / (tiles_y * tiles_x);
    const int trem = tile - tn * tiles_y * tiles_x;
    const int tyy = trem / tiles_x, txx = trem - tyy * tiles_x;
    const int c0 = chunk * BCH;
    const int py0 = tyy * BT * S - 1 - BPR, px0 = txx * BT * S - 1 - BPR;  // image coordinates of patch pixel (0, 0)
    for (int i = tid; i < NPP; i += 256) cnt[i] = 0;
    if (tid == 0) n_global = 0;
    __syncthreads();
    const int dq[4] = {0, 1, BPS, BPS + 1};
    for (int e = tid; e < BNE; e += 256) {
        const int p = e / 9, k = e - 9 * p;
        const int ho = tyy * BT + (p >> 3), wo = txx * BT + (p & 7);
        const int gp = (ho < Ho && wo < Wo) ? (tn * Ho + ho) * Wo + wo : -1;
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        int off = 0;
        if (gp >= 0) {
            const int kh = k / 3, kw = k - 3 * kh;
            const float ry = (float)((p >> 3) * S + kh + BPR) + offset[(size_t)gp * 18 + 2 * k];      // patch coordinates
            const float rx = (float)((p & 7) * S + kw + BPR) + offset[(size_t)gp * 18 + 2 * k + 1];
            const float h_im = ry + (float)py0, w_im = rx + (float)px0;
            if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const float fy = floorf(ry), fx = floorf(rx);
                const int hl = (int)fy, wl = (int)fx;
                const float lh = ry - fy, lw = rx - fx, uh = 1.f - lh, uw = 1.f - lw;
                w4 = make_float4(uh * uw, uh * lw, lh * uw, lh * lw);
                if (hl >= 0 && hl < BPS - 1 && wl >= 0 && wl < BPS - 1) {
                    off = hl * BPS + wl;
                    const float wq[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (wq[q] != 0.f) atomicAdd(&cnt[off + dq[q]], 1);
                } else {
                    off = -1 - ((hl + py0 + 2) * (W + 4) + (wl + px0 + 2));        // image coordinates for the global path
                    n_global = 1;
                }
            }
        }
        tw[e] = w4;
        toff[e] = off;
        trow[e] = gp >= 0 ? gp * 9 + k : -1;
    }
    __syncthreads();
    if (wave == 0) {                           // exclusive scan of the patch-pixel counts (PER per lane + wave scan)
        int c[PER], s4 = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = lane * PER + j;
            c[j] = idx < NPP ? cnt[idx] : 0;
            s4 += c[j];
        }
        int inc = s4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        int base = inc - s4;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = lane * PER + j;
            if (idx < NPP) { start[idx] = base; cursor[idx] = base; }
            base += c[j];
        }
        if (lane == 63) start[NPP] = inc;
    }
    __syncthreads();
    for (int e = tid; e < BNE; e += 256) {
        const int off = toff[e];
        const int row = trow[e];
        if (row < 0 || off < 0) continue;
        const float4 w4 = tw[e];
        const float wq[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (wq[q] != 0.f) {
                const int slot = atomicAdd(&cursor[off + dq[q]], 1);
                ent_row[slot] = row;
                ent_w[slot] = wq[q];
            }
    }
    __syncthreads();
    const float* __restrict__ dc = dcol + col_index(0, 0, c0 + lane, (long)batch * Ho * Wo, cg);   // + row * cg per entry
    for (int pp = wave; pp < NPP; pp += 4) {
        const int n0 = start[pp], n1 = start[pp + 1];
        const int r = pp / BPS, cc = pp - r * BPS;
        const int iy = py0 + r, ix = px0 + cc;
        if (n1 == n0 || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;     // weights on pixels outside the image: no gradient
        float acc = 0.f;
        for (int i = n0; i < n1; i += 16) {
            float g[16], wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = (i + u < n1) ? i + u : n1 - 1;
                wv[u] = (i + u < n1) ? ent_w[j] : 0.f;
                g[u] = dc[(size_t)ent_row[j] * cg];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += wv[u] * g[u];
        }
#ifdef WD_DXG_NOATOMIC        // experiments: what do the global atomics cost? (results are wrong)
        dx[((size_t)(tn * H + iy) * W + ix) * C + c0 + lane] = acc;
#else
        atomicAdd(&dx[((size_t)(tn * H + iy) * W + ix) * C + c0 + lane], acc);
#endif
    }
    if (n_global) {                            // large offsets: per-corner global atomics for the flagged samples only
        for (int e = wave; e < BNE; e += 4) {
            const int off = toff[e];
            const int row = trow[e];
            if (off >= 0 || row < 0) continue;
            const float g = dc[(size_t)row * cg];
            const float4 w4 = tw[e];
            const float wq[4] = {w4.x, w4.y, w4.z, w4.w};
            const int code = -1 - off;
            const int ih = code / (W + 4) - 2, iw = code - (ih + 2) * (W + 4) - 2;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int yy = ih + (qd >> 1), xx = iw + (qd & 1);
                if (yy >= 0 && yy < H && xx >= 0 && xx < W && wq[qd] != 0.f)
                    atomicAdd(&dx[((size_t)(tn * H + yy) * W + xx) * C + c0 + lane], wq[qd] * g);
            }
        }
    }
}

}  // namespace

extern "C" {

int wd_roi_pool_fpn_bwd_f32(float* const* grad_feats, const int32_t* heights, const int32_t* widths, const float* scales,
                            int n_levels, int channels, int batch, const float* rois, int n_rois, int pooled, int min_level,
                            int canonical_level, float canonical_size, const float* grad_out, void* stream) {
    WT_TRY(wt::ensure_device());
    if (n_levels < 1 || n_levels > kMaxLevels || channels < 1 || pooled < 1) { wt::set_error("wd_roi_pool_fpn_bwd_f32: bad shape"); return WT_ERR_INVALID; }
    if (n_rois <= 0) return WT_OK;
    LevelsW lv;
    for (int i = 0; i < n_levels; ++i) { lv.grad[i] = grad_feats[i]; lv.h[i] = heights[i]; lv.w[i] = widths[i]; lv.scale[i] = scales[i]; }
    const char* mode = getenv("WD_ROI_BWD");                 // experiments: "sample" = 4 atomics per sample (the round-1 kernel)
    int split = 4;
    if (const char* e = getenv("WD_ROI_BWD_SPLIT")) split = atoi(e);
    if (split < 1) split = 1;
    if (pooled == 7 && !(mode && strcmp(mode, "sample") == 0))
        hipLaunchKernelGGL(roi_pool_fpn_bwd_kernel<7>, dim3((unsigned)(n_rois * split)), dim3(256), 0, (hipStream_t)stream, lv, n_levels, channels,
                           batch, rois, pooled, min_level, canonical_level, canonical_size, grad_out, split);
    else
        hipLaunchKernelGGL(roi_pool_fpn_bwd_kernel<0>, dim3((unsigned)n_rois), dim3(256), 0, (hipStream_t)stream, lv, n_levels, channels,
                           batch, rois, pooled, min_level, canonical_level, canonical_size, grad_out, 1);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

static int check_deform(int c, int groups, int h, int w, int stride) {
    if (c % 128 || h < 1 || w < 1 || stride < 1 || groups < 1 || c % groups || (c / groups) % 4) {
        wt::set_error("deform backward: need C %% 128 == 0 and (C / groups) %% 4 == 0 (C=%d groups=%d)", c, groups);
        return WT_ERR_INVALID;
    }
    return WT_OK;
}

int wd_deform_im2col_f32(const float* x, const float* offset, int batch, int h, int w, int c, int groups, int stride, int pad,
                         float* col, void* stream) {
    WT_TRY(wt::ensure_device());
    WT_TRY(check_deform(c, groups, h, w, stride));
    const int ho = (h + 2 * pad - 3) / stride + 1, wo = (w + 2 * pad - 3) / stride + 1;
    const long npix = (long)batch * ho * wo;
    const long total = npix * 9 * (c / 4);
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(deform_im2col_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream,
                       x, offset, npix, ho, wo, h, w, c, c / groups, stride, pad, col);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

int wd_deform_col2im_f32(const float* dcol, const float* x, const float* offset, int batch, int h, int w, int c, int groups,
                         int stride, int pad, float* dx, float* doffset, void* stream) {
    WT_TRY(wt::ensure_device());
    WT_TRY(check_deform(c, groups, h, w, stride));
    const int ho = (h + 2 * pad - 3) / stride + 1, wo = (w + 2 * pad - 3) / stride + 1;
    const long npix = (long)batch * ho * wo;
    const long total = npix * 9 * (c / 4);
    const long blocks = (total + 255) / 256;
    const char* mode = getenv("WD_COL2IM");                 // experiments: "atomic" = one global atomic per corner value
    if ((stride == 1 || stride == 2) && pad == 1 && !(mode && strcmp(mode, "atomic") == 0)) {
        // doffset (gather + channel reduction) and dx (inverted sampling table, gather) as two kernels
        hipLaunchKernelGGL(deform_col2im_kernel<false>, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                           (hipStream_t)stream, dcol, x, offset, npix, ho, wo, h, w, c, c / groups, stride, pad, dx, doffset);
        const long nwg = (long)batch * ((ho + BT - 1) / BT) * ((wo + BT - 1) / BT) * (c / BCH);
        if (stride == 1)
            hipLaunchKernelGGL(deform_col2im_dx_gather_kernel<1>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, dcol,
                               offset, batch, ho, wo, h, w, c, c / groups, dx);
        else
            hipLaunchKernelGGL(deform_col2im_dx_gather_kernel<2>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, dcol,
                               offset, batch, ho, wo, h, w, c, c / groups, dx);
    } else {
        hipLaunchKernelGGL(deform_col2im_kernel<true>, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                           (hipStream_t)stream, dcol, x, offset, npix, ho, wo, h, w, c, c / groups, stride, pad, dx, doffset);
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // extern "C"
