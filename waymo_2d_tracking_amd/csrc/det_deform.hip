// Deformable 3x3 convolution forward (detectron2 DeformConv / ModulatedDeformConv; logs/12442/job.log:412-415,
// SURVEY.md App. C) as an implicit GEMM on the gfx950 f32 matrix cores, NHWC, groups = 32, deformable_groups = 1.
//
// Workgroup = 64 output pixels x 128 channels (= 128/Cg whole groups), 256 threads:
//   1. sampling table in LDS: for each (pixel, tap) the 4 bilinear corner pixels and weights (x mask); the table is
//      shared by every channel because deformable_groups = 1;
//   2. per tap: the 64 x 128 im2col slab is gathered with 512-byte coalesced runs (32 lanes x float4 = 128 channels
//      of one corner pixel), blended, and staged in LDS (row stride 130 floats: conflict-free ds_read_b32 for the
//      MFMA A fragments, 8-byte aligned rows for the ds_write_b64 staging);
//   3. each wave multiplies the slab with its 32 output channels' weights: v_mfma_f32_16x16x4_f32, A from LDS,
//      B (packed [group][tap][ci][co]) straight from L2, 4 x 2 accumulator tiles per wave;
//   4. epilogue fuses the FrozenBatchNorm affine and ReLU and writes NHWC.
// Per layer: 2*C_out*(C_in/groups)*9*H_out*W_out flops (5.66 GFLOP for every res3/res4/res5 layer at 1920x1280).
#include "common.h"
#include "../../include/waymodet.h"
#include <cstdlib>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int TP = 64;          // pixels per workgroup
constexpr int CCH = 128;        // channels per workgroup
constexpr int LDC = CCH + 2;    // LDS row stride of the im2col slab

struct Sample {
    unsigned idx[4];  // BYTE offset of the corner pixel's channel vector (pixel index * C * 4); corners outside the image
    float wgt[4];   // point at pixel 0 with weight 0, so the gather issues all its loads unconditionally (no branches)
};                  // 32-bit offsets keep the gather's address math to one v_add per load (saddr + voffset form)

template <int CG, bool DEFORM>
__global__ __launch_bounds__(256, (CG <= 32 ? 3 : 2)) void deform_conv3x3_kernel(
    const float* __restrict__ x, const float* __restrict__ offset, const float* __restrict__ mask,
    const float* __restrict__ wp, const float* __restrict__ scale, const float* __restrict__ bias, int relu,
    int batch, int H, int W, int C, int Cout, int Ho, int Wo, int stride, int pad, float* __restrict__ y) {
    // CG <= 32: every wave gathers, stages and multiplies ITS OWN 32 channels (wave-private 64 x 32 slab, row stride
    // 34): producer and consumer of a slab are the same wave, so the tap loop needs no workgroup barrier at all and
    // the 12 waves of a CU drift apart - one wave's MFMAs run under another wave's gather latency.
    // CG == 64: a group spans two waves; the shared 64 x 128 slab with two barriers per tap is kept.
    constexpr bool PRIV = CG <= 32;
    constexpr int LDW = PRIV ? 34 : LDC;
    __shared__ Sample tab[TP * 9];
    __shared__ __attribute__((aligned(16))) float col[PRIV ? 4 * TP * 34 : TP * LDC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware work mapping (speed only): workgroup b is dispatched to XCD b % 8, whose private 4 MiB L2 should see
    // a compact slice of the input.  The gather re-reads every input pixel ~9x per channel chunk, so each XCD gets
    // whole channel chunks (>= 8 chunks) or a contiguous range of pixel tiles of one chunk (< 8 chunks); its
    // working set is then (pixels in flight) x 128 channels instead of the whole feature map.
    // 8 x 8 output-pixel tiles: the undeformed 3x3 footprint of a tile is 10 x 10 input pixels (1.56x its outputs)
    // instead of 3 x 66 (3.1x) for a 64 x 1 strip - less L1 / L2 traffic per tile and more reuse between taps
    const int tiles_x = (Wo + 7) >> 3, tiles_y = (Ho + 7) >> 3;
    const int ntiles = batch * tiles_y * tiles_x;
    const int nchunks = C / CCH;
    int tile, chunk;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, slot = b >> 3;
        if (nchunks >= 8 && (nchunks & 7) == 0) {
            const int cpx = nchunks >> 3;                    // chunks per XCD
            chunk = xcd * cpx + slot % cpx;
            tile = slot / cpx;
        } else if (nchunks < 8 && (8 % nchunks) == 0 && (ntiles % (8 / nchunks)) == 0) {
            const int xpc = 8 / nchunks;                     // XCDs per chunk
            const int tpp = ntiles / xpc;                    // tiles per XCD
            chunk = xcd / xpc;
            tile = (xcd % xpc) * tpp + slot;
        } else {
            chunk = b / ntiles;
            tile = b - chunk * ntiles;
        }
    }
    if (tile >= ntiles || chunk >= nchunks) return;
    const int tn = tile / (tiles_y * tiles_x);
    const int trem = tile - tn * tiles_y * tiles_x;
    const int tyy = trem / tiles_x, txx = trem - tyy * tiles_x;
    __shared__ int pix[TP];                      // linear output pixel index of the tile's 64 pixels, -1 = outside
    if (tid < TP) {
        const int ho = tyy * 8 + (tid >> 3), wo = txx * 8 + (tid & 7);
        pix[tid] = (ho < Ho && wo < Wo) ? (tn * Ho + ho) * Wo + wo : -1;
    }
    __syncthreads();
    const int c0 = chunk * CCH;                  // first input (= output) channel of this chunk
    // ---- 1. sampling table ----
    for (int e = tid; e < TP * 9; e += 256) {
        const int p = e / 9, k = e - 9 * p;
        const long gp = pix[p];
        Sample s;
#pragma unroll
        for (int q = 0; q < 4; ++q) { s.idx[q] = 0u; s.wgt[q] = 0.f; }
        if (gp >= 0) {
            const int n = (int)(gp / ((long)Ho * Wo));
            const int rem = (int)(gp - (long)n * Ho * Wo);
            const int ho = rem / Wo, wo = rem - ho * Wo;
            const int kh = k / 3, kw = k - 3 * kh;
            if (!DEFORM) {                    // plain grouped 3x3 convolution: one integer tap, zero padding
                const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
                if (hi >= 0 && hi < H && wi >= 0 && wi < W) { s.idx[0] = (unsigned)(n * H * W + hi * W + wi) * (unsigned)C * 4u; s.wgt[0] = 1.f; }
                tab[e] = s;
                continue;
            }
            const float* off = offset + (size_t)gp * 18;
            const float h_im = (float)(ho * stride - pad + kh) + off[2 * k];
            const float w_im = (float)(wo * stride - pad + kw) + off[2 * k + 1];
            if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
                const int hh = hl + 1, wh = wl + 1;
                const float lh = h_im - (float)hl, lw = w_im - (float)wl;
                const float uh = 1.f - lh, uw = 1.f - lw;
                const float m = mask ? mask[(size_t)gp * 9 + k] : 1.f;
                const int base = n * H * W;
                if (hl >= 0 && wl >= 0) { s.idx[0] = (unsigned)(base + hl * W + wl) * (unsigned)C * 4u; s.wgt[0] = uh * uw * m; }
                if (hl >= 0 && wh <= W - 1) { s.idx[1] = (unsigned)(base + hl * W + wh) * (unsigned)C * 4u; s.wgt[1] = uh * lw * m; }
                if (hh <= H - 1 && wl >= 0) { s.idx[2] = (unsigned)(base + hh * W + wl) * (unsigned)C * 4u; s.wgt[2] = lh * uw * m; }
                if (hh <= H - 1 && wh <= W - 1) { s.idx[3] = (unsigned)(base + hh * W + wh) * (unsigned)C * 4u; s.wgt[3] = lh * lw * m; }
            }
        }
        tab[e] = s;
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int gq = PRIV ? (lane & 7) : (tid & 31);     // float4 column of the slab this thread gathers
    const int gp0 = PRIV ? (lane >> 3) : (tid >> 5);   // first pixel row (rows gp0, gp0 + 8, ...)
    const int co_w = wave * 32;        // this wave's 32 output channels inside the chunk
    float* colw = PRIV ? col + wave * (TP * 34) : col;
    auto slab_sync = [&]() {
        if (PRIV) {                    // same-wave LDS traffic is executed in order: only the compiler must not reorder
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    };
    // Software pipeline: the first half of tap k+1's gather (16 x 16-byte loads per lane) is issued right after the
    // slab barrier and stays in flight under the MFMAs of tap k (PMC on the unpipelined loop: 53 % of wave cycles in
    // s_waitcnt / barriers, 14 % issuing); only the second half's latency is exposed.
    constexpr int NQ = DEFORM ? 4 : 1;
    const char* xb = reinterpret_cast<const char*>(x);
    const unsigned lane_off = (unsigned)(c0 + (PRIV ? co_w : 0) + gq * 4) * 4u;
    float4 tA[4][NQ];
    // the bilinear weights are re-read from the LDS table at blend time (broadcast reads) instead of living in 16
    // VGPRs across the MFMA phase: that register room holds the prefetched B operands below at 3 waves / SIMD
    auto issue = [&](int k, int half, float4 (&t)[4][NQ]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = gp0 + 8 * (half * 4 + i);
            const Sample& s = tab[p * 9 + k];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                t[i][q] = *reinterpret_cast<const float4*>(xb + (size_t)(s.idx[q] + lane_off));
        }
    };
    auto blend = [&](int k, int half, const float4 (&t)[4][NQ]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = gp0 + 8 * (half * 4 + i);
            const Sample& s = tab[p * 9 + k];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float wq = s.wgt[q];
                v.x += wq * t[i][q].x; v.y += wq * t[i][q].y;
                v.z += wq * t[i][q].z; v.w += wq * t[i][q].w;
            }
            float2* d = reinterpret_cast<float2*>(&colw[p * LDW + gq * 4]);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        }
    };
    // B operands (this wave's weights of tap k) are fetched BEFORE the second half of the gather: vector loads return
    // in order, so a B load queued behind the next tap's prefetch would make the MFMAs wait for that prefetch too.
    constexpr int KS = (CG >= 16) ? CG / 4 : 4;          // k-steps per N tile
    float bR[2][KS];
    auto load_b = [&](int k) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int co_l = co_w + 16 * nt;
            if (CG >= 16) {
                const int g = (c0 + co_l) / CG;
                const int co_g = (co_l % CG) + (lane & 15);
                const float* wb = wp + ((size_t)(g * 9 + k) * CG) * CG + co_g;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) bR[nt][kk] = wb[(size_t)(kk * 4 + (lane >> 4)) * CG];
            } else {
                // CG == 8: the 16-wide N tile spans two groups -> block-diagonal B over the 16 input channels
                const int j = lane & 15;
                const int g = (c0 + co_l) / 8 + (j >> 3);
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    const int ci = kk * 4 + (lane >> 4);         // 0..15 inside the tile's channel range
                    const float wv = wp[((size_t)(g * 9 + k) * 8 + (ci & 7)) * 8 + (j & 7)];
                    bR[nt][kk] = ((ci >> 3) == (j >> 3)) ? wv : 0.f;
                }
            }
        }
    };
    issue(0, 0, tA);
    for (int k = 0; k < 9; ++k) {
        // ---- 2. gather + blend the tap's im2col slab ----
        blend(k, 0, tA);
        load_b(k);
        {
            float4 tB[4][NQ];
            issue(k, 1, tB);
            __builtin_amdgcn_sched_barrier(0);           // keep all 16 loads in flight (the scheduler would serialise
            blend(k, 1, tB);                             // them 4 at a time to save registers)
        }
        slab_sync();
        if (k + 1 < 9) issue(k + 1, 0, tA);              // in flight during the MFMAs below
        __builtin_amdgcn_sched_barrier(0);
        // ---- 3. MFMA: out[64 px][32 co of this wave] += slab[64 px][ci of the group] * W[ci][co] ----
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int co_l = co_w + 16 * nt;                 // first co of this N tile inside the chunk
            // first slab column of the tile's input channels (wave-private slabs hold only the wave's 32 channels)
            const int cl = PRIV ? 16 * nt : co_l;
            const int a_col = (CG >= 16) ? (cl / CG) * CG : cl;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int ci = kk * 4 + (lane >> 4);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const float a = colw[(mt * 16 + (lane & 15)) * LDW + a_col + ci];
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bR[nt][kk], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        slab_sync();
    }
    // ---- 4. epilogue ----
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int co = c0 + co_w + 16 * nt + (lane & 15);
        const float sc = scale ? scale[co] : 1.f;
        const float bi = bias ? bias[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long gp = pix[mt * 16 + (lane >> 4) * 4 + r];
                if (gp >= 0) {
                    float v = acc[mt][nt][r] * sc + bi;
                    if (relu) v = fmaxf(v, 0.f);
                    y[(size_t)gp * Cout + co] = v;
                }
            }
    }
}


// ---- stride-1 fast path: the input patch of the tile is staged ONCE in LDS ----------------------------------------
// The v2 kernel above gathers every (pixel, tap, corner) from L1/L2: 64 x 9 x 4 = 2304 corner reads of 512 B per
// workgroup, which made it L1-bandwidth bound at ~45 TFLOP/s.  Here the 14 x 14 input pixels around an 8 x 8 output
// tile (10 x 10 undeformed footprint + a 2-pixel halo for the learned offsets) are loaded once per 64-channel chunk
// (196 coalesced 256-byte rows, zero-filled outside the image) and all bilinear corners are read from LDS with
// conflict-free ds_read_b128 (16 lanes = the 64 channels of one patch pixel).  Samples whose corners leave the patch
// (|offset| > ~2 px) fall back to global loads for that (pixel, tap).  Slab, MFMA and epilogue as in v2.
constexpr int PCH = 64;              // channels per workgroup
constexpr int PR = 2;                // halo in pixels
constexpr int PS = 10 + 2 * PR;      // patch side
constexpr int LDP = PCH + 2;         // slab row stride (== 2 mod 32: conflict-free A-fragment reads, 8-byte aligned rows)

template <int CG>
__global__ __launch_bounds__(256, 2) void deform_conv3x3_patch_kernel(
    const float* __restrict__ x, const float* __restrict__ offset, const float* __restrict__ mask,
    const float* __restrict__ wp, const float* __restrict__ scale, const float* __restrict__ bias, int relu,
    int batch, int H, int W, int C, int Cout, int Ho, int Wo, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float patch[PS * PS * PCH];
    __shared__ __attribute__((aligned(16))) float col[TP * LDP];
    __shared__ float offs[TP * 18];
    __shared__ float msk[TP * 9];
    __shared__ int pix[TP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (Wo + 7) >> 3, tiles_y = (Ho + 7) >> 3;
    const int ntiles = batch * tiles_y * tiles_x;
    const int nchunks = C / PCH;
    int tile, chunk;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, slot = b >> 3;
        if ((nchunks & 7) == 0) {                    // whole channel chunks per XCD (see the v2 kernel)
            const int cpx = nchunks >> 3;
            chunk = xcd * cpx + slot % cpx;
            tile = slot / cpx;
        } else {
            chunk = b / ntiles;
            tile = b - chunk * ntiles;
        }
    }
    if (tile >= ntiles || chunk >= nchunks) return;
    const int tn = tile / (tiles_y * tiles_x);
    const int trem = tile - tn * tiles_y * tiles_x;
    const int tyy = trem / tiles_x, txx = trem - tyy * tiles_x;
    const int c0 = chunk * PCH;
    const int py0 = tyy * 8 - 1 - PR, px0 = txx * 8 - 1 - PR;       // image coordinates of patch pixel (0, 0)
    if (tid < TP) {
        const int ho = tyy * 8 + (tid >> 3), wo = txx * 8 + (tid & 7);
        pix[tid] = (ho < Ho && wo < Wo) ? (tn * Ho + ho) * Wo + wo : -1;
    }
    // ---- stage the patch (zero outside the image) ----
    for (int e = tid; e < PS * PS * (PCH / 4); e += 256) {
        const int pp = e >> 4, q = e & 15;
        const int r = pp / PS, cc = pp - r * PS;
        const int iy = py0 + r, ix = px0 + cc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W)
            v = *reinterpret_cast<const float4*>(x + ((size_t)(tn * H + iy) * W + ix) * C + c0 + q * 4);
        *reinterpret_cast<float4*>(&patch[pp * PCH + q * 4]) = v;
    }
    __syncthreads();
    for (int e = tid; e < TP * 18; e += 256) {
        const int p = e / 18;
        const int gp = pix[p];
        offs[e] = gp >= 0 ? offset[(size_t)gp * 18 + (e - 18 * p)] : 0.f;
    }
    for (int e = tid; e < TP * 9; e += 256) {
        const int p = e / 9;
        const int gp = pix[p];
        msk[e] = (gp >= 0) ? (mask ? mask[(size_t)gp * 9 + (e - 9 * p)] : 1.f) : 0.f;
    }
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int gq = tid & 15;            // float4 column of the 64-channel slab
    const int gp0 = tid >> 4;           // first pixel (pixels gp0, gp0 + 16, ...)
    const int co_l = wave * 16;         // this wave's 16 output channels inside the chunk
    const int g_l = co_l / CG;
    const int g = (c0 + co_l) / CG;
    const int co_g = (co_l % CG) + (lane & 15);
    for (int k = 0; k < 9; ++k) {
        const int kh = k / 3, kw = k - 3 * kh;
        // weights of this tap first: their L2 latency hides behind the gather phase instead of stalling the MFMAs
        float breg[CG / 4];
        {
            const float* wb = wp + ((size_t)(g * 9 + k) * CG) * CG + co_g;
#pragma unroll
            for (int kk = 0; kk < CG / 4; ++kk) breg[kk] = wb[(size_t)(kk * 4 + (lane >> 4)) * CG];
        }
        // ---- gather from the LDS patch ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = gp0 + 16 * i;
            const float ry = (float)((p >> 3) + kh + PR) + offs[p * 18 + 2 * k];       // patch coordinates
            const float rx = (float)((p & 7) + kw + PR) + offs[p * 18 + 2 * k + 1];
            const float h_im = ry + (float)py0, w_im = rx + (float)px0;
            const float m = msk[p * 9 + k];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m != 0.f && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const float fy = floorf(ry), fx = floorf(rx);
                const int hl = (int)fy, wl = (int)fx;
                const float lh = ry - fy, lw = rx - fx, uh = 1.f - lh, uw = 1.f - lw;
                if (hl >= 0 && hl < PS - 1 && wl >= 0 && wl < PS - 1) {
                    const float* pb = &patch[(hl * PS + wl) * PCH + gq * 4];
                    const float4 a = *reinterpret_cast<const float4*>(pb);
                    const float4 b = *reinterpret_cast<const float4*>(pb + PCH);
                    const float4 c = *reinterpret_cast<const float4*>(pb + PS * PCH);
                    const float4 d = *reinterpret_cast<const float4*>(pb + PS * PCH + PCH);
                    const float w1 = uh * uw, w2 = uh * lw, w3 = lh * uw, w4 = lh * lw;
                    v.x = w1 * a.x + w2 * b.x + w3 * c.x + w4 * d.x;
                    v.y = w1 * a.y + w2 * b.y + w3 * c.y + w4 * d.y;
                    v.z = w1 * a.z + w2 * b.z + w3 * c.z + w4 * d.z;
                    v.w = w1 * a.w + w2 * b.w + w3 * c.w + w4 * d.w;
                } else {                                  // large offset: corners from global memory
                    const int ih = hl + py0, iw = wl + px0;
                    const float wq[4] = {uh * uw, uh * lw, lh * uw, lh * lw};
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int yy = ih + (qd >> 1), xx = iw + (qd & 1);
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                            const float4 t = *reinterpret_cast<const float4*>(x + ((size_t)(tn * H + yy) * W + xx) * C + c0 + gq * 4);
                            v.x += wq[qd] * t.x; v.y += wq[qd] * t.y; v.z += wq[qd] * t.z; v.w += wq[qd] * t.w;
                        }
                    }
                }
                v.x *= m; v.y *= m; v.z *= m; v.w *= m;
            }
            float2* dd = reinterpret_cast<float2*>(&col[p * LDP + gq * 4]);
            dd[0] = make_float2(v.x, v.y);
            dd[1] = make_float2(v.z, v.w);
        }
        __syncthreads();
        // ---- MFMA: out[64 px][16 co of this wave] += slab[64 px][ci of the group] * W[ci][co] ----
#pragma unroll
        for (int kk = 0; kk < CG / 4; ++kk) {
            const int ci = kk * 4 + (lane >> 4);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float a = col[(mt * 16 + (lane & 15)) * LDP + g_l * CG + ci];
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[kk], acc[mt], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const int co = c0 + co_l + (lane & 15);
    const float sc = scale ? scale[co] : 1.f;
    const float bi = bias ? bias[co] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long gp = pix[mt * 16 + (lane >> 4) * 4 + r];
            if (gp >= 0) {
                float v = acc[mt][r] * sc + bi;
                if (relu) v = fmaxf(v, 0.f);
                y[(size_t)gp * Cout + co] = v;
            }
        }
}

// (C_out, C_in/groups, 3, 3) OIHW -> [group][tap][ci][co]
__global__ void pack_weight_kernel(const float* __restrict__ w, int cg, int cout, float* __restrict__ packed) {
    const long total = (long)cout * cg * 9;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        // e indexes packed: ((g*9 + k)*cg + ci)*cg + co
        const int co = (int)(e % cg);
        const int ci = (int)((e / cg) % cg);
        const int k = (int)((e / ((long)cg * cg)) % 9);
        const int g = (int)(e / ((long)cg * cg * 9));
        packed[e] = w[(((size_t)(g * cg + co) * cg) + ci) * 9 + k];
    }
}

}  // namespace

extern "C" {

size_t wd_deform_packed_weight_floats(int c_in, int c_out, int groups) {
    if (groups <= 0 || c_in % groups || c_in != c_out) return 0;
    return (size_t)c_out * (size_t)(c_in / groups) * 9;
}

int wd_deform_pack_weight(const float* weight_oihw, int c_in, int c_out, int groups, float* packed, void* stream) {
    WT_TRY(wt::ensure_device());
    if (!wd_deform_packed_weight_floats(c_in, c_out, groups)) {
        wt::set_error("wd_deform_pack_weight: needs c_in == c_out divisible by groups");
        return WT_ERR_INVALID;
    }
    hipLaunchKernelGGL(pack_weight_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, weight_oihw, c_in / groups, c_out, packed);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

int wd_deform_conv3x3_f32(const float* x, const float* offset, const float* mask, const float* packed_weight,
                          const float* scale, const float* bias, int relu, int batch, int h, int w, int c_in,
                          int c_out, int groups, int stride, int pad, float* y, void* stream_) {
    WT_TRY(wt::ensure_device());
    if (c_in != c_out || groups <= 0 || c_in % groups || c_in % CCH || stride < 1 || batch < 1 || h < 1 || w < 1 ||
        ((uintptr_t)x & 15)) {
        wt::set_error("wd_deform_conv3x3_f32: unsupported shape (c_in=%d c_out=%d groups=%d; need c_in == c_out, c_in %% 128 == 0)",
                      c_in, c_out, groups);
        return WT_ERR_INVALID;
    }
    if ((double)batch * h * w * c_in * 4.0 >= 4294967296.0) {
        wt::set_error("wd_deform_conv3x3_f32: input of %d x %d x %d x %d floats exceeds the kernel's 32-bit byte offsets (4 GiB); "
                      "split the batch", batch, h, w, c_in);
        return WT_ERR_CAPACITY;
    }
    const int cg = c_in / groups;
    const int ho = (h + 2 * pad - 3) / stride + 1, wo = (w + 2 * pad - 3) / stride + 1;
    if (ho < 1 || wo < 1) return WT_OK;
    // 1-D grid, padded to a multiple of 8 so that the (xcd, slot) decomposition covers every (tile, chunk) pair
    const long ntiles = (long)batch * ((ho + 7) / 8) * ((wo + 7) / 8);
    const long nwg = ntiles * (c_in / CCH);
    dim3 grid((unsigned)((nwg + 7) / 8 * 8));
    hipStream_t stream = (hipStream_t)stream_;
    // measured on MI355X (tools/deform_bench.py): the LDS-patch kernel wins for 64 channels per group (res5: 96 vs
    // 110 us) and loses to the L1-gather kernel for 16 / 32 (res3 / res4), where its smaller 64-channel chunks
    // expose the per-tap barrier latency; WD_DEFORM_PATCH=all|none overrides for experiments
    const char* mode = getenv("WD_DEFORM_PATCH");
    const bool want_patch = mode ? (strcmp(mode, "all") == 0) : (cg == 64);
    if (offset && stride == 1 && pad == 1 && (cg == 16 || cg == 32 || cg == 64) && want_patch) {
        const long nwg_p = ntiles * (c_in / PCH);
        dim3 gridp((unsigned)((nwg_p + 7) / 8 * 8));
#define WD_LAUNCH_P(CG)                                                                                              \
    hipLaunchKernelGGL(deform_conv3x3_patch_kernel<CG>, gridp, dim3(256), 0, stream, x, offset, mask, packed_weight, \
                       scale, bias, relu, batch, h, w, c_in, c_out, ho, wo, y)
        if (cg == 16) WD_LAUNCH_P(16);
        else if (cg == 32) WD_LAUNCH_P(32);
        else WD_LAUNCH_P(64);
#undef WD_LAUNCH_P
        WT_HIP(hipGetLastError());
        return WT_OK;
    }
#define WD_LAUNCH(CG)                                                                                                     \
    do {                                                                                                                  \
        if (offset)                                                                                                       \
            hipLaunchKernelGGL((deform_conv3x3_kernel<CG, true>), grid, dim3(256), 0, stream, x, offset, mask,           \
                               packed_weight, scale, bias, relu, batch, h, w, c_in, c_out, ho, wo, stride, pad, y);       \
        else                                                                                                              \
            hipLaunchKernelGGL((deform_conv3x3_kernel<CG, false>), grid, dim3(256), 0, stream, x, offset, mask,          \
                               packed_weight, scale, bias, relu, batch, h, w, c_in, c_out, ho, wo, stride, pad, y);       \
    } while (0)
    if (cg == 16) WD_LAUNCH(16);
    else if (cg == 32) WD_LAUNCH(32);
    else if (cg == 64) WD_LAUNCH(64);
    else if (cg == 8) WD_LAUNCH(8);
    else {
        wt::set_error("wd_deform_conv3x3_f32: channels per group must be 8, 16, 32 or 64 (got %d)", cg);
        return WT_ERR_INVALID;
    }
#undef WD_LAUNCH
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // extern "C"
