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

// ---- stride-1, 16 / 32 channels per group: LDS patch + sampling table + WAVE-PRIVATE slabs -------------------------
// PMC on the L1-gather kernel (tools/pmc_deform.sh, res4): texture path (TA 50 % / TD 66 % busy) and MFMA (37 %) take
// turns - every bilinear corner (64 px x 9 taps x 4) travels through the vector-memory return path although a tile only
// touches ~14 x 14 distinct input pixels.  Here each of those pixels crosses that path ONCE per 64-channel chunk
// (patch in LDS, zero-filled outside the image) and the 2304 corner reads are ds_read_b128 (4x the L1 width).
//   * sampling table (built once per workgroup): patch offset of corner (hl, wl) + the 4 bilinear weights (x mask);
//     samples whose corners leave the patch (|offset| > ~2 px) are flagged and read from global memory;
//   * a wave owns (group, pixel range): it blends ITS channels of ITS pixels into a private slab and multiplies them,
//     so the tap loop has no workgroup barrier; B operands of tap k+1 are requested before the MFMAs of tap k.
// tile pixel p (0..63) -> row: a 16-pixel M tile is rows (t, t + 4) x 8 columns, t = p >> 4.  With the 14-pixel patch pitch two
// ADJACENT rows land on bank groups shifted by 2 (6 of 16 lanes of a ds_read_b128 service group collide); rows 4 apart are
// shifted by 8 and share none.
#define PROW(p) (((p) >> 4) + 4 * (((p) >> 3) & 1))
constexpr int PST = PCH + 4;         // patch pixel stride in floats: 16 lanes reading 16 different pixels at the same
                                     // channel offset hit 16 different 4-bank groups (conflict-free ds_read_b128)
template <int CG>
__global__ __launch_bounds__(256, 2) void deform_conv3x3_lds_kernel(
    const float* __restrict__ x, const float* __restrict__ offset, const float* __restrict__ mask,
    const float* __restrict__ wp, const float* __restrict__ scale, const float* __restrict__ bias, int relu,
    int batch, int H, int W, int C, int Cout, int Ho, int Wo, float* __restrict__ y) {
    constexpr int GPW = PCH / CG;            // groups per workgroup (2 / 4)
    constexpr int WPG = 4 / GPW;             // waves per group (2 / 1)
    constexpr int MT = 4 / WPG;              // 16-pixel M tiles per wave (2 / 4)
    constexpr int NT = CG / 16;              // 16-channel N tiles per wave (2 / 1)
    constexpr int KS = CG / 4;               // k-steps per tap (8 / 4) = channels blended per lane
    constexpr int PW = 16 * MT;              // pixels per wave (32 / 64)
    __shared__ __attribute__((aligned(16))) float patch[PS * PS * PST];
    __shared__ __attribute__((aligned(16))) float4 tw[TP * 9];
    __shared__ int toff[TP * 9];
    __shared__ int pix[TP];
    __shared__ int any_fb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (Wo + 7) >> 3, tiles_y = (Ho + 7) >> 3;
    const int ntiles = batch * tiles_y * tiles_x;
    const int nchunks = C / PCH;
    int tile, chunk;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, slot = b >> 3;
        if ((nchunks & 7) == 0) {                    // whole channel chunks per XCD (see the L1-gather kernel)
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
    if (tid == 0) any_fb = 0;
    if (tid < TP) {
        const int ho = tyy * 8 + PROW(tid), wo = txx * 8 + (tid & 7);
        pix[tid] = (ho < Ho && wo < Wo) ? (tn * Ho + ho) * Wo + wo : -1;
    }
    // MFMA A fragments are blended straight into registers: lane (row r = lane & 15, kq = lane >> 4) owns pixel r of each
    // M tile and the KS consecutive channels [kq * KS, kq * KS + KS) of its group; k-step kk then multiplies channel
    // kq * KS + kk (a permutation of the reduction index, applied identically to the B operand below).
    const int g_l = wave / WPG;                          // group inside the chunk
    const int p_base = (wave % WPG) * PW;                // first pixel of this wave
    const int r16 = lane & 15, kq = lane >> 4;
    const int ch_l = g_l * CG + kq * KS;                 // first channel (inside the chunk) this lane blends
    const int g = (c0 + g_l * CG) / CG;                  // global group
    float bR[2][NT][KS];
    const float* __restrict__ wfrag = wp + (size_t)Cout * CG * 9 + ((size_t)g * 9 * 64 + lane) * (NT * KS);
    auto load_b = [&](int k, float (&b)[NT][KS]) {
        const float4* src = reinterpret_cast<const float4*>(wfrag + (size_t)k * 64 * (NT * KS));
#pragma unroll
        for (int j = 0; j < NT * KS / 4; ++j) {
            const float4 v = src[j];
            b[(4 * j) / KS][(4 * j) % KS + 0] = v.x; b[(4 * j) / KS][(4 * j) % KS + 1] = v.y;
            b[(4 * j) / KS][(4 * j) % KS + 2] = v.z; b[(4 * j) / KS][(4 * j) % KS + 3] = v.w;
        }
    };
    load_b(0, bR[0]);              // tap 0's weights: their L2 latency hides behind the patch fill and the table build
    // ---- table inputs (offsets / mask of up to 3 (pixel, tap) entries per thread) are requested before the patch
    //      loads, so the prologue pays ONE global round trip ----
    constexpr int NE = (TP * 9 + 255) / 256;
    float e_dy[NE], e_dx[NE], e_m[NE];
    int e_gp[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int e = tid + 256 * j;
        const int p = e / 9, k = e - 9 * p;
        int gp = -1;
        if (e < TP * 9) {
            const int ho = tyy * 8 + PROW(p), wo = txx * 8 + (p & 7);
            gp = (ho < Ho && wo < Wo) ? (tn * Ho + ho) * Wo + wo : -1;
        }
        e_gp[j] = gp;
        e_dy[j] = e_dx[j] = 0.f;
        e_m[j] = 1.f;
        if (gp >= 0) {
            e_dy[j] = offset[(size_t)gp * 18 + 2 * k];
            e_dx[j] = offset[(size_t)gp * 18 + 2 * k + 1];
            if (mask) e_m[j] = mask[(size_t)gp * 9 + k];
        }
    }
    // ---- stage the patch (zero outside the image): all loads of a thread in flight, then the LDS stores ----
    {
        constexpr int NL = (PS * PS * (PCH / 4) + 255) / 256;
        float4 v[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = tid + 256 * j;
            const int pp = e >> 4, q = e & 15;
            const int r = pp / PS, cc = pp - r * PS;
            const int iy = py0 + r, ix = px0 + cc;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < PS * PS * (PCH / 4) && iy >= 0 && iy < H && ix >= 0 && ix < W)
                v[j] = *reinterpret_cast<const float4*>(x + ((size_t)(tn * H + iy) * W + ix) * C + c0 + q * 4);
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = tid + 256 * j;
            if (e < PS * PS * (PCH / 4)) *reinterpret_cast<float4*>(&patch[(e >> 4) * PST + (e & 15) * 4]) = v[j];
        }
    }
    __syncthreads();                                  // any_fb = 0 above is ordered before the table's any_fb = 1
    // ---- sampling table ----
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int e = tid + 256 * j;
        if (e >= TP * 9) continue;
        const int p = e / 9, k = e - 9 * p;
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        int off = 0;
        if (e_gp[j] >= 0) {
            const int kh = k / 3, kw = k - 3 * kh;
            const float ry = (float)(PROW(p) + kh + PR) + e_dy[j];        // patch coordinates
            const float rx = (float)((p & 7) + kw + PR) + e_dx[j];
            const float h_im = ry + (float)py0, w_im = rx + (float)px0;
            const float m = e_m[j];
            if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const float fy = floorf(ry), fx = floorf(rx);
                const int hl = (int)fy, wl = (int)fx;
                const float lh = ry - fy, lw = rx - fx, uh = 1.f - lh, uw = 1.f - lw;
                w4 = make_float4(uh * uw * m, uh * lw * m, lh * uw * m, lh * lw * m);
                if (hl >= 0 && hl < PS - 1 && wl >= 0 && wl < PS - 1) {
                    off = (hl * PS + wl) * PST;
                } else {                          // corners outside the patch: image coordinates packed for the slow path
                    off = -1 - ((hl + py0 + 2) * (W + 4) + (wl + px0 + 2));
                    any_fb = 1;
                }
            }
        }
        tw[e] = w4;
        toff[e] = off;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const bool slow = any_fb != 0;

    auto mfma = [&](const float (&a)[MT][KS], const float (&b)[NT][KS]) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][kk], b[nt][kk], acc[mt][nt], 0, 0, 0);
        }
    };
    if (!slow) {
        // Software pipeline inside the wave: the LDS reads of tap k+1 are issued BEFORE the MFMA burst of tap k and blended
        // after it has been issued, so corner latency and blend VALU run under the matrix pipe's 32-cycle instructions.
        // (Waves of a SIMD issue their MFMAs round-robin and stay in phase: without this, every wave gathers while the
        // matrix pipe idles and vice versa - measured: t(gather only) + t(MFMA only) == t(kernel).)
        float4 cv[MT][4][KS / 4];
        float4 wv[MT];
        float a[2][MT][KS];
        auto gather_issue = [&](int k) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int e = (p_base + mt * 16 + r16) * 9 + k;
                const float* pb = &patch[toff[e] + ch_l];
                wv[mt] = tw[e];
#pragma unroll
                for (int h = 0; h < KS / 4; ++h) {
                    cv[mt][0][h] = *reinterpret_cast<const float4*>(pb + 4 * h);
                    cv[mt][1][h] = *reinterpret_cast<const float4*>(pb + PST + 4 * h);
                    cv[mt][2][h] = *reinterpret_cast<const float4*>(pb + PS * PST + 4 * h);
                    cv[mt][3][h] = *reinterpret_cast<const float4*>(pb + PS * PST + PST + 4 * h);
                }
            }
        };
        auto blend = [&](float (&o)[MT][KS]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int h = 0; h < KS / 4; ++h) {
                    o[mt][4 * h + 0] = wv[mt].x * cv[mt][0][h].x + wv[mt].y * cv[mt][1][h].x + wv[mt].z * cv[mt][2][h].x + wv[mt].w * cv[mt][3][h].x;
                    o[mt][4 * h + 1] = wv[mt].x * cv[mt][0][h].y + wv[mt].y * cv[mt][1][h].y + wv[mt].z * cv[mt][2][h].y + wv[mt].w * cv[mt][3][h].y;
                    o[mt][4 * h + 2] = wv[mt].x * cv[mt][0][h].z + wv[mt].y * cv[mt][1][h].z + wv[mt].z * cv[mt][2][h].z + wv[mt].w * cv[mt][3][h].z;
                    o[mt][4 * h + 3] = wv[mt].x * cv[mt][0][h].w + wv[mt].y * cv[mt][1][h].w + wv[mt].z * cv[mt][2][h].w + wv[mt].w * cv[mt][3][h].w;
                }
        };
        gather_issue(0);
        blend(a[0]);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (k + 1 < 9) load_b(k + 1, bR[(k + 1) & 1]);
            if (k + 1 < 9) gather_issue(k + 1);
            mfma(a[k & 1], bR[k & 1]);
            if (k + 1 < 9) blend(a[(k + 1) & 1]);
            if (k + 1 < 9) {
                // interleave inside the MFMA burst (a wave issues in order: VALU / LDS instructions only overlap the matrix
                // pipe when they sit BETWEEN two MFMAs): first the LDS reads of tap k+1, then its blend VALU
                constexpr int NM = MT * NT * KS;            // MFMAs per tap (32 / 16)
                constexpr int NDS = MT * (2 + KS);          // LDS reads per tap
                constexpr int H1 = NM / 4;
                __builtin_amdgcn_sched_group_barrier(0x020, NT * KS / 4, 0);   // B operand loads first (oldest in vmcnt order)
#pragma unroll
                for (int j = 0; j < H1; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, (NDS + H1 - 1) / H1, 0);
                }
#pragma unroll
                for (int j = 0; j < NM / 4; ++j) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                for (int j = 0; j < NM - H1 - NM / 4; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (MT * KS * 4 + (NM - H1 - NM / 4) - 1) / (NM - H1 - NM / 4), 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // large offsets somewhere in the tile: per-sample choice between the patch and global memory (not pipelined)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (k + 1 < 9) load_b(k + 1, bR[(k + 1) & 1]);
            float a[MT][KS];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int e = (p_base + mt * 16 + r16) * 9 + k;
                const int off = toff[e];
                const float4 w4 = tw[e];
                const float wq[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) a[mt][kk] = 0.f;
                if (off >= 0) {
                    const float* pb = &patch[off + ch_l];
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                        for (int h = 0; h < KS / 4; ++h) {
                            const float4 t = *reinterpret_cast<const float4*>(pb + (qd >> 1) * PS * PST + (qd & 1) * PST + 4 * h);
                            a[mt][4 * h + 0] += wq[qd] * t.x; a[mt][4 * h + 1] += wq[qd] * t.y;
                            a[mt][4 * h + 2] += wq[qd] * t.z; a[mt][4 * h + 3] += wq[qd] * t.w;
                        }
                } else {
                    const int code = -1 - off;
                    const int ih = code / (W + 4) - 2, iw = code - (ih + 2) * (W + 4) - 2;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int yy = ih + (qd >> 1), xx = iw + (qd & 1);
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                            const float* gb = x + ((size_t)(tn * H + yy) * W + xx) * C + c0 + ch_l;
#pragma unroll
                            for (int h = 0; h < KS / 4; ++h) {
                                const float4 t = *reinterpret_cast<const float4*>(gb + 4 * h);
                                a[mt][4 * h + 0] += wq[qd] * t.x; a[mt][4 * h + 1] += wq[qd] * t.y;
                                a[mt][4 * h + 2] += wq[qd] * t.z; a[mt][4 * h + 3] += wq[qd] * t.w;
                            }
                        }
                    }
                }
            }
            mfma(a, bR[k & 1]);
        }
    }
    // ---- epilogue ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = c0 + g_l * CG + 16 * nt + (lane & 15);
        const float sc = scale ? scale[co] : 1.f;
        const float bi = bias ? bias[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long gp = pix[p_base + mt * 16 + (lane >> 4) * 4 + r];
                if (gp >= 0) {
                    float v = acc[mt][nt][r] * sc + bi;
                    if (relu) v = fmaxf(v, 0.f);
                    y[(size_t)gp * Cout + co] = v;
                }
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
        const float v = w[(((size_t)(g * cg + co) * cg) + ci) * 9 + k];
        packed[e] = v;
        // second copy, MFMA-B-fragment order of the LDS kernel (16 / 32 channels per group): lane (kq = ci / KS, r16 = co % 16)
        // holds its NT x KS values contiguously, so a tap's B operands are NT*KS/4 dwordx4 loads per lane (1 KiB coalesced per
        // wave instruction) instead of NT*KS strided dword loads
        if (cg == 16 || cg == 32) {
            const int KS = cg / 4, NT = cg / 16;
            const int kq = ci / KS, kk = ci - kq * KS, nt = co >> 4, r16 = co & 15;
            const int lane = kq * 16 + r16;
            packed[total + (((size_t)(g * 9 + k) * 64 + lane) * NT + nt) * KS + kk] = v;
        }
    }
}

}  // namespace

int wt::victim_regs_deform64() {
    hipFuncAttributes at{};
    if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(deform_conv3x3_kernel<64, true>)) != hipSuccess) return 0;
    return (at.numRegs + 7) / 8 * 8;
}


int wd_deform_pp_launch(const float* x, const float* offset, const float* packed_weight, const float* scale,
                        const float* bias, int relu, int batch, int h, int w, int c, int cg, int stride, hipStream_t stream, float* y,
                        const void* table);
int wd_grouped_conv3x3_c8_launch(const float* x, const float* packed_weight, const float* scale, const float* bias, int relu,
                                 int batch, int h, int w, int c, hipStream_t stream, float* y);

// 0 = L1-gather kernel, 1 = LDS patch + shared slab, 2 = LDS patch + register A fragments, 3 = ping-pong (det_deform_pp.hip)
static int deform_variant(int cg, int stride, int pad, bool has_offset, const char* mode) {
    const bool fits = has_offset && stride == 1 && pad == 1;
    if (mode && strcmp(mode, "none") == 0) return 0;
    // stride 2, 16 / 32 channels per group (round 3): the persistent kernel with its patch over the middle of the 17 x 17 footprint and
    // the far path for the rest: res3 322 -> ~125 us, res4 160 -> ~90 us
    if (has_offset && stride == 2 && pad == 1 && (cg == 32 || cg == 16) && !(mode && strcmp(mode, "lds") == 0)) return 3;
    if (mode && strcmp(mode, "all") == 0) return (fits && (cg == 16 || cg == 32 || cg == 64)) ? 1 : 0;
    if (fits && (cg == 32 || cg == 16) && !(mode && strcmp(mode, "lds") == 0)) return 3;      // persistent kernel (falls back to 2 with a mask)
    if (fits && (cg == 16 || cg == 32)) return 2;
    if (fits && cg == 64) return 1;
    return 0;
}

extern "C" {

const char* wd_deform_conv3x3_variant(int c_in, int groups, int stride, int pad, int has_offset) {
    if (groups <= 0 || c_in % groups) return "invalid";
    const int cg = c_in / groups;
    switch (deform_variant(cg, stride, pad, has_offset != 0, getenv("WD_DEFORM_PATCH"))) {
        case 3: return cg == 16 ? "deform_conv3x3_pp_kernel<16>" : "deform_conv3x3_pp_kernel<32>";
        case 2: return cg == 16 ? "deform_conv3x3_lds_kernel<16>" : "deform_conv3x3_lds_kernel<32>";
        case 1: return cg == 16 ? "deform_conv3x3_patch_kernel<16>" : cg == 32 ? "deform_conv3x3_patch_kernel<32>" : "deform_conv3x3_patch_kernel<64>";
        default: break;
    }
    if (!has_offset && cg == 8 && stride == 1 && pad == 1 && (c_in % 128) == 0) return "grouped_conv3x3_c8_kernel";
    if (has_offset) return cg == 8 ? "deform_conv3x3_kernel<8,true>" : cg == 16 ? "deform_conv3x3_kernel<16,true>" : cg == 32 ? "deform_conv3x3_kernel<32,true>" : "deform_conv3x3_kernel<64,true>";
    return cg == 8 ? "deform_conv3x3_kernel<8,false>" : cg == 16 ? "deform_conv3x3_kernel<16,false>" : cg == 32 ? "deform_conv3x3_kernel<32,false>" : "deform_conv3x3_kernel<64,false>";
}

size_t wd_deform_packed_weight_floats(int c_in, int c_out, int groups) {
    if (groups <= 0 || c_in % groups || c_in != c_out) return 0;
    return 2 * (size_t)c_out * (size_t)(c_in / groups) * 9;      // [group][tap][ci][co] + the lane-major fragment copy
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
    return wd_deform_conv3x3_tab_f32(x, offset, mask, packed_weight, scale, bias, relu, batch, h, w, c_in, c_out, groups, stride, pad,
                                     0, nullptr, y, stream_);
}

int wd_deform_conv3x3_hint_f32(const float* x, const float* offset, const float* mask, const float* packed_weight,
                               const float* scale, const float* bias, int relu, int batch, int h, int w, int c_in,
                               int c_out, int groups, int stride, int pad, int far_offsets, float* y, void* stream_) {
    return wd_deform_conv3x3_tab_f32(x, offset, mask, packed_weight, scale, bias, relu, batch, h, w, c_in, c_out, groups, stride, pad,
                                     far_offsets, nullptr, y, stream_);
}

int wd_deform_conv3x3_tab_f32(const float* x, const float* offset, const float* mask, const float* packed_weight,
                              const float* scale, const float* bias, int relu, int batch, int h, int w, int c_in,
                              int c_out, int groups, int stride, int pad, int far_offsets, const void* table, float* y,
                              void* stream_) {
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
    // dispatch measured on MI355X (tools/deform_bench.py, bench.py): stride 1 with offsets -> LDS-patch kernels
    // (16 / 32 channels per group: register-fragment kernel, 64: shared-slab patch kernel); stride 2, plain grouped
    // convolution and 8 channels per group -> L1-gather kernel.  WD_DEFORM_PATCH=lds|all|none overrides (experiments).
    const char* mode = getenv("WD_DEFORM_PATCH");
    // res2: plain grouped conv with 8 channels per group -> vector-unit kernel (det_gconv.hip); WD_GCONV=mfma keeps the MFMA path
    if (!offset && !mask && cg == 8 && stride == 1 && pad == 1 && (c_in % 128) == 0 && ((uintptr_t)y & 7) == 0) {
        const char* gm = getenv("WD_GCONV");
        if (!(gm && strcmp(gm, "mfma") == 0))
            return wd_grouped_conv3x3_c8_launch(x, packed_weight, scale, bias, relu, batch, h, w, c_in, stream, y);
    }
    int variant = deform_variant(cg, stride, pad, offset != nullptr, mode);
    if (variant == 3 && stride != 1 && mask) variant = 0;    // stride 2 + modulation mask: the gather kernel
    if (variant == 3 && stride == 1 && (mask || far_offsets)) variant = 2;  // the ping-pong kernel has no modulation mask; with many samples leaving
                                                             // the 14x14 patch its per-lane far path loses to the per-tile one
    if (variant == 3)
        return wd_deform_pp_launch(x, offset, packed_weight, scale, bias, relu, batch, h, w, c_in, cg, stride, stream, y, table);
    if (variant == 2) {
        const long nwg_p = ntiles * (c_in / PCH);
        dim3 gridp((unsigned)((nwg_p + 7) / 8 * 8));
        if (cg == 16)
            hipLaunchKernelGGL(deform_conv3x3_lds_kernel<16>, gridp, dim3(256), 0, stream, x, offset, mask, packed_weight, scale,
                               bias, relu, batch, h, w, c_in, c_out, ho, wo, y);
        else
            hipLaunchKernelGGL(deform_conv3x3_lds_kernel<32>, gridp, dim3(256), 0, stream, x, offset, mask, packed_weight, scale,
                               bias, relu, batch, h, w, c_in, c_out, ho, wo, y);
        WT_HIP(hipGetLastError());
        return WT_OK;
    }
    if (variant == 1) {
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
