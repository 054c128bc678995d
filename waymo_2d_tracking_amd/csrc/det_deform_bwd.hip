// Fused deformable-conv backward for the training graph, pad 1, stride 1 (default) or 2, 16 or 32 channels per
// group, gfx950.
// detectron2's CUDA op (deformable_im2col -> GEMM -> deformable_col2im / col2im_coord) materialises the 9*C*P column slab three times per
// layer; det_backward.hip restates that form (5 passes over the slab, gather-bound through L1: tools/deform_bwd_bench.py).  Here the columns
// never leave the CU:
//   * deform_dw_kernel   : dW[g][o][k][ci] = sum_p dY[p][o] * col[p][k][ci].  A workgroup (3 waves, wave = kernel row kh) owns one group
//     and a slice of the 8x8-pixel tiles; per tile it stages the 14x14xCG input patch (zero-filled outside the image) and the sampling
//     table of the tile in LDS, blends the column fragment of (4 pixels x 16 channels) in registers and feeds it to
//     v_mfma_f32_16x16x4_f32 as the B operand with K = pixels; dY is the A operand straight from global memory (one 8-byte load per
//     k-step, reused by the wave's three taps).  The 9 x CG x CG accumulators stay in registers across the slice, go out as coalesced
//     partial sums and deform_dw_reduce_kernel adds the slices into the weight's own OIHW layout (float atomics with this access
//     pattern cost 44 us per layer).  Samples whose corners leave the patch (|offset| > ~2 px) are zero in the main loop and added by
//     a second pass with their corners from global memory (only for tiles that have such samples).
//   * dX / dOffset: deform_bwd_tables_kernel, deform_dxoff_kernel, deform_bwd_far_kernel - see the comment in front of namespace tt.
// Same arithmetic as det_backward.hip up to the summation order (tests/test_gpu_detops.py compares both with the float64 restatement).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/waymodet.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// Stride 1 (pad 1): output = input size, the 14 x 14 patch of a tile starts at image pixel 8 t - 3 (2-pixel halo around the 10 x 10 footprint
// of the undeformed taps).  Stride 2: the footprint is 17 x 17, the patch holds its rows / columns 1 .. 14 (origin 16 t + 1) and the samples of
// the outer taps (about a third) are "far" - the same split as the forward kernel makes.
struct Geo {
    int H, W;        // input (x, dX)
    int Ho, Wo;      // output (dY, offset, dOffset); tiles are 8 x 8 output pixels
    int S, org;      // stride; patch origin = image pixel S * 8 * t - org (org = 3 for stride 1, -1 for stride 2)
};

namespace fb {
constexpr int PS = 14;                 // patch side: 8 + 2 (3x3 footprint) + 2 * 2 (halo for the learned offsets)
constexpr int NPIX = PS * PS;          // 196; pixel 196 = zeros (samples / pixels outside the image)
constexpr int ZERO = NPIX;
constexpr int NE = 64 * 9;             // (pixel, tap) entries per tile
constexpr int XPAD = 4;                // deform_dxoff_kernel: floats of padding per patch pixel (a lane reads 16 bytes of ITS pixel's corner: with a
                                       // pitch of CG floats the 16 lanes of a ds_read_b128 group would share 4 bank slots)
}  // namespace fb

// Sampling entry of (tile pixel (yy, xx), tap (kh, kw)).  Patch origin = image pixel (8 ty - 3, 8 tx - 3).
//   x, y : BYTE offsets of the four corners inside the patch buffer (16 bits each; the zero pixel for a sample outside (-1, H) x (-1, W))
//   z, w : the bilinear fractions lh, lw
// A sample whose corners are not all inside the patch ("far", |offset| > ~2 px) points at the zero pixel as well - the main loops stay
// branch-free - and leaves far = (row + 32768) | (column + 32768) << 16 of its upper-left corner in IMAGE coordinates (otherwise 0) for a
// second pass that only runs for tiles with such samples.
template <unsigned PB>                     // PB = bytes per patch pixel in the LDS patch buffer
__device__ __forceinline__ uint4 fb_entry(bool pixel_in_image, int yy, int xx, int kh, int kw, float oy, float ox, int ty, int tx, const Geo& G,
                                          unsigned& far) {
    const int H = G.H, W = G.W;
    unsigned c0 = fb::ZERO, c1 = fb::ZERO, c2 = fb::ZERO, c3 = fb::ZERO;
    float lh = 0.f, lw = 0.f;
    far = 0;
    if (pixel_in_image) {
        const float ry = (float)(G.S * yy + kh - 1 + G.org) + oy, rx = (float)(G.S * xx + kw - 1 + G.org) + ox;          // patch coordinates
        const float h_im = ry + (float)(ty * 8 * G.S - G.org), w_im = rx + (float)(tx * 8 * G.S - G.org);
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
            const float fy = floorf(ry), fx = floorf(rx);
            const int hl = (int)fy, wl = (int)fx;
            lh = ry - fy; lw = rx - fx;
            if ((unsigned)hl <= (unsigned)(fb::PS - 2) && (unsigned)wl <= (unsigned)(fb::PS - 2)) {
                const unsigned u = hl * fb::PS + wl;
                c0 = u; c1 = u + 1; c2 = u + fb::PS; c3 = u + fb::PS + 1;
            } else {
                far = (unsigned)(hl + ty * 8 * G.S - G.org + 32768) | ((unsigned)(wl + tx * 8 * G.S - G.org + 32768) << 16);
            }
        }
    }
    uint4 e;
    e.x = (c0 * PB) | ((c1 * PB) << 16); e.y = (c2 * PB) | ((c3 * PB) << 16);
    e.z = __float_as_uint(lh); e.w = __float_as_uint(lw);
    return e;
}

template <int V> struct fb_vec { using type = float; };
template <> struct fb_vec<2> { using type = f32x2; };

template <int V> __device__ __forceinline__ typename fb_vec<V>::type fb_zero() { typename fb_vec<V>::type z = {}; return z; }
__device__ __forceinline__ float fb_mask(float g, float y) { return y > 0.f ? g : 0.f; }
__device__ __forceinline__ f32x2 fb_mask(f32x2 g, f32x2 y) { return f32x2{y[0] > 0.f ? g[0] : 0.f, y[1] > 0.f ? g[1] : 0.f}; }
__device__ __forceinline__ f32x4 fb_mask(f32x4 g, f32x4 y) {
    return f32x4{y[0] > 0.f ? g[0] : 0.f, y[1] > 0.f ? g[1] : 0.f, y[2] > 0.f ? g[2] : 0.f, y[3] > 0.f ? g[3] : 0.f};
}
__device__ __forceinline__ float fb_at(float v, int) { return v; }
__device__ __forceinline__ float fb_at(f32x2 v, int i) { return v[i]; }

// corner (ih, iw) of a far sample from global memory (zero outside the image)
template <int V>
__device__ __forceinline__ typename fb_vec<V>::type fb_far_corner(const float* __restrict__ xg, int ih, int iw, int H, int W, int C) {
    using vec = typename fb_vec<V>::type;
    if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) return *reinterpret_cast<const vec*>(xg + ((size_t)ih * W + iw) * C);
    return fb_zero<V>();
}

// Stage the tile's input patch (one group) and sampling table in LDS.  NTHR threads.  farflag[kh] != 0: kernel row kh has far samples.
template <int CG, int NTHR>
__device__ __forceinline__ void fb_stage(const float* __restrict__ x, const float* __restrict__ offset, int tn, int ty, int tx, const Geo& G, int C,
                                         int c0, float* __restrict__ xs, uint4* __restrict__ tab, unsigned* __restrict__ farpos,
                                         int* __restrict__ farflag, int tid) {
    constexpr int Q = CG / 4;
    const int H = G.H, W = G.W;
    for (int i = tid; i < fb::NPIX * Q; i += NTHR) {
        const int pp = i / Q, q = i - pp * Q;
        const int r = pp / fb::PS;
        const int iy = G.S * 8 * ty - G.org + r, ix = G.S * 8 * tx - G.org + (pp - r * fb::PS);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = *reinterpret_cast<const f32x4*>(x + ((size_t)(tn * H + iy) * W + ix) * C + c0 + q * 4);
        *reinterpret_cast<f32x4*>(xs + pp * CG + q * 4) = v;
    }
    for (int e = tid; e < fb::NE; e += NTHR) {
        const int p = e / 9, k = e - 9 * p;
        const int yy = p >> 3, xx = p & 7, oy = 8 * ty + yy, ox = 8 * tx + xx;
        const bool in = oy < G.Ho && ox < G.Wo;
        float2 ov = make_float2(0.f, 0.f);
        if (in) ov = *reinterpret_cast<const float2*>(offset + ((size_t)(tn * G.Ho + oy) * G.Wo + ox) * 18 + 2 * k);
        const int kh = k / 3;
        unsigned far;
        tab[k * 64 + p] = fb_entry<CG * 4>(in, yy, xx, kh, k - 3 * kh, ov.x, ov.y, ty, tx, G, far);
        farpos[k * 64 + p] = far;
        if (far) farflag[kh] = 1;
    }
}

// col[pixel of entry e][ci = V * n .. V * n + V - 1]: bilinear blend of the four corners (xl = the lane's channel inside the patch buffer)
template <int V>
__device__ __forceinline__ typename fb_vec<V>::type fb_sample(const uint4 e, const char* __restrict__ xl) {
    using vec = typename fb_vec<V>::type;
    const float lh = __uint_as_float(e.z), lw = __uint_as_float(e.w), uh = 1.f - lh, uw = 1.f - lw;
    const vec v0 = *reinterpret_cast<const vec*>(xl + (e.x & 0xFFFFu));
    const vec v1 = *reinterpret_cast<const vec*>(xl + (e.x >> 16));
    const vec v2 = *reinterpret_cast<const vec*>(xl + (e.y & 0xFFFFu));
    const vec v3 = *reinterpret_cast<const vec*>(xl + (e.y >> 16));
    return (uh * uw) * v0 + (uh * lw) * v1 + (lh * uw) * v2 + (lh * lw) * v3;
}

// the same value for a far sample, corners from global memory (xg = image tn, the lane's channel)
template <int V>
__device__ __forceinline__ typename fb_vec<V>::type fb_sample_far(const uint4 e, unsigned far, const float* __restrict__ xg, int H, int W, int C) {
    using vec = typename fb_vec<V>::type;
    const float lh = __uint_as_float(e.z), lw = __uint_as_float(e.w), uh = 1.f - lh, uw = 1.f - lw;
    const int ih = (int)(far & 0xFFFFu) - 32768, iw = (int)(far >> 16) - 32768;
    const vec v0 = fb_far_corner<V>(xg, ih, iw, H, W, C), v1 = fb_far_corner<V>(xg, ih, iw + 1, H, W, C);
    const vec v2 = fb_far_corner<V>(xg, ih + 1, iw, H, W, C), v3 = fb_far_corner<V>(xg, ih + 1, iw + 1, H, W, C);
    return (uh * uw) * v0 + (uh * lw) * v1 + (lh * uw) * v2 + (lh * lw) * v3;
}

template <int CG>
__global__ __launch_bounds__(192) void deform_dw_kernel(const float* __restrict__ x, const float* __restrict__ offset, const float* __restrict__ dy,
                                                        const float* __restrict__ yact, const float* __restrict__ scale,
                                                        int batch, Geo geo, int C, int slices, float* __restrict__ part) {
    constexpr int MT = CG / 16;                  // 16-wide tiles along o and along ci; a lane holds MT consecutive channels (o = MT i + mt)
    using vec = typename fb_vec<MT>::type;
    __shared__ __attribute__((aligned(16))) float xs[(fb::NPIX + 1) * CG];
    __shared__ uint4 tab[fb::NE];
    __shared__ unsigned farpos[fb::NE];
    __shared__ int farflag[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, j = lane >> 4;
    const int G = C / CG;
    const int g = blockIdx.x % G, slice = blockIdx.x / G;
    const int H = geo.H, W = geo.W, Ho = geo.Ho, Wo = geo.Wo;
    const int tiles_x = (Wo + 7) >> 3, tiles_y = (Ho + 7) >> 3, ntiles = batch * tiles_y * tiles_x;
    f32x4 acc[3][MT][MT];   // experiments: -DFB_NO_EPILOGUE (no atomics), -DFB_NO_STAGE (stage the first tile only)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < CG) xs[fb::ZERO * CG + tid] = 0.f;
    // backward of the block's fused epilogue on the fly: dY_eff = dY * (y > 0) * scale[channel] (yact / scale may be null)
    vec sc = fb_zero<MT>();
    if (scale) sc = *reinterpret_cast<const vec*>(scale + g * CG + MT * n);
    for (int tile = slice; tile < ntiles; tile += slices) {
        const int tn = tile / (tiles_y * tiles_x), trem = tile - tn * tiles_y * tiles_x;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        __syncthreads();                                   // the previous tile's readers are done
        if (tid < 4) farflag[tid] = 0;
        __syncthreads();
#ifdef FB_NO_STAGE
        if (tile == slice)
#endif
        fb_stage<CG, 192>(x, offset, tn, ty, tx, geo, C, g * CG, xs, tab, farpos, farflag, tid);
        // A fragments: dY[pixel 4 s + j][o = MT n + mt]  (K = pixels: k-step s covers 4 consecutive pixels of a tile row)
        vec a[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int oy = 8 * ty + (s >> 1), ox = 8 * tx + 4 * (s & 1) + j;
            a[s] = fb_zero<MT>();
            if (oy < Ho && ox < Wo) {
                const size_t at = ((size_t)(tn * Ho + oy) * Wo + ox) * C + g * CG + MT * n;
                a[s] = *reinterpret_cast<const vec*>(dy + at);
                if (yact) a[s] = fb_mask(a[s], *reinterpret_cast<const vec*>(yact + at));
                if (scale) a[s] = a[s] * sc;
            }
        }
        __syncthreads();
        const char* xl = reinterpret_cast<const char*>(xs + MT * n);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const uint4* tp = tab + (3 * wave + t) * 64 + j;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const vec b = fb_sample<MT>(tp[4 * s], xl);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < MT; ++nt)
                        acc[t][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb_at(a[s], mt), fb_at(b, nt), acc[t][mt][nt], 0, 0, 0);
            }
        }
        if (farflag[wave]) {                               // rare: samples outside the patch, corners from global memory (zero in the pass above)
            const float* xg = x + (size_t)tn * H * W * C + g * CG + MT * n;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                for (int s = 0; s < 16; ++s) {
                    const int ei = (3 * wave + t) * 64 + 4 * s + j;
                    const unsigned far = farpos[ei];
                    if (!__builtin_amdgcn_ballot_w64(far != 0)) continue;
                    vec b = fb_zero<MT>();
                    if (far) b = fb_sample_far<MT>(tab[ei], far, xg, H, W, C);
                    const vec as = a[s];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < MT; ++nt)
                            acc[t][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb_at(as, mt), fb_at(b, nt), acc[t][mt][nt], 0, 0, 0);
                }
            }
        }
    }
#ifdef FB_NO_EPILOGUE
    if (acc[0][0][0][0] != 123.456f) return;
#endif
    // partial sums of this workgroup, in register order (256-byte stores): part[slice][g][wave][t][mt][nt][r][lane]
    float* __restrict__ pw = part + ((size_t)blockIdx.x * 3 + wave) * (3 * MT * MT * 4 * 64) + lane;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < MT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pw[(((t * MT + mt) * MT + nt) * 4 + r) * 64] = acc[t][mt][nt][r];
}

// dW[g CG + o][ci][tap] (OIHW) = sum over slices of the partials.  D[i = 4 j + r][n] of tile (mt, nt) = dW[o = MT i + mt][tap][ci = MT n + nt].
template <int CG>
__global__ __launch_bounds__(256) void deform_dw_reduce_kernel(const float* __restrict__ part, int G, int slices, float* __restrict__ dw) {
    constexpr int MT = CG / 16, PER = 9 * CG * CG;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= G * PER) return;
    const int g = e / PER, idx = e - g * PER;
    float s = 0.f;
    for (int sl = 0; sl < slices; ++sl) s += part[((size_t)sl * G + g) * PER + idx];
    const int lane = idx & 63, r = (idx >> 6) & 3;
    int q = idx >> 8;
    const int nt = q % MT; q /= MT;
    const int mt = q % MT; q /= MT;
    const int t = q % 3, wave = q / 3;
    const int o = MT * (4 * (lane >> 4) + r) + mt, ci = MT * (lane & 15) + nt;
    dw[(((size_t)g * CG + o) * CG + ci) * 9 + 3 * wave + t] = s;               // OIHW, the layout of the weight itself
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dX and dOffset.  Three launches per layer:
//   deform_bwd_tables_kernel : per 8x8 tile, from the offsets only (shared by all groups): the sampling table, the INVERTED table - for
//     every pixel of the 14x14 input patch the list of (sample, corner) pairs that touch it with a non-zero weight (count with LDS integer
//     atomics, scan, fill) - and the patch pixels ordered by list length (so that the lanes of a wave gather lists of similar length);
//   deform_dxoff_kernel      : persistent workgroups (6 waves) over (tile, group) items.  Per item: the group's input patch -> LDS;
//     dcol^T = W^T dY^T on the MFMAs (A = packed weights from L2, B = dY from global memory, K = output channels of the group): a lane
//     ends up with 4 consecutive input channels of ITS pixel and tap, so the four corners it needs for dOffset are four ds_read_b128 and
//     the fragment goes to LDS as one ds_write_b128; then every patch pixel sums its list out of LDS (4 lanes x 4 channels per pixel)
//     and issues ONE global float atomic per (patch pixel, channel) - the patches of neighbouring tiles overlap.  dcol lives in LDS for 16
//     channels x 576 samples at a time (36 KB): two workgroups per CU.  dOffset accumulates in registers over the groups of an item range
//     and is added to global memory once per tile and workgroup.
//   deform_bwd_far_kernel    : the samples whose corners leave the patch (zero-weight in the kernel above), one wave per (sample, group),
//     plain dot products and per-corner atomics; exits at once for tiles without such samples.
// (LDS float atomics run at ~194 cycles per wave instruction on gfx950 (tools/micro/lds_atomic_rate.hip) - a scatter into an LDS patch
// accumulator is not an option; coalesced global float atomics cost ~5 us per 14 M, measured on det_backward.hip's gather kernel.)
namespace tt {                             // per-tile tables in global memory (byte offsets); the first LDS_BYTES are copied to LDS as they are
constexpr int NROW = fb::NE + 1;           // row 576 = the dummy sample (zero weights) that pads the lists to multiples of 4 entries
constexpr int NINV = 2304 + 3 * fb::NPIX + 4;          // 2896
constexpr int TAB = 0;                     // uint4[577]  sampling entries, index tap * 64 + pixel
constexpr int INV = TAB + NROW * 16;       // u16[2896]   ((row * 4 + swizzle(row)) << 2 | corner), grouped by patch pixel, 4 entries (8 bytes) aligned
constexpr int START = INV + NINV * 2;      // u16[200]    list of patch pixel pp = inv[start[pp] .. start[pp + 1])
constexpr int ORDER = START + 400;         // u16[200]    patch pixels by descending list length
constexpr int LDS_BYTES = ORDER + 400;     // 15 824
constexpr int FARPOS = LDS_BYTES;          // u32[576]
constexpr int NFAR = FARPOS + fb::NE * 4;  // u32
constexpr int BYTES = NFAR + 32;
static_assert(LDS_BYTES % 16 == 0 && INV % 8 == 0 && BYTES % 16 == 0, "table alignment");
}  // namespace tt

// (the same launch also zeroes dX / dOffset and packs the weights: blocks >= the number of tiles, see below)
template <int CG>
__global__ __launch_bounds__(256) void deform_bwd_tables_kernel(const float* __restrict__ offset, int batch, Geo geo, unsigned char* __restrict__ tbl,
                                                                const float* __restrict__ weight, int C, float* __restrict__ wpk,
                                                                float* __restrict__ dx, float* __restrict__ doff) {
    const int H = geo.H, W = geo.W, Ho = geo.Ho, Wo = geo.Wo;
    {
        const int ntiles = batch * ((Ho + 7) >> 3) * ((Wo + 7) >> 3);
        if ((int)blockIdx.x >= ntiles) {
            // auxiliary blocks: dX = 0 (the gather adds into it), and the weights (C_out, CG, 3, 3) OIHW in MFMA A-fragment order:
            // wpk[g][tap][mt][lane = 16 j + n][s] = W[g CG + (CG / 4) j + s][mt 16 + n][tap]
            constexpr int MT = CG / 16, KS = CG / 4;
            const long nb = gridDim.x - ntiles, b = blockIdx.x - ntiles;
            const long n4 = (long)batch * H * W * C / 4;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            for (long i = b * 256 + threadIdx.x; i < n4; i += nb * 256) reinterpret_cast<f32x4*>(dx)[i] = z;
            const long nw = (long)C * 9 * CG;
            for (long e = b * 256 + threadIdx.x; e < nw; e += nb * 256) {
                const int s = (int)(e % KS);
                long q = e / KS;
                const int lane = (int)(q & 63); q >>= 6;
                const int mt = (int)(q % MT); q /= MT;
                const int tap = (int)(q % 9), g = (int)(q / 9);
                const int o = KS * (lane >> 4) + s, ci = mt * 16 + (lane & 15);
                wpk[e] = weight[((size_t)(g * CG + o) * CG + ci) * 9 + tap];
            }
            return;
        }
    }
    __shared__ uint4 tab[tt::NROW];
    __shared__ unsigned farpos[fb::NE];
    __shared__ int cnt[fb::NPIX], start[fb::NPIX + 1], cursor[fb::NPIX], hist[64], hcur[64];
    __shared__ unsigned short inv[tt::NINV], order[fb::NPIX];
    __shared__ int nfar;
    constexpr unsigned PB = (CG + fb::XPAD) * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (Wo + 7) >> 3, tiles_y = (Ho + 7) >> 3;
    const int tile = blockIdx.x;
    const int tn = tile / (tiles_y * tiles_x), trem = tile - tn * tiles_y * tiles_x;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    for (int i = tid; i < fb::NPIX; i += 256) cnt[i] = 0;
    if (tid < 64) hist[tid] = 0;
    if (tid == 0) nfar = 0;
    __syncthreads();
    // which corners of entry e take part in the dx gather: inside the image, non-zero weight
    auto corners = [&](const uint4 e, unsigned (&pp)[4], bool (&use)[4]) {
        const float lh = __uint_as_float(e.z), lw = __uint_as_float(e.w), uh = 1.f - lh, uw = 1.f - lw;
        const float wq[4] = {uh * uw, uh * lw, lh * uw, lh * lw};
        pp[0] = (e.x & 0xFFFFu) / PB; pp[1] = (e.x >> 16) / PB; pp[2] = (e.y & 0xFFFFu) / PB; pp[3] = (e.y >> 16) / PB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = pp[q] / fb::PS, cc = pp[q] - r * fb::PS;
            const int iy = geo.S * 8 * ty - geo.org + r, ix = geo.S * 8 * tx - geo.org + cc;
            use[q] = pp[q] != (unsigned)fb::ZERO && wq[q] != 0.f && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        }
    };
    for (int e = tid; e < fb::NE; e += 256) {
        const int p = e / 9, k = e - 9 * p;
        const int yy = p >> 3, xx = p & 7, oy = 8 * ty + yy, ox = 8 * tx + xx;
        const bool in = oy < Ho && ox < Wo;
        float2 ov = make_float2(0.f, 0.f);
        if (in) ov = *reinterpret_cast<const float2*>(offset + ((size_t)(tn * Ho + oy) * Wo + ox) * 18 + 2 * k);
        const int kh = k / 3;
        unsigned far;
        const uint4 en = fb_entry<PB>(in, yy, xx, kh, k - 3 * kh, ov.x, ov.y, ty, tx, geo, far);
        tab[k * 64 + p] = en;
        farpos[k * 64 + p] = far;
        if (far) atomicAdd(&nfar, 1);
        unsigned pp[4]; bool use[4];
        corners(en, pp, use);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (use[q]) atomicAdd(&cnt[pp[q]], 1);
    }
    __syncthreads();
    if (wave == 0) {                           // exclusive scan of the 196 list lengths, each padded to a multiple of 4 (4 per lane + wave scan)
        int c[4], s4 = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = lane * 4 + u;
            c[u] = idx < fb::NPIX ? (cnt[idx] + 3) & ~3 : 0;
            s4 += c[u];
        }
        int inc = s4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        int base = inc - s4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = lane * 4 + u;
            if (idx < fb::NPIX) { start[idx] = base; cursor[idx] = base; }
            base += c[u];
        }
        if (lane == 63) start[fb::NPIX] = inc;
    }
    if (tid == 0) {                            // the dummy sample: zero weights (lh = lw = 0, corner 3), corners on the zero pixel
        uint4 z;
        z.x = (fb::ZERO * PB) | ((fb::ZERO * PB) << 16); z.y = z.x; z.z = 0u; z.w = 0u;
        tab[fb::NE] = z;
    }
    for (int i = tid; i < fb::NPIX; i += 256) atomicAdd(&hist[cnt[i] < 63 ? cnt[i] : 63], 1);
    __syncthreads();
    if (wave == 0) {                           // longest lists first: hcur[c] = number of patch pixels with a longer list
        const int h = hist[lane];
        int inc = h;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_down(inc, o, 64);
            if (lane + o < 64) inc += t;
        }
        hcur[lane] = inc - h;
    }
    for (int row = tid; row < fb::NE; row += 256) {
        unsigned pp[4]; bool use[4];
        corners(tab[row], pp, use);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (use[q]) inv[atomicAdd(&cursor[pp[q]], 1)] = (unsigned short)((((row << 2) | ((row >> 1) & 3)) << 2) | q);
    }
    __syncthreads();
    for (int i = tid; i < fb::NPIX; i += 256)
        for (int k = cursor[i]; k < start[i + 1]; ++k) inv[k] = (unsigned short)(((fb::NE << 2) << 2) | 3);
    for (int i = tid; i < fb::NPIX; i += 256) order[atomicAdd(&hcur[cnt[i] < 63 ? cnt[i] : 63], 1)] = (unsigned short)i;
    __syncthreads();
    unsigned char* out = tbl + (size_t)tile * tt::BYTES;
    for (int i = tid; i < tt::NROW; i += 256) {
        reinterpret_cast<uint4*>(out + tt::TAB)[i] = tab[i];
        if (i < fb::NE) reinterpret_cast<unsigned*>(out + tt::FARPOS)[i] = farpos[i];
    }
    for (int i = tid; i < tt::NINV; i += 256) reinterpret_cast<unsigned short*>(out + tt::INV)[i] = i < start[fb::NPIX] ? inv[i] : (unsigned short)0;
    for (int i = tid; i < 200; i += 256) {
        reinterpret_cast<unsigned short*>(out + tt::START)[i] = (unsigned short)(i <= fb::NPIX ? start[i] : 0);
        reinterpret_cast<unsigned short*>(out + tt::ORDER)[i] = i < fb::NPIX ? order[i] : (unsigned short)0;
    }
    if (tid == 0) *reinterpret_cast<unsigned*>(out + tt::NFAR) = (unsigned)nfar;
    for (int i = tid; i < 64 * 18; i += 256) {             // dOffset of the tile = 0 (the dX / dOffset kernels add into it)
        const int pixel = i / 18, oy = 8 * ty + (pixel >> 3), ox = 8 * tx + (pixel & 7);
        if (oy < Ho && ox < Wo) doff[((size_t)(tn * Ho + oy) * Wo + ox) * 18 + (i - pixel * 18)] = 0.f;
    }
}

// keeps the loads of the next (tap, pixel tile) unit from being hoisted over this one (register pressure: 3 waves per SIMD = 168 VGPRs)
#define FB_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// -DFB_TIMING: cycles per phase, summed over wave 0 of every workgroup (tools only; read back with wd_deform_fb_ticks)
#ifdef FB_TIMING
__device__ unsigned long long fb_ticks[8];
#define FB_T(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (tid0 == 0) atomicAdd(&fb_ticks[k], now_ - t_last); t_last = now_; } while (0)
#else
#define FB_T(k) do { } while (0)
#endif
#ifndef FB_GU
#define FB_GU 4
#endif
// Workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() carries a fence that would also wait for the global float
// atomics of the gather (fire-and-forget adds into dX; nothing in the kernel reads them back) - a memory round trip per phase.
#ifdef FB_FULL_BARRIER
#define FB_BARRIER() __syncthreads()
#else
#define FB_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
#endif
#ifndef FB_WAVES
#define FB_WAVES __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
template <int CG> constexpr size_t dxoff_smem_bytes() { return (size_t)(fb::NPIX + 1) * (CG + fb::XPAD) * 4 + tt::NROW * 16 * 4 + tt::LDS_BYTES; }

template <int CG>
__global__ __launch_bounds__(384) FB_WAVES void deform_dxoff_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    const float* __restrict__ yact, const float* __restrict__ scale, const float* __restrict__ wpk,
                                                           const unsigned char* __restrict__ tbl, int batch, Geo geo, int C, int items_total,
                                                           float* __restrict__ dx, float* __restrict__ doff) {
    constexpr int MT = CG / 16, KS = CG / 4, KQ = KS / 4;       // k-steps of 4 output channels; float4s per operand fragment
    constexpr int XP = CG + fb::XPAD;                           // patch pitch in floats
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* xs = reinterpret_cast<float*>(smem);                                   // [197][XP]
    float* dc = xs + (fb::NPIX + 1) * XP;                                         // [576 rows][16 channels], 16-byte slots XOR-swizzled by row
    unsigned char* tl = reinterpret_cast<unsigned char*>(dc + tt::NROW * 16);     // the tile's tables (tt:: layout)
    const uint4* tab = reinterpret_cast<const uint4*>(tl + tt::TAB);
    const unsigned short* inv = reinterpret_cast<const unsigned short*>(tl + tt::INV);
    const unsigned short* start = reinterpret_cast<const unsigned short*>(tl + tt::START);
    const unsigned short* order = reinterpret_cast<const unsigned short*>(tl + tt::ORDER);
    const int tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int kh = wave % 3, half = wave / 3;
    const int G = C / CG;
    const int H = geo.H, W = geo.W, Ho = geo.Ho, Wo = geo.Wo;
    const int tiles_x = (Wo + 7) >> 3, tiles_y = (Ho + 7) >> 3;
    int tid = tid0, lane = tid0 & 63, n = lane & 15, j = lane >> 4;
    // work is split at the granularity of a 16-channel phase (MT per item): 2240 items over 512 workgroups would be 4 or 5 items each (the 5 set
    // the time), 4480 phases are 8 or 9.  A workgroup's first / last item can be a partial one; two workgroups then stage the same patch.
    const long ph_total = (long)items_total * MT;
    const long p0 = (long)blockIdx.x * ph_total / gridDim.x, p1 = (long)(blockIdx.x + 1) * ph_total / gridDim.x;
    if (p0 >= p1) return;
    const long i0 = p0 / MT, i1 = (p1 + MT - 1) / MT;
    int cur_tile = -1, tn = 0, ty = 0, tx = 0;
    float od[3][2][2];
#ifdef FB_TIMING
    unsigned long long t_last = __builtin_readcyclecounter();
#endif
    if (tid < CG) xs[fb::ZERO * XP + tid] = 0.f;
    if (tid < 16) dc[fb::NE * 16 + tid] = 0.f;                 // the dummy sample's dcol row (list padding)

    // dOffset of the finished tile: the four channel quarters (j) of every (pixel, tap) meet in LDS, one global atomic per value
    auto flush = [&]() {
        FB_BARRIER();
        float* sc = dc;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
                for (int d = 0; d < 2; ++d) sc[((wave * 12) + (t * 2 + p2) * 2 + d) * 64 + lane] = od[t][p2][d];
        FB_BARRIER();
        for (int i = tid; i < 64 * 18; i += 384) {
            const int pixel = i / 18, c18 = i - pixel * 18;
            const int tap = c18 >> 1, d = c18 & 1, fkh = tap / 3, t = tap - 3 * fkh;
            const int pt = pixel >> 4, pn = pixel & 15;
            const float* q = sc + (((pt >> 1) * 3 + fkh) * 12 + (t * 2 + (pt & 1)) * 2 + d) * 64 + pn;
            const float v = (q[0] + q[16]) + (q[32] + q[48]);
            const int oy = 8 * ty + (pixel >> 3), ox = 8 * tx + (pixel & 7);
            if (oy < Ho && ox < Wo) atomicAdd(doff + ((size_t)(tn * Ho + oy) * Wo + ox) * 18 + c18, v);
        }
    };

    // prefetch registers: the NEXT item's patch (written to LDS after the item barrier) and dY fragments are requested before the last gather of
    // the current item, the weights of the next (tap, channel half) before the MFMAs of the current one - with 3 waves per SIMD nothing else
    // hides the L2 / HBM latency
    constexpr int Q = CG / 4, NXP = (fb::NPIX * Q + 383) / 384;
    f32x4 xp[NXP], bq[2][KQ], aqn[KQ];
    auto load_item = [&](int ptile, int pg) {
        const int ptn = ptile / (tiles_y * tiles_x), prem = ptile - ptn * tiles_y * tiles_x;
        const int pty = prem / tiles_x, ptx = prem - pty * tiles_x;
#pragma unroll
        for (int u = 0; u < NXP; ++u) {
            const int i = tid + 384 * u;
            const int pp = i / Q, q = i - pp * Q;
            const int r = pp / fb::PS;
            const int iy = geo.S * 8 * pty - geo.org + r, ix = geo.S * 8 * ptx - geo.org + (pp - r * fb::PS);
            xp[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < fb::NPIX * Q && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                xp[u] = *reinterpret_cast<const f32x4*>(x + ((size_t)(ptn * H + iy) * W + ix) * C + pg * CG + q * 4);
        }
        // B fragments: dY[pixel (2 half + p2) 16 + n][o = KS j + s]
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            const int pixel = (2 * half + p2) * 16 + n;
            const int oy = 8 * pty + (pixel >> 3), ox = 8 * ptx + (pixel & 7);
            const bool in = oy < Ho && ox < Wo;
#pragma unroll
            for (int u = 0; u < KQ; ++u) {
                bq[p2][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (in) {
                    const size_t at = ((size_t)(ptn * Ho + oy) * Wo + ox) * C + pg * CG + KS * j + 4 * u;
                    bq[p2][u] = *reinterpret_cast<const f32x4*>(dy + at);
                    // backward of the block's fused epilogue on the fly: dY_eff = dY * (y > 0) * scale[channel]
                    if (yact) bq[p2][u] = fb_mask(bq[p2][u], *reinterpret_cast<const f32x4*>(yact + at));
                    if (scale) bq[p2][u] = bq[p2][u] * *reinterpret_cast<const f32x4*>(scale + pg * CG + KS * j + 4 * u);
                }
            }
        }
    };
    auto load_w = [&](int pg, int pmt, int ptap) {
        const f32x4* wp = reinterpret_cast<const f32x4*>(wpk + ((((size_t)pg * 9 + ptap) * MT + pmt) * 64 + lane) * KS);
#pragma unroll
        for (int u = 0; u < KQ; ++u) aqn[u] = wp[u];
    };
    if (i0 < i1) {
        const int ftile = (int)(i0 / G), fg = (int)(i0 - (long)ftile * G);
        load_item(ftile, fg);
        load_w(fg, (int)(p0 - i0 * MT), 3 * kh);
    }
    for (long item = i0; item < i1; ++item) {
        // the lane id goes through an opaque asm once per item: otherwise LLVM hoists ~60 registers of lane-dependent addresses out of this
        // loop and the kernel no longer fits 3 waves per SIMD (168 VGPRs) without spilling
        tid = tid0;
        asm volatile("" : "+v"(tid));
        lane = tid & 63; n = lane & 15; j = lane >> 4;
        const int tile = (int)(item / G), g = (int)(item - (long)tile * G);
        const int ntile = (int)((item + 1) / G), ng = (int)(item + 1 - (long)ntile * G);       // the next item (prefetch)
        const bool more = item + 1 < i1;
        const int mt_begin = (int)(p0 > item * MT ? p0 - item * MT : 0), mt_end = (int)(p1 - item * MT < MT ? p1 - item * MT : MT);
        if (tile != cur_tile) {
            if (cur_tile >= 0) flush();
            FB_BARRIER();
            const uint4* src = reinterpret_cast<const uint4*>(tbl + (size_t)tile * tt::BYTES);
            for (int i = tid; i < tt::LDS_BYTES / 16; i += 384) reinterpret_cast<uint4*>(tl)[i] = src[i];
            cur_tile = tile;
            tn = tile / (tiles_y * tiles_x);
            const int trem = tile - tn * tiles_y * tiles_x;
            ty = trem / tiles_x; tx = trem - ty * tiles_x;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) od[t][p2][0] = od[t][p2][1] = 0.f;
        }
        FB_BARRIER();                                   // the previous item's gather is done with xs / dc
        FB_T(0);
#pragma unroll
        for (int u = 0; u < NXP; ++u) {
            const int i = tid + 384 * u;
            const int pp = i / Q, q = i - pp * Q;
            if (i < fb::NPIX * Q) *reinterpret_cast<f32x4*>(xs + pp * XP + q * 4) = xp[u];
        }
        FB_BARRIER();
        FB_T(1);
#pragma unroll 1
        for (int mt = mt_begin; mt < mt_end; ++mt) {
            const char* cb = reinterpret_cast<const char*>(xs + mt * 16 + 4 * j);
            const int row0 = 3 * kh * 64 + 2 * half * 16 + n;                     // unit (t, p2): row0 + 64 t + 16 p2
            uint4 en = tab[row0];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int tap = 3 * kh + t;
                f32x4 aq[KQ];
#pragma unroll
                for (int u = 0; u < KQ; ++u) aq[u] = aqn[u];
                if (t < 2) load_w(g, mt, tap + 1);
                else if (mt + 1 < mt_end) load_w(g, mt + 1, 3 * kh);
                else if (more) load_w(ng, 0, 3 * kh);                            // only a workgroup's first item can start at a later phase
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    const int row = row0 + 64 * t + 16 * p2;
#ifndef FB_NO_DOFF
                    // the unit's LDS requests first (they land under the MFMA chain), then the next unit's table entry
                    const uint4 e = en;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(cb + (e.x & 0xFFFFu));
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(cb + (e.x >> 16));
                    const f32x4 v2 = *reinterpret_cast<const f32x4*>(cb + (e.y & 0xFFFFu));
                    const f32x4 v3 = *reinterpret_cast<const f32x4*>(cb + (e.y >> 16));
                    if (t * 2 + p2 < 5) en = tab[row0 + 64 * ((t * 2 + p2 + 1) >> 1) + 16 * ((t * 2 + p2 + 1) & 1)];
                    asm volatile("" ::: "memory");
#endif
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < KQ; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[u][v], bq[p2][u][v], acc, 0, 0, 0);
                    // acc[r] = dcol[pixel][tap][ci = mt 16 + 4 j + r]
#ifndef FB_NO_DOFF
                    const float lh = __uint_as_float(e.z), lw = __uint_as_float(e.w), uh = 1.f - lh, uw = 1.f - lw;
                    // d val / d h = (v2 - v0)(1 - lw) + (v3 - v1) lw ;  d val / d w = (v1 - v0)(1 - lh) + (v3 - v2) lh, summed over the channels with
                    // dcol as the weight: four dot products s_q = <dcol, v_q> (16 FMAs; VALU time adds to the MFMA time on this machine), then
                    // the blend of their differences
                    const float s0 = (acc[0] * v0[0] + acc[1] * v0[1]) + (acc[2] * v0[2] + acc[3] * v0[3]);
                    const float s1 = (acc[0] * v1[0] + acc[1] * v1[1]) + (acc[2] * v1[2] + acc[3] * v1[3]);
                    const float s2 = (acc[0] * v2[0] + acc[1] * v2[1]) + (acc[2] * v2[2] + acc[3] * v2[3]);
                    const float s3 = (acc[0] * v3[0] + acc[1] * v3[1]) + (acc[2] * v3[2] + acc[3] * v3[3]);
                    od[t][p2][0] += (s2 - s0) * uw + (s3 - s1) * lw;
                    od[t][p2][1] += (s1 - s0) * uh + (s3 - s2) * lh;
                    asm volatile("" : "+v"(od[t][p2][0]), "+v"(od[t][p2][1]));     // or LLVM sinks this arithmetic (and the 20 registers of every unit
                                                                                  // it needs) below the gather
#endif
                    *reinterpret_cast<f32x4*>(dc + row * 16 + ((j ^ ((row >> 1) & 3)) << 2)) = acc;
                    FB_FENCE();
                }
            }
            FB_T(2);
            FB_BARRIER();
            FB_T(3);
            if (mt + 1 == mt_end && more) load_item(ntile, ng);    // in flight during the gather; bq is dead after the last MFMA phase
            // gather: patch pixel pp sums its list; 4 lanes x 4 channels per patch pixel, longest lists first
#ifndef FB_NO_GATHER       // experiments (wrong results): -DFB_NO_GATHER, -DFB_NO_DOFF, -DFB_NO_DXATOMIC
            {
                const int c = tid & 3;
#pragma unroll 1
                for (int round = 0; round < 3; ++round) {
                    // 16 patch pixels per wave and round; the order is by descending list length: round 0 hands chunk w to wave w, round 1 chunk
                    // 11 - w (the wave with the longest lists gets the shortest next), round 2 the last 4 (nearly empty) lists
                    const int chunk = round == 0 ? wave : (round == 1 ? 11 - wave : 12 + wave);
                    const int task = chunk * 16 + ((tid & 63) >> 2);
                    if (task < fb::NPIX) {
                        const int pp = order[task];
                        const int n0 = start[pp], n1 = start[pp + 1];
                        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                        for (int i = n0; i < n1; i += 4) {                 // lists are padded to 4 entries (8 bytes, aligned) with zero-weight samples
                            const uint2 e4 = *reinterpret_cast<const uint2*>(inv + i);
                            const unsigned ent[4] = {e4.x & 0xFFFFu, e4.x >> 16, e4.y & 0xFFFFu, e4.y >> 16};
                            float2 l[4];
                            f32x4 v[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                l[u] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(tab) + (ent[u] & 0xFFF0u) + 8);
                                v[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(dc) + (((ent[u] >> 2) ^ c) << 4));
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float wy = (ent[u] & 2) ? l[u].x : 1.f - l[u].x, wx = (ent[u] & 1) ? l[u].y : 1.f - l[u].y;
                                a += (wy * wx) * v[u];
                            }
                        }
                        if (n1 > n0) {
                            const int r = pp / fb::PS;
                            const int iy = geo.S * 8 * ty - geo.org + r, ix = geo.S * 8 * tx - geo.org + (pp - r * fb::PS);
                            float* d = dx + ((size_t)(tn * H + iy) * W + ix) * C + g * CG + mt * 16 + 4 * c;
#ifdef FB_NO_DXATOMIC
                            *reinterpret_cast<f32x4*>(d) = a;
#else
                            atomicAdd(d + 0, a[0]); atomicAdd(d + 1, a[1]); atomicAdd(d + 2, a[2]); atomicAdd(d + 3, a[3]);
#endif
                        }
                    }
                }
            }
#endif
            FB_T(4);
            if (mt + 1 < mt_end) FB_BARRIER();
            FB_T(5);
        }
    }
    if (cur_tile >= 0) flush();
}

// Samples whose corners are not all inside the tile's patch: dcol row by plain dot products (lane = input channel of the group; the packed
// weights make the loads 16 bytes per lane), per-corner global atomics for dX, dOffset reduced over the channel lanes.  grid (tiles, 8 group
// chunks); 64 / CG samples per wave at a time.
template <int CG>
__global__ __launch_bounds__(256) void deform_bwd_far_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ yact,
                                                             const float* __restrict__ scale, const float* __restrict__ wpk,
                                                             const unsigned char* __restrict__ tbl, int batch, Geo geo, int C,
                                                             float* __restrict__ dx, float* __restrict__ doff) {
    const int H = geo.H, W = geo.W, Ho = geo.Ho, Wo = geo.Wo;
    const unsigned char* tb = tbl + (size_t)blockIdx.x * tt::BYTES;
    if (*reinterpret_cast<const unsigned*>(tb + tt::NFAR) == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = C / CG;
    const int tiles_x = (Wo + 7) >> 3, tiles_y = (Ho + 7) >> 3;
    const int tile = blockIdx.x;
    const int tn = tile / (tiles_y * tiles_x), trem = tile - tn * tiles_y * tiles_x;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const unsigned* farpos = reinterpret_cast<const unsigned*>(tb + tt::FARPOS);
    const uint4* tab = reinterpret_cast<const uint4*>(tb + tt::TAB);
    // 64 / CG samples per wave at a time: lane = (sample slot, input channel of the group)
    constexpr int R = 64 / CG, MT = CG / 16, KS = CG / 4, KQ = KS / 4;
    const int sub = lane / CG, ci = lane % CG;
    __shared__ int nlist;
    __shared__ unsigned short list[fb::NE];
    if (tid == 0) nlist = 0;
    __syncthreads();
    for (int row = tid; row < fb::NE; row += 256)
        if (farpos[row]) list[atomicAdd(&nlist, 1)] = (unsigned short)row;
    __syncthreads();
    const int nl = nlist;
    for (int lb = wave * R; lb < nl; lb += 4 * R) {
        const bool valid = lb + sub < nl;
        const int row = list[valid ? lb + sub : lb];
        const unsigned far = farpos[row];
        const int tap = row >> 6, pixel = row & 63;
        const int oy = 8 * ty + (pixel >> 3), ox = 8 * tx + (pixel & 7);        // inside the image (far is only set for such pixels)
        const uint4 e = tab[row];
        const float lh = __uint_as_float(e.z), lw = __uint_as_float(e.w), uh = 1.f - lh, uw = 1.f - lw;
        const int ih = (int)(far & 0xFFFFu) - 32768, iw = (int)(far >> 16) - 32768;
        const float wq[4] = {uh * uw, uh * lw, lh * uw, lh * lw};
        for (int g = blockIdx.y; g < G; g += gridDim.y) {
            // dcol[ci] = sum_o dY_eff[o] W[o][tap][ci]: the packed weights (MFMA fragment order) give 16 bytes per lane and 4 output channels,
            // 512 contiguous bytes per 16 lanes; dY is one 16-byte broadcast load per sample slot
            const size_t dat = ((size_t)(tn * Ho + oy) * Wo + ox) * C + g * CG;
            const float* wp = wpk + ((((size_t)g * 9 + tap) * MT + (ci >> 4)) * 64 + (ci & 15)) * KS;
            float gcol = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < KQ; ++u) {
                    const int o0 = KS * j + 4 * u;
                    f32x4 d = *reinterpret_cast<const f32x4*>(dy + dat + o0);
                    if (yact) d = fb_mask(d, *reinterpret_cast<const f32x4*>(yact + dat + o0));
                    if (scale) d = d * *reinterpret_cast<const f32x4*>(scale + g * CG + o0);
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wp + (16 * j) * KS + 4 * u);
                    gcol += (d[0] * w4[0] + d[1] * w4[1]) + (d[2] * w4[2] + d[3] * w4[3]);
                }
            if (!valid) gcol = 0.f;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int yy = ih + (q >> 1), xx = iw + (q & 1);
                const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                const size_t at = ((size_t)(tn * H + (in ? yy : 0)) * W + (in ? xx : 0)) * C + g * CG + ci;
                v[q] = in ? x[at] : 0.f;
                if (in && valid && wq[q] != 0.f) atomicAdd(dx + at, wq[q] * gcol);
            }
            float dh = gcol * ((v[2] - v[0]) * uw + (v[3] - v[1]) * lw);
            float dw = gcol * ((v[1] - v[0]) * uh + (v[3] - v[2]) * lh);
#pragma unroll
            for (int o = CG / 2; o > 0; o >>= 1) { dh += __shfl_xor(dh, o, 64); dw += __shfl_xor(dw, o, 64); }
            if (ci == 0 && valid) {
                atomicAdd(doff + ((size_t)(tn * Ho + oy) * Wo + ox) * 18 + 2 * tap, dh);
                atomicAdd(doff + ((size_t)(tn * Ho + oy) * Wo + ox) * 18 + 2 * tap + 1, dw);
            }
        }
    }
}

int check_fused(const char* who, int c, int groups, int h, int w, int stride) {
    const int cg = groups > 0 ? c / groups : 0;
    if (groups < 1 || c % groups || (cg != 16 && cg != 32) || h < 1 || w < 1 || (stride != 1 && stride != 2)) {
        wt::set_error("%s: 16 or 32 channels per group, stride 1 or 2 only (C=%d groups=%d stride=%d)", who, c, groups, stride);
        return WT_ERR_INVALID;
    }
    return WT_OK;
}

// 3 x 3, pad 1: output size and patch origin of a stride
Geo make_geo(int h, int w, int stride) {
    Geo g;
    g.H = h; g.W = w; g.S = stride;
    g.Ho = (h + 2 - 3) / stride + 1; g.Wo = (w + 2 - 3) / stride + 1;
    g.org = stride == 1 ? 3 : -1;
    return g;
}
int geo_tiles(const Geo& g, int batch) { return batch * ((g.Ho + 7) / 8) * ((g.Wo + 7) / 8); }

// Tile slices per group: every workgroup resident at once (4 per CU at 32 channels per group - 34 KB of LDS, 3 waves each; 6 at 16), the
// tiles spread evenly over the slices.
int fused_dw_slices(int ntiles, int groups, int cg) {
    const int target = (cg == 32 ? 4 : 6) * 256;
    int per_wg = (int)(((long)ntiles * groups + target - 1) / target);
    if (const char* e = getenv("WD_DW_TILES_PER_WG")) per_wg = atoi(e);
    if (per_wg < 1) per_wg = 1;
    return (ntiles + per_wg - 1) / per_wg;
}

}  // namespace

extern "C" {

size_t wd_deform_dw_scratch_floats(int batch, int h, int w, int c, int groups, int stride) {
    if (groups < 1 || c % groups || stride < 1) return 0;
    const int cg = c / groups;
    return (size_t)fused_dw_slices(geo_tiles(make_geo(h, w, stride), batch), groups, cg) * groups * 9 * cg * cg;
}

int wd_deform_dw_f32(const float* x, const float* offset, const float* dy, const float* y_act, const float* scale, int batch, int h, int w, int c,
                     int groups, int stride, float* scratch, float* dw, void* stream) {
    WT_TRY(wt::ensure_device());
    WT_TRY(check_fused("wd_deform_dw_f32", c, groups, h, w, stride));
    const int cg = c / groups;
    hipStream_t st = (hipStream_t)stream;
    const Geo geo = make_geo(h, w, stride);
    const int slices = fused_dw_slices(geo_tiles(geo, batch), groups, cg);
    const int nred = (groups * 9 * cg * cg + 255) / 256;
    if (cg == 32) {
        hipLaunchKernelGGL(deform_dw_kernel<32>, dim3((unsigned)(groups * slices)), dim3(192), 0, st, x, offset, dy, y_act, scale, batch, geo, c, slices, scratch);
        hipLaunchKernelGGL(deform_dw_reduce_kernel<32>, dim3((unsigned)nred), dim3(256), 0, st, scratch, groups, slices, dw);
    } else {
        hipLaunchKernelGGL(deform_dw_kernel<16>, dim3((unsigned)(groups * slices)), dim3(192), 0, st, x, offset, dy, y_act, scale, batch, geo, c, slices, scratch);
        hipLaunchKernelGGL(deform_dw_reduce_kernel<16>, dim3((unsigned)nred), dim3(256), 0, st, scratch, groups, slices, dw);
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

#ifdef FB_TIMING
int wd_deform_fb_ticks(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(fb_ticks), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(fb_ticks), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

size_t wd_deform_bwd_tables_bytes(int batch, int h, int w, int stride) {
    return stride < 1 ? 0 : (size_t)geo_tiles(make_geo(h, w, stride), batch) * tt::BYTES;
}

int wd_deform_dxoff_f32(const float* x, const float* offset, const float* dy, const float* y_act, const float* scale, const float* weight, int batch,
                        int h, int w, int c, int groups, int stride, unsigned char* tables, float* packed_weight, float* dx, float* doffset,
                        void* stream) {
    WT_TRY(wt::ensure_device());
    WT_TRY(check_fused("wd_deform_dxoff_f32", c, groups, h, w, stride));
    const int cg = c / groups;
    hipStream_t st = (hipStream_t)stream;
    const Geo geo = make_geo(h, w, stride);
    const int ntiles = geo_tiles(geo, batch);
    const int items = ntiles * groups;
    const int cus = wt::device_cus();                       // per device: a process that drives a second GPU sizes its grid from THAT one
    static wt::OncePerDevice attr32, attr16;
    const int dev = wt::device_index();
    if (cus <= 0) { wt::set_error("wd_deform_dxoff_f32: cannot read the device properties"); return WT_ERR_HIP; }
    int nwg = 2 * cus;                                       // two workgroups (77 KB of LDS, 6 waves each) per CU, all resident
    if (const char* e = getenv("WD_DXOFF_WGS")) nwg = atoi(e);
    if (nwg > items) nwg = items;
    if (nwg < 1) nwg = 1;
    int far_chunks = 32;                                     // group chunks of the far-sample kernel (workgroups of tiles without such samples exit at once)
    if (const char* e = getenv("WD_FAR_CHUNKS")) far_chunks = atoi(e);
    if (far_chunks < 1) far_chunks = 1;
    if (far_chunks > groups) far_chunks = groups;
    const int naux = 4 * cus;                                // blocks that zero dX and pack the weights, in the launch that builds the tables
    if (cg == 32) {
        if (attr32.needed(dev)) {
            WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deform_dxoff_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)dxoff_smem_bytes<32>()));
            attr32.mark(dev);
        }
        hipLaunchKernelGGL(deform_bwd_tables_kernel<32>, dim3((unsigned)(ntiles + naux)), dim3(256), 0, st, offset, batch, geo, tables, weight, c,
                           packed_weight, dx, doffset);
        hipLaunchKernelGGL(deform_dxoff_kernel<32>, dim3((unsigned)nwg), dim3(384), dxoff_smem_bytes<32>(), st, x, dy, y_act, scale, packed_weight, tables, batch, geo,
                           c, items, dx, doffset);
        hipLaunchKernelGGL(deform_bwd_far_kernel<32>, dim3((unsigned)ntiles, (unsigned)far_chunks), dim3(256), 0, st, x, dy, y_act, scale, packed_weight, tables, batch, geo, c,
                           dx, doffset);
    } else {
        if (attr16.needed(dev)) {
            WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deform_dxoff_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)dxoff_smem_bytes<16>()));
            attr16.mark(dev);
        }
        hipLaunchKernelGGL(deform_bwd_tables_kernel<16>, dim3((unsigned)(ntiles + naux)), dim3(256), 0, st, offset, batch, geo, tables, weight, c,
                           packed_weight, dx, doffset);
        hipLaunchKernelGGL(deform_dxoff_kernel<16>, dim3((unsigned)nwg), dim3(384), dxoff_smem_bytes<16>(), st, x, dy, y_act, scale, packed_weight, tables, batch, geo,
                           c, items, dx, doffset);
        hipLaunchKernelGGL(deform_bwd_far_kernel<16>, dim3((unsigned)ntiles, (unsigned)far_chunks), dim3(256), 0, st, x, dy, y_act, scale, packed_weight, tables, batch, geo, c,
                           dx, doffset);
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // extern "C"
