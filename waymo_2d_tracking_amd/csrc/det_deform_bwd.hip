// Fused deformable-conv backward for the training graph, stride 1 / pad 1, 16 or 32 channels per group, gfx950.
// detectron2's CUDA op (deformable_im2col -> GEMM -> deformable_col2im / col2im_coord) materialises the 9*C*P column slab three times per
// layer; det_backward.hip restates that form (5 passes over the slab, gather-bound through L1: tools/deform_bwd_bench.py).  Here the columns
// never leave the CU:
//   * deform_dw_kernel   : dW[g][o][k][ci] = sum_p dY[p][o] * col[p][k][ci].  A workgroup (3 waves, wave = kernel row kh) owns one group
//     and a slice of the 8x8-pixel tiles; per tile it stages the 14x14xCG input patch (zero-filled outside the image) and the sampling
//     table of the tile in LDS, blends the column fragment of (4 pixels x 16 channels) in registers and feeds it to
//     v_mfma_f32_16x16x4_f32 as the B operand with K = pixels; dY is the A operand straight from global memory (one 8-byte load per
//     k-step, reused by the wave's three taps).  The 9 x CG x CG accumulators stay in registers across the slice and are added to dW with
//     one float atomic per element and workgroup.
// Samples whose corners leave the patch (|offset| > ~2 px) fetch their corners from global memory (per lane, rare).
// Same arithmetic as det_backward.hip up to the summation order (tests/test_gpu_detops.py compares both with the float64 restatement).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/waymodet.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

namespace fb {
constexpr int PS = 14;                 // patch side: 8 + 2 (3x3 footprint) + 2 * 2 (halo for the learned offsets)
constexpr int NPIX = PS * PS;          // 196; pixel 196 = zeros (samples / pixels outside the image)
constexpr int ZERO = NPIX;
constexpr int NE = 64 * 9;             // (pixel, tap) entries per tile
constexpr unsigned FAR = 0xFFFFFFFFu;  // entry.x of a sample whose corners are not all inside the patch
}  // namespace fb

// Sampling entry of (tile pixel (yy, xx), tap (kh, kw)).  Patch origin = image pixel (8 ty - 3, 8 tx - 3).
//   x, y : BYTE offsets of the four corners inside the patch buffer (16 bits each; the zero pixel for a sample outside (-1, H) x (-1, W))
//   z, w : the bilinear fractions lh, lw
// A sample whose corners are not all inside the patch ("far", |offset| > ~2 px) points at the zero pixel as well - the main loops stay
// branch-free - and leaves far = (row + 32768) | (column + 32768) << 16 of its upper-left corner in IMAGE coordinates (otherwise 0) for a
// second pass that only runs for tiles with such samples.
template <int CG>
__device__ __forceinline__ uint4 fb_entry(bool pixel_in_image, int yy, int xx, int kh, int kw, float oy, float ox, int ty, int tx, int H, int W,
                                          unsigned& far) {
    constexpr unsigned PB = CG * 4;        // bytes per patch pixel
    unsigned c0 = fb::ZERO, c1 = fb::ZERO, c2 = fb::ZERO, c3 = fb::ZERO;
    float lh = 0.f, lw = 0.f;
    far = 0;
    if (pixel_in_image) {
        const float ry = (float)(yy + kh + 2) + oy, rx = (float)(xx + kw + 2) + ox;          // patch coordinates
        const float h_im = ry + (float)(ty * 8 - 3), w_im = rx + (float)(tx * 8 - 3);
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
            const float fy = floorf(ry), fx = floorf(rx);
            const int hl = (int)fy, wl = (int)fx;
            lh = ry - fy; lw = rx - fx;
            if ((unsigned)hl <= (unsigned)(fb::PS - 2) && (unsigned)wl <= (unsigned)(fb::PS - 2)) {
                const unsigned u = hl * fb::PS + wl;
                c0 = u; c1 = u + 1; c2 = u + fb::PS; c3 = u + fb::PS + 1;
            } else {
                far = (unsigned)(hl + ty * 8 - 3 + 32768) | ((unsigned)(wl + tx * 8 - 3 + 32768) << 16);
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
__device__ __forceinline__ void fb_stage(const float* __restrict__ x, const float* __restrict__ offset, int tn, int ty, int tx, int H, int W, int C,
                                         int c0, float* __restrict__ xs, uint4* __restrict__ tab, unsigned* __restrict__ farpos,
                                         int* __restrict__ farflag, int tid) {
    constexpr int Q = CG / 4;
    for (int i = tid; i < fb::NPIX * Q; i += NTHR) {
        const int pp = i / Q, q = i - pp * Q;
        const int r = pp / fb::PS;
        const int iy = 8 * ty - 3 + r, ix = 8 * tx - 3 + (pp - r * fb::PS);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = *reinterpret_cast<const f32x4*>(x + ((size_t)(tn * H + iy) * W + ix) * C + c0 + q * 4);
        *reinterpret_cast<f32x4*>(xs + pp * CG + q * 4) = v;
    }
    for (int e = tid; e < fb::NE; e += NTHR) {
        const int p = e / 9, k = e - 9 * p;
        const int yy = p >> 3, xx = p & 7, oy = 8 * ty + yy, ox = 8 * tx + xx;
        const bool in = oy < H && ox < W;
        float2 ov = make_float2(0.f, 0.f);
        if (in) ov = *reinterpret_cast<const float2*>(offset + ((size_t)(tn * H + oy) * W + ox) * 18 + 2 * k);
        const int kh = k / 3;
        unsigned far;
        tab[k * 64 + p] = fb_entry<CG>(in, yy, xx, kh, k - 3 * kh, ov.x, ov.y, ty, tx, H, W, far);
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
                                                        int batch, int H, int W, int C, int slices, float* __restrict__ part) {
    constexpr int MT = CG / 16;                  // 16-wide tiles along o and along ci; a lane holds MT consecutive channels (o = MT i + mt)
    using vec = typename fb_vec<MT>::type;
    __shared__ __attribute__((aligned(16))) float xs[(fb::NPIX + 1) * CG];
    __shared__ uint4 tab[fb::NE];
    __shared__ unsigned farpos[fb::NE];
    __shared__ int farflag[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, j = lane >> 4;
    const int G = C / CG;
    const int g = blockIdx.x % G, slice = blockIdx.x / G;
    const int tiles_x = (W + 7) >> 3, tiles_y = (H + 7) >> 3, ntiles = batch * tiles_y * tiles_x;
    f32x4 acc[3][MT][MT];   // experiments: -DFB_NO_EPILOGUE (no atomics), -DFB_NO_STAGE (stage the first tile only)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < CG) xs[fb::ZERO * CG + tid] = 0.f;
    for (int tile = slice; tile < ntiles; tile += slices) {
        const int tn = tile / (tiles_y * tiles_x), trem = tile - tn * tiles_y * tiles_x;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        __syncthreads();                                   // the previous tile's readers are done
        if (tid < 4) farflag[tid] = 0;
        __syncthreads();
#ifdef FB_NO_STAGE
        if (tile == slice)
#endif
        fb_stage<CG, 192>(x, offset, tn, ty, tx, H, W, C, g * CG, xs, tab, farpos, farflag, tid);
        // A fragments: dY[pixel 4 s + j][o = MT n + mt]  (K = pixels: k-step s covers 4 consecutive pixels of a tile row)
        vec a[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int oy = 8 * ty + (s >> 1), ox = 8 * tx + 4 * (s & 1) + j;
            a[s] = fb_zero<MT>();
            if (oy < H && ox < W) a[s] = *reinterpret_cast<const vec*>(dy + ((size_t)(tn * H + oy) * W + ox) * C + g * CG + MT * n);
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

// dW[g][o][tap][ci] = sum over slices of the partials.  D[i = 4 j + r][n] of tile (mt, nt) = dW[o = MT i + mt][tap][ci = MT n + nt].
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
    dw[(((size_t)g * CG + o) * 9 + 3 * wave + t) * CG + ci] = s;
}

int check_fused(const char* who, int c, int groups, int h, int w) {
    const int cg = groups > 0 ? c / groups : 0;
    if (groups < 1 || c % groups || (cg != 16 && cg != 32) || h < 1 || w < 1) {
        wt::set_error("%s: 16 or 32 channels per group only (C=%d groups=%d)", who, c, groups);
        return WT_ERR_INVALID;
    }
    return WT_OK;
}

// Tile slices per group: every workgroup resident at once (4 per CU at 32 channels per group - 34 KB of LDS, 3 waves each; 6 at 16), the
// tiles spread evenly over the slices.
int fused_dw_slices(int batch, int h, int w, int groups, int cg) {
    const int ntiles = batch * ((h + 7) / 8) * ((w + 7) / 8);
    const int target = (cg == 32 ? 4 : 6) * 256;
    int per_wg = (int)(((long)ntiles * groups + target - 1) / target);
    if (const char* e = getenv("WD_DW_TILES_PER_WG")) per_wg = atoi(e);
    if (per_wg < 1) per_wg = 1;
    return (ntiles + per_wg - 1) / per_wg;
}

}  // namespace

extern "C" {

size_t wd_deform_dw_scratch_floats(int batch, int h, int w, int c, int groups) {
    if (groups < 1 || c % groups) return 0;
    const int cg = c / groups;
    return (size_t)fused_dw_slices(batch, h, w, groups, cg) * groups * 9 * cg * cg;
}

int wd_deform_dw_f32(const float* x, const float* offset, const float* dy, int batch, int h, int w, int c, int groups, float* scratch, float* dw,
                     void* stream) {
    WT_TRY(wt::ensure_device());
    WT_TRY(check_fused("wd_deform_dw_f32", c, groups, h, w));
    const int cg = c / groups;
    hipStream_t st = (hipStream_t)stream;
    const int slices = fused_dw_slices(batch, h, w, groups, cg);
    const int nred = (groups * 9 * cg * cg + 255) / 256;
    if (cg == 32) {
        hipLaunchKernelGGL(deform_dw_kernel<32>, dim3((unsigned)(groups * slices)), dim3(192), 0, st, x, offset, dy, batch, h, w, c, slices, scratch);
        hipLaunchKernelGGL(deform_dw_reduce_kernel<32>, dim3((unsigned)nred), dim3(256), 0, st, scratch, groups, slices, dw);
    } else {
        hipLaunchKernelGGL(deform_dw_kernel<16>, dim3((unsigned)(groups * slices)), dim3(192), 0, st, x, offset, dy, batch, h, w, c, slices, scratch);
        hipLaunchKernelGGL(deform_dw_reduce_kernel<16>, dim3((unsigned)nred), dim3(256), 0, st, scratch, groups, slices, dw);
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // extern "C"
