// Deformable 3x3 convolution, stride 1, 32 or 16 channels per group, no modulation mask: persistent LDS-resident kernel for gfx950
// (detectron2 DeformConv, logs/12442/job.log:412-415,538-541).  Same arithmetic as det_deform.hip's kernels.
//
// Hardware facts this kernel is built on (measured on MI355X: tools/probe/fp32_share_probe.hip, profiles/r03_fp32_share_probe.txt):
//   * v_mfma_f32_16x16x4_f32 occupies a SIMD's matrix pipe for 32 cycles; the waves of a SIMD share that pipe;
//   * VALU work does NOT hide behind another wave's f32 MFMAs: its datapath time (2.6 cycles per v_fma_f32, 4.7 per
//     v_pk_fma_f32) adds to the MFMA time of the SIMD - only instruction issue, LDS traffic and waits overlap;
//   * VALU instructions interleaved between MFMAs cost more than the same instructions in front of the burst;
//   * 13 ds_read_b128 per wave and tap fit under the burst (256 B/clk LDS), if they are issued BEFORE it.
// So: few VALU instructions per MFMA, MFMA bursts kept contiguous, every LDS request of tap k+1 in flight before the burst of
// tap k, as little synchronisation between waves as possible.
//   * persistent: a 512-thread workgroup owns one ITEM = 32 input / output channels (one group of 32 channels, or two groups of
//     16) and a contiguous range of 8x8-pixel tiles; its two TEAMS of 4 waves work on alternate tiles; the item's weights live in
//     MFMA fragment order in registers (taps 0-2) and in LDS (taps 3-8) for the whole kernel;
//   * per team two 14x14x32-channel input patches (+ one all-zero pixel; zero-filled outside the image, XOR-swizzled 16-byte
//     slots) and two sampling tables, all filled by global_load_lds (no staging registers, no LDS stores): patch and table of the
//     team's NEXT tile are requested in tap 1 and have landed long before the ONE workgroup barrier of a tile (tap 7);
//   * the sampling table (entry per (pixel, tap) = the 4 corner byte offsets + (lh, lw)) depends on the offsets only: it is built
//     once per layer (in the offset conv's gather launch, or by deform_table_kernel) and DMA'd per tile; a sample outside the
//     image points at the zero pixel; a sample whose corners leave the patch (|offset| > ~2 px) carries its image coordinates
//     instead: ITS lane fetches the 4 corners from global memory INTO THE SAME REGISTERS the other lanes fill from LDS (one tap
//     ahead, under the MFMA burst), so the blend that follows is the same instruction stream for every lane;
//   * the 9 taps are unrolled; every lane blends the MFMA fragment of ITS pixel / ITS 8 channels in registers (explicit
//     2-vectors -> v_pk_fma_f32; the file is built with -fno-slp-vectorize, the SLP pass would move the blend in front of the
//     burst); the 16 pixels of a wave are rows (m, m + 4) of the tile with even columns on fragment rows {0-3, 12-15} and odd
//     columns on rows {4-11}: with the swizzle an undeformed tap reads the four corners conflict-free;
//   * four independent accumulator chains; D = W x samples (weights are the A operand): a lane ends up with 4 consecutive output
//     channels of its pixel -> the epilogue (FrozenBN affine + ReLU) writes 16 bytes per lane and 16-channel tile.
// LDS (32 channels per group): 24 576 (weights of 6 taps) + 4 x 25 216 (patches) + 4 x 9 216 (tables) = 162 304 bytes.
#include <cstdlib>
#include <mutex>
#include <vector>
#include <type_traits>
#include "common.h"
#include "../../include/waymodet.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

#ifndef PP_SPLIT32
#define PP_SPLIT32 0
#endif
namespace pp {
constexpr int PS = 14;                 // patch side: 8 + 2 (3x3 footprint) + 2 * 2 (halo for the learned offsets)
constexpr int NPIX = PS * PS;          // 196 (+ pixel 196 = zeros)
constexpr int CH = 32;                 // channels per item (one group of 32 or two groups of 16)
constexpr int PATCH_F = (NPIX + 1) * CH;   // floats per patch buffer
constexpr int PATCH_B = PATCH_F * 4;
constexpr int NE = 64 * 9;             // (pixel, tap) entries per tile
constexpr int TAB_B = NE * 16;         // bytes per table buffer
constexpr unsigned FAR_Y = 0xFFFFFFFFu;    // entry.y of a sample whose corners are fetched from global memory
#ifndef PP_RES_TAPS
#define PP_RES_TAPS 3                  // taps whose weights stay in registers for the whole kernel; the others stream from LDS per tile
#endif
constexpr int RW = PP_RES_TAPS;
template <int CG> constexpr int tap_floats() { return (CG == 32 ? 4 : 2) * 64 * 4; }     // weights of one tap of an item
constexpr int PS_WIDE = 21;            // stride 2: 17 x 17 footprint of the undeformed taps of an 8 x 8 output tile + the same 2-pixel halo
template <int CG, int PSIDE = PS, int NT = 2> constexpr size_t smem_bytes() {
    return (size_t)(9 - RW) * tap_floats<CG>() * 4 + 2 * NT * (size_t)((PSIDE * PSIDE + 1) * CH * 4) + 2 * NT * (size_t)TAB_B;
}
}  // namespace pp

// Workgroup barrier without the fence of __syncthreads() (the fence would make every wave wait for its outstanding LDS reads)
#ifdef PP_NO_BARRIER      // experiments: upper bound of what the barrier costs (results are wrong)
#define PP_BARRIER() do { asm volatile("" ::: "memory"); } while (0)
#else
#define PP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif
// slot boundary: the "memory" clobber keeps LLVM's IR passes from sinking the LDS loads to their uses behind the MFMA burst, the
// sched_barrier keeps the machine scheduler from moving anything across
#define PP_SLOT() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// EXPLORATORY (bf16x3 mode, never the benchmarked path): v = hi + lo + O(2^-16 |v|) with hi, lo in bfloat16
__device__ __forceinline__ void pp_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = (__bf16)v[i];
        lo[i] = (__bf16)(v[i] - (float)hi[i]);
    }
}

// fragment row r (0..15) of M tile m -> pixel (y, x) of the 8x8 tile (see header: even columns on rows {0-3, 12-15})
__device__ __forceinline__ int pp_row_pixel(int r, int m) {
    const int blk = r >> 2, i = r & 3;
    const int y = m + 4 * (blk >> 1);
    const int x = 2 * i + ((blk == 1 || blk == 2) ? 1 : 0);
    return y * 8 + x;
}

// 16-byte slot index of quad q (0..7) of patch pixel p
__device__ __forceinline__ int pp_slot(int p, int q) { return (p << 3) + (q ^ ((p >> 1) & 7)); }

// Sampling-table entry of (pixel, tap).  t_fy / t_fx = undeformed sample position in patch coordinates (pixel row + kh + 2, pixel
// column + kw + 2); (ty, tx) = tile.
//   x, y : the four corner slots (byte offsets inside a patch buffer, 16 bits each); a sample outside the image (or a pixel outside
//          the image) points at the zero pixel
//   z, w : the bilinear fractions lh, lw
//   far  : y = FAR_Y, x = (row + 32768) | (column + 32768) << 16 of the sample's upper-left corner in IMAGE coordinates
// Stride S (1 or 2): the patch origin is image pixel (S * 8 ty - org, S * 8 tx - org) with org = 3 for stride 1 (2-pixel halo around the
// 10 x 10 footprint of undeformed taps) and org = -1 for stride 2: there the undeformed footprint is 17 x 17, the 14 x 14 patch holds its
// middle rows / columns 1 .. 14 and the rest (about a third of the samples) takes the far path - still 2 - 3 x faster than the gather kernel.
// round 4: stride 2 gets its own instantiation with 21 x 21 patches (one team, one wave per SIMD: the patches take the LDS of the second team):
// origin 3 like stride 1, no sample of an undeformed tap leaves the patch.
__device__ __forceinline__ int pp_origin(int stride, int ps) { return (stride == 1 || ps == pp::PS_WIDE) ? 3 : -1; }

__device__ __forceinline__ uint4 pp_make_entry(bool pixel_in_image, int yy, int xx, int kh, int kw, float2 ov, int ty, int tx, int H, int W,
                                               int stride, int ps) {
    const int org = pp_origin(stride, ps);
    const float t_fy = (float)(stride * yy + kh - 1 + org), t_fx = (float)(stride * xx + kw - 1 + org);   // undeformed sample, patch coordinates
    const float py0 = (float)(ty * 8 * stride - org), px0 = (float)(tx * 8 * stride - org);
    const float fH = (float)H, fW = (float)W;
    unsigned s0 = pp_slot(ps * ps, 0), s1 = s0, s2 = s0, s3 = s0;         // the zero pixel: contributes nothing
    float lh = 0.f, lw = 0.f;
    uint4 e;
    if (pixel_in_image) {
        const float ry = t_fy + ov.x, rx = t_fx + ov.y;                   // patch coordinates
        const float h_im = ry + py0, w_im = rx + px0;
        if (h_im > -1.f && w_im > -1.f && h_im < fH && w_im < fW) {
            const float fy = floorf(ry), fx = floorf(rx);
            const int hl = (int)fy, wl = (int)fx;
            lh = ry - fy; lw = rx - fx;
            if ((unsigned)hl <= (unsigned)(ps - 2) && (unsigned)wl <= (unsigned)(ps - 2)) {
                const int u = hl * ps + wl;
                s0 = pp_slot(u, 0); s1 = pp_slot(u + 1, 0); s2 = pp_slot(u + ps, 0); s3 = pp_slot(u + ps + 1, 0);
            } else {
                const int ih = hl + ty * 8 * stride - org, iw = wl + tx * 8 * stride - org;     // |.| < 2^15: feature maps are a few thousand pixels at most
                e.x = (unsigned)(ih + 32768) | ((unsigned)(iw + 32768) << 16);
                e.y = pp::FAR_Y;
                e.z = __float_as_uint(lh); e.w = __float_as_uint(lw);
                return e;
            }
        }
    }
    e.x = (s0 << 4) | (s1 << 20); e.y = (s2 << 4) | (s3 << 20);           // byte offsets inside the patch buffer (< 2^16)
    e.z = __float_as_uint(lh); e.w = __float_as_uint(lw);
    return e;
}

// Per-layer pre-pass, fused with the offset conv's tap gather (det_misc.hip tap_shift_add_kernel, same arithmetic and order):
// thread = (tile, pixel, tap).  Writes the two offsets of the tap (NHWC, 18 per pixel) and the table entry.
// table[tile][pixel * 9 + tap], tile = (n * tiles_y + ty) * tiles_x + tx; slots of pixels outside the image hold the zero-pixel entry.
// One workgroup = one tile, one thread = one of its 576 entries.  Behind the table: one word per tile, 1 if any of the tile's samples
// takes the far path (plain stores, no state between launches) - see the far-free loop copy of the 16-channel kernel.
__device__ __forceinline__ void pp_tile_flag(bool far, int tile, int ntiles, uint4* __restrict__ table) {
    const int any = __syncthreads_or(far ? 1 : 0);
    if (threadIdx.x == 0) reinterpret_cast<unsigned*>(table + (size_t)ntiles * pp::NE)[tile] = any ? 1u : 0u;
}

__global__ __launch_bounds__(576) void deform_offsets_table_kernel(const float* __restrict__ partial, int ld,
                                                                  const float* __restrict__ bias, int batch, int H, int W,
                                                                  float* __restrict__ offsets, uint4* __restrict__ table) {
    const int tiles_x = (W + 7) >> 3, tiles_y = (H + 7) >> 3;
    const int ntiles = batch * tiles_y * tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int e = threadIdx.x;
        int tile = t;
        const int tx = tile % tiles_x;
        tile /= tiles_x;
        const int ty = tile % tiles_y, tn = tile / tiles_y;
        const int pxl = e / 9, k = e - 9 * pxl;
        const int yy = pxl >> 3, xx = pxl & 7, kh = k / 3, kw = k - 3 * kh;
        const int oy = ty * 8 + yy, ox = tx * 8 + xx;
        const bool in = oy < H && ox < W;
        float2 ov = make_float2(0.f, 0.f);
        if (in) {
            float a0 = bias ? bias[2 * k] : 0.f, a1 = bias ? bias[2 * k + 1] : 0.f;
            const float* base = partial + (size_t)tn * H * W * ld + 2 * k;
#pragma unroll
            for (int qh = 0; qh < 3; ++qh) {
                const int y = oy + qh - 1;
                if (y < 0 || y >= H) continue;
#pragma unroll
                for (int qw = 0; qw < 3; ++qw) {
                    const int x = ox + qw - 1;
                    if (x < 0 || x >= W) continue;
                    const float2 v = *reinterpret_cast<const float2*>(base + ((size_t)y * W + x) * ld + (qh * 3 + qw) * 18);
                    a0 += v.x; a1 += v.y;
                }
            }
            ov = make_float2(a0, a1);
            *reinterpret_cast<float2*>(offsets + (((size_t)tn * H + oy) * W + ox) * 18 + 2 * k) = ov;
        }
        const uint4 ent = pp_make_entry(in, yy, xx, kh, kw, ov, ty, tx, H, W, 1, pp::PS);
        table[(size_t)t * pp::NE + e] = ent;
        pp_tile_flag(ent.y == pp::FAR_Y, t, ntiles, table);
    }
}

// The same table from an offsets tensor (N, H, W, 18) that already exists (callers without the fused pre-pass).
// H, W: the OUTPUT grid (tiles, offsets); Hin, Win: the sampled image (== H, W at stride 1).
__global__ __launch_bounds__(576) void deform_table_kernel(const float* __restrict__ offsets, int batch, int Hin, int Win, int H, int W,
                                                          int stride, int ps, uint4* __restrict__ table) {
    const int tiles_x = (W + 7) >> 3, tiles_y = (H + 7) >> 3;
    const int ntiles = batch * tiles_y * tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int e = threadIdx.x;
        int tile = t;
        const int tx = tile % tiles_x;
        tile /= tiles_x;
        const int ty = tile % tiles_y, tn = tile / tiles_y;
        const int pxl = e / 9, k = e - 9 * pxl;
        const int yy = pxl >> 3, xx = pxl & 7, kh = k / 3, kw = k - 3 * kh;
        const int oy = ty * 8 + yy, ox = tx * 8 + xx;
        const bool in = oy < H && ox < W;
        float2 ov = make_float2(0.f, 0.f);
        if (in) ov = *reinterpret_cast<const float2*>(offsets + (((size_t)tn * H + oy) * W + ox) * 18 + 2 * k);
        const uint4 ent = pp_make_entry(in, yy, xx, kh, kw, ov, ty, tx, Hin, Win, stride, ps);
        table[(size_t)t * pp::NE + e] = ent;
        pp_tile_flag(ent.y == pp::FAR_Y, t, ntiles, table);
    }
}

// BF3 (exploratory, 32 channels per group only): the implicit GEMM as three bf16 MFMAs per 16-channel tile and tap (hi.hi + hi.lo + lo.hi of a
// 2-way bfloat16 split of weights and samples, f32 accumulation) on the bf16 matrix pipe - 6 x ~17 cycles instead of 16 x 32, and that pipe
// overlaps with VALU.  About 16 mantissa bits per operand: NOT fp32; measured and reported as an `extra` only (tools/bf16x3_experiment.py).
template <int CG, bool BF3 = false, int PSIDE = pp::PS, int NT = 2>
__global__ __launch_bounds__(256 * NT, NT) __attribute__((amdgpu_waves_per_eu(NT, NT))) void deform_conv3x3_pp_kernel(
    const float* __restrict__ x, const float* __restrict__ wfrag, const float* __restrict__ scale, const float* __restrict__ bias,
    int relu, int batch, int H, int W, int Ho, int Wo, int stride, int C, int Cout, int nsplit, float* __restrict__ y,
    const uint4* __restrict__ table) {
    // H, W: the sampled image; Ho, Wo: the output grid (== H, W at stride 1); stride 1 or 2 (pp_origin)
    static_assert(CG == 32 || CG == 16, "32 channels per group, or two groups of 16 per item");
    static_assert(!BF3 || CG == 32, "the bf16x3 experiment exists for 32 channels per group only");
    constexpr int NQ = CG == 32 ? 4 : 2;                     // float4 weight fragments per lane and tap
    constexpr int RW = pp::RW;
    constexpr int PS = PSIDE, NPIX = PS * PS;                // patch side / pixels (+ pixel NPIX = zeros)
    constexpr int PATCH_F = (NPIX + 1) * pp::CH, PATCH_B = PATCH_F * 4;
    constexpr int NPC = (NPIX * 8 + 255) / 256;              // LDS-DMA pieces (64 slots of 16 bytes) per wave and patch
    static_assert((NPIX + 1) * 8 * 16 <= 65536, "corner byte offsets are 16 bits");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bw = reinterpret_cast<float*>(smem);                                            // taps RW..8
    constexpr unsigned PATCH0 = (unsigned)((9 - RW) * pp::tap_floats<CG>() * 4);           // LDS byte offset of the patch buffers
    constexpr unsigned TAB0 = PATCH0 + 2u * NT * PATCH_B;                                   // ... of the table buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform: scalar branches on team / slot
    const int team = wave >> 2, m = wave & 3;
    const int r16 = lane & 15, kq = lane >> 4;
    const int tiles_x = (Wo + 7) >> 3, tiles_y = (Ho + 7) >> 3;
    const int ntiles = batch * tiles_y * tiles_x;
    const int org = pp_origin(stride, PS), tstep = 8 * stride;                                 // patch origin = tile * tstep - org
    const int g = blockIdx.x / nsplit, sidx = blockIdx.x - g * nsplit;                     // g = item (32 channels)
    // contiguous tile range of this workgroup
    const int tq = ntiles / nsplit, trm = ntiles - tq * nsplit;
    const int t0 = sidx * tq + (sidx < trm ? sidx : trm);
    const int t1 = t0 + tq + (sidx < trm ? 1 : 0);
    const int n_items = (t1 - t0 + NT - 1) / NT;             // per team (the last one of team 1 may be a dummy)
    const unsigned patch_b0 = PATCH0 + (unsigned)team * 2u * PATCH_B;                  // this team's patch buffer 0
    const unsigned tab_b0 = TAB0 + (unsigned)team * 2u * pp::TAB_B;                        // ... table buffer 0
    const int c0 = g * pp::CH;
    const long HW = (long)H * W, HWo = (long)Ho * Wo;
    const char* xb = reinterpret_cast<const char*>(x);

    // ---- per-thread constants of the patch fill (no division inside the loop) ------------------------------------
    // The patch goes global -> LDS without passing through registers (global_load_lds: the LDS image of one wave
    // instruction is lane-linear, 64 x 16 bytes; the XOR swizzle is applied on the SOURCE side).  Wave m of a team issues
    // instructions j = 0..6 covering LDS slots (7 m + j) * 64 + lane.
    int p_rel[NPC], p_rc[NPC];                  // byte offset from the patch origin pixel, (row << 8 | col) or 0xFFFF
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int sl = (NPC * m + j) * 64 + lane;
        const int pxl = sl >> 3, q = (sl & 7) ^ ((pxl >> 1) & 7);
        const int r = pxl / PS, cc = pxl - r * PS;
        p_rel[j] = ((r * W + cc) * C + q * 4) * 4;
        p_rc[j] = (sl < NPIX * 8) ? ((r << 8) | cc) : 0xFFFF;
    }

    struct TileXY { int tn, ty, tx; };
    auto tile_of = [&](int t) {                 // work-order tile number -> (image, tile row, tile column)
        TileXY r;
        r.tn = t / (tiles_y * tiles_x);
        const int trem = t - r.tn * tiles_y * tiles_x;
        // tiles are numbered in bands of two tile rows, column by column inside a band: the two teams (even / odd t) work on
        // vertically adjacent tiles at the same time and the next pair is the horizontal neighbour, so 3 of the 4 halo sides are
        // re-read from L2 while still hot (PMC: 79 MB fetched per res4 launch with plain row-major numbering, 66 MB with bands)
        const int band = trem / (2 * tiles_x), rb = trem - band * 2 * tiles_x;
        if (2 * band + 1 < tiles_y) { r.tx = rb >> 1; r.ty = 2 * band + (rb & 1); }
        else { r.tx = rb; r.ty = 2 * band; }
        return r;
    };
    auto tile_xy = [&](int it) {
        int t = t0 + team + NT * it;
        t = t < t1 ? t : t1 - 1;
        return tile_of(t);
    };
    auto tile_valid = [&](int it) { return t0 + team + NT * it < t1; };

    // one LDS-DMA instruction: 64 lanes x 16 bytes from (sbase + voff) to LDS byte offset `dst` + 16 * lane.  Inline asm, not
    // __builtin_amdgcn_global_load_lds: hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of every LDS-DMA that follows another
    // one; the wait before the first read of the data is explicit (vmcnt(0) + the barrier in tap 7)
    auto dma16 = [&](const char* sbase, int voff, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
    };
    auto issue_patch = [&](const TileXY& T, unsigned base) {
        const int py0 = T.ty * tstep - org, px0 = T.tx * tstep - org;
        const char* pbase = xb + (((long)T.tn * HW + (long)py0 * W + px0) * C + c0) * 4;
        const bool inner = py0 >= 0 && px0 >= 0 && py0 + PS <= H && px0 + PS <= W;
        if (inner) {
            // interior tile (wave-uniform): scalar base + the per-lane constant offset, no bounds logic
#pragma unroll
            for (int j = 0; j < NPC; ++j) {
                const int s0 = (NPC * m + j) * 64;                     // wave-uniform first slot of this instruction
                if (s0 + 64 <= NPIX * 8) dma16(pbase, p_rel[j], base + s0 * 16);
                else if (s0 < NPIX * 8) { if (p_rc[j] != 0xFFFF) dma16(pbase, p_rel[j], base + s0 * 16); }
            }
            return;
        }
        float* pb = reinterpret_cast<float*>(smem + base);
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int s0 = (NPC * m + j) * 64;
            if (s0 >= NPIX * 8) continue;
            const int iy = py0 + (p_rc[j] >> 8), ix = px0 + (p_rc[j] & 255);
            const bool in = p_rc[j] != 0xFFFF && iy >= 0 && iy < H && ix >= 0 && ix < W;
            if (in) {
                const char* src = pbase + p_rel[j];
                unsigned keep;
                const unsigned dst = base + s0 * 16;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            } else if (p_rc[j] != 0xFFFF) {
                *reinterpret_cast<float4*>(pb + (s0 + lane) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);   // outside the image
            }
        }
    };
    // the tile's 576 table entries (9 KiB, built once per layer) straight into LDS: 9 wave instructions per team
    const int tab_voff = lane * 16;
    auto issue_table = [&](const TileXY& T, unsigned tbase) {
        const char* tb = reinterpret_cast<const char*>(table) + ((size_t)(T.tn * tiles_y + T.ty) * tiles_x + T.tx) * pp::TAB_B;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = m + 4 * i;                             // wave-uniform
            if (piece < 9) dma16(tb + piece * 1024, tab_voff, tbase + piece * 1024);
        }
    };

    // ---- prologue: item weights (registers + LDS), affine, zero pixels, first tile of each team --------------------
    f32x4 wres[RW > 0 ? RW : 1][NQ];   // weights of taps 0 .. RW-1: resident
    float4 aff_sc[2], aff_bi[2];       // FrozenBN scale / shift of this lane's 2 x 4 output channels
    {
        // wfrag (pack_weight_kernel's fragment copy): CG 32: [group][tap][lane][16]; CG 16: [group][tap][lane][4], an item = groups 2g, 2g+1
        // -> bw[tap - RW][q][lane][4]
        if constexpr (CG == 32 && BF3) {
            // per (tap, lane): the 16 weights [tile][k] -> 4 x 16 bytes [tile 0 hi | tile 0 lo | tile 1 hi | tile 1 lo] (8 bf16 each)
            const float* src = wfrag + (size_t)g * 9 * 64 * 16;
            auto split_tap = [&](int k, int ln, f32x4 (&dst)[4]) {
                float v0[8], v1[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { v0[i] = src[(k * 64 + ln) * 16 + i]; v1[i] = src[(k * 64 + ln) * 16 + 8 + i]; }
                bf16x8 h0, l0, h1, l1;
                pp_split8(v0, h0, l0); pp_split8(v1, h1, l1);
                dst[0] = __builtin_bit_cast(f32x4, h0); dst[1] = __builtin_bit_cast(f32x4, l0);
                dst[2] = __builtin_bit_cast(f32x4, h1); dst[3] = __builtin_bit_cast(f32x4, l1);
            };
            for (int e = tid; e < (9 - RW) * 64; e += 256 * NT) {
                const int ln = e & 63, k = e >> 6;
                f32x4 d[4];
                split_tap(k + RW, ln, d);
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(bw + ((k * 4 + j) * 64 + ln) * 4) = d[j];
            }
#pragma unroll
            for (int k = 0; k < RW; ++k) split_tap(k, lane, wres[k]);
        } else if constexpr (CG == 32) {
            const f32x4* src = reinterpret_cast<const f32x4*>(wfrag + (size_t)g * 9 * 64 * 16);
            for (int e = tid; e < (9 - RW) * 64 * 4; e += 256 * NT) {
                const int j = e & 3, ln = (e >> 2) & 63, k = e >> 8;
                *reinterpret_cast<f32x4*>(bw + ((k * 4 + j) * 64 + ln) * 4) = src[e + RW * 256];
            }
#pragma unroll
            for (int k = 0; k < RW; ++k)
#pragma unroll
                for (int j = 0; j < NQ; ++j) wres[k][j] = src[(k * 64 + lane) * 4 + j];
        } else {
            const f32x4* src = reinterpret_cast<const f32x4*>(wfrag + (size_t)(2 * g) * 9 * 64 * 4);
            for (int e = tid; e < (9 - RW) * 2 * 64; e += 256 * NT) {
                const int ln = e & 63, j = (e >> 6) & 1, k = e >> 7;
                *reinterpret_cast<f32x4*>(bw + ((k * 2 + j) * 64 + ln) * 4) = src[(j * 9 + k + RW) * 64 + ln];
            }
#pragma unroll
            for (int k = 0; k < RW; ++k)
#pragma unroll
                for (int j = 0; j < NQ; ++j) wres[k][j] = src[(j * 9 + k) * 64 + lane];
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = c0 + 16 * nt + 4 * kq;
            aff_sc[nt] = scale ? *reinterpret_cast<const float4*>(scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
            aff_bi[nt] = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < 2 * NT * pp::CH)                                                                       // the zero pixels
            reinterpret_cast<float*>(smem + PATCH0)[(tid >> 5) * PATCH_F + NPIX * pp::CH + (tid & 31)] = 0.f;
        const TileXY T0 = tile_xy(0);
        issue_patch(T0, patch_b0);
        issue_table(T0, tab_b0);
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): patch and table have landed
    }
    // accumulators of the current tile: [even k-steps, odd k-steps] x [16-channel tile] - four independent chains (a dependent
    // v_mfma_f32_16x16x4_f32 pair 64 cycles apart measured 39 cycles per MFMA instead of 32); done = finished tile awaiting its epilogue
    f32x4 acc[4], done[2];
    const int my_p = pp_row_pixel(r16, m);
    const int my_y = my_p >> 3, my_x = my_p & 7;

    TileXY prev = {0, 0, 0};
    bool prev_valid = false;
    auto epilogue = [&]() {
        if (!prev_valid) return;
        const int ho = prev.ty * 8 + my_y, wo = prev.tx * 8 + my_x;
        if (ho < Ho && wo < Wo) {
            float* dst = y + ((long)prev.tn * HWo + (long)ho * Wo + wo) * Cout + c0 + 4 * kq;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float4 sc = aff_sc[nt], bi = aff_bi[nt];
                float4 v;
                v.x = done[nt][0] * sc.x + bi.x; v.y = done[nt][1] * sc.y + bi.y;
                v.z = done[nt][2] * sc.z + bi.z; v.w = done[nt][3] * sc.w + bi.w;
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(dst + 16 * nt) = v;
            }
        }
    };

    // ---- per-tap state ---------------------------------------------------------------------------------------------
    float a[8];                    // blended samples of the current tap (MFMA B operand)
    f32x4 b[2][NQ];                // weights of a streamed tap k in b[k & 1] (MFMA A operand)
    f32x4 cv[8];                   // the 4 corners x 2 quads of the tap being gathered
    float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;      // weights of the tap whose corners are in cv
    bool far_nxt = false;                                   // the prepared tap's sample comes from global memory (this lane)
    unsigned ad0 = 0, ad1 = 0, ad2 = 0, ad3 = 0;            // LDS byte addresses of the prepared tap's corners (first quad)
    u32x4 ent = {0u, 0u, 0u, 0u};                           // table entry read one tap ahead
    // this lane's two channel quads: 32 channels per group: (2 kq, 2 kq + 1) = 8 consecutive input channels;
    // 16 per group: (kq, kq + 4) = channels 4 kq .. 4 kq + 3 of both groups of the item
    constexpr unsigned QX = CG == 32 ? 16u : 64u;           // byte XOR from the first to the second quad
    const unsigned kx = (unsigned)(CG == 32 ? 2 * kq : kq) << 4;
    const int my_ch = (CG == 32 ? 8 * kq : 4 * kq);         // first channel of the first quad inside the item
    const unsigned my_ent = (unsigned)(my_p * 9) * 16u;     // byte offset of this pixel's entries inside a table buffer
    using lds4 = const __attribute__((address_space(3))) f32x4*;
    using ldsu4 = const __attribute__((address_space(3))) u32x4*;

    auto read_entry = [&](unsigned tbase, int k) { ent = *(ldsu4)(uintptr_t)(tbase + my_ent + (unsigned)k * 16u); };      // LDS
    auto prep = [&](unsigned base) {                                      // VALU: weights + corner addresses from `ent`
        const float lh = __uint_as_float(ent.z), lw = __uint_as_float(ent.w);
        const float uh = 1.f - lh, uw = 1.f - lw;
        w00 = uh * uw; w01 = uh * lw; w10 = lh * uw; w11 = lh * lw;
        far_nxt = ent.y == pp::FAR_Y;
        ad0 = ((ent.x & 0xFFFFu) ^ kx) + base;
        ad1 = ((ent.x >> 16) ^ kx) + base;
        ad2 = ((ent.y & 0xFFFFu) ^ kx) + base;
        ad3 = ((ent.y >> 16) ^ kx) + base;
#ifdef PP_LIN_CORNER
        ad0 = base + lane * 32; ad1 = ad0 + 2048; ad2 = ad0 + 4096; ad3 = ad0 + 6144;      // experiments: lane-linear corner reads
#endif
    };
    auto lds_corners = [&]() {
        cv[0] = *(lds4)(uintptr_t)ad0; cv[1] = *(lds4)(uintptr_t)(ad0 ^ QX);
        cv[2] = *(lds4)(uintptr_t)ad1; cv[3] = *(lds4)(uintptr_t)(ad1 ^ QX);
        cv[4] = *(lds4)(uintptr_t)ad2; cv[5] = *(lds4)(uintptr_t)(ad2 ^ QX);
        cv[6] = *(lds4)(uintptr_t)ad3; cv[7] = *(lds4)(uintptr_t)(ad3 ^ QX);
    };
    // requests of the prepared tap: its corners -> cv (LDS; lanes whose sample left the patch: global memory, same registers, a
    // corner outside the image gets weight 0) and the weights of tap k -> bb
    auto issue_reads = [&](auto FARC, int k, f32x4 (&bb)[NQ], bool want_w, int tn) {
        constexpr bool FAR = decltype(FARC)::value;
#ifndef PP_NO_FAR
        if (FAR && __builtin_expect(__ballot(far_nxt) != 0, 0)) {
            if (far_nxt) {
                const int ih = (int)(ent.x & 0xFFFFu) - 32768, iw = (int)(ent.x >> 16) - 32768;
                const bool y0 = ih >= 0 && ih < H, y1 = ih + 1 >= 0 && ih + 1 < H, x0 = iw >= 0 && iw < W, x1 = iw + 1 >= 0 && iw + 1 < W;
                w00 = (y0 && x0) ? w00 : 0.f; w01 = (y0 && x1) ? w01 : 0.f; w10 = (y1 && x0) ? w10 : 0.f; w11 = (y1 && x1) ? w11 : 0.f;
                const int ya = min(max(ih, 0), H - 1), yb = min(max(ih + 1, 0), H - 1);
                const int xa = min(max(iw, 0), W - 1), xc = min(max(iw + 1, 0), W - 1);
                // uniform base (scalar registers) + 32-bit byte offsets per lane: feature maps are far below 4 GB
                const char* gb = reinterpret_cast<const char*>(x + (long)tn * HW * C + c0);
                constexpr unsigned Q2 = CG == 32 ? 16u : 64u;         // bytes from the first to the second quad
                const unsigned ch_b = (unsigned)my_ch * 4u, rowa = (unsigned)(ya * W), rowb = (unsigned)(yb * W), cb = (unsigned)C * 4u;
                const unsigned o00 = (rowa + (unsigned)xa) * cb + ch_b, o01 = (rowa + (unsigned)xc) * cb + ch_b;
                const unsigned o10 = (rowb + (unsigned)xa) * cb + ch_b, o11 = (rowb + (unsigned)xc) * cb + ch_b;
                cv[0] = *reinterpret_cast<const f32x4*>(gb + o00); cv[1] = *reinterpret_cast<const f32x4*>(gb + (o00 + Q2));
                cv[2] = *reinterpret_cast<const f32x4*>(gb + o01); cv[3] = *reinterpret_cast<const f32x4*>(gb + (o01 + Q2));
                cv[4] = *reinterpret_cast<const f32x4*>(gb + o10); cv[5] = *reinterpret_cast<const f32x4*>(gb + (o10 + Q2));
                cv[6] = *reinterpret_cast<const f32x4*>(gb + o11); cv[7] = *reinterpret_cast<const f32x4*>(gb + (o11 + Q2));
            } else {
                lds_corners();
            }
        } else
#endif
        {
            lds_corners();
        }
        if (want_w) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) bb[j] = *(lds4)(uintptr_t)((unsigned)(((k - RW) * NQ + j) * 1024) + (unsigned)lane * 16u);
        }
    };
    auto blend = [&]() {
#ifdef PP_NO_BLEND
        a[0] = cv[0].x; a[1] = cv[1].y; a[2] = cv[2].x; a[3] = cv[3].y; a[4] = cv[4].x; a[5] = cv[5].x; a[6] = cv[6].x; a[7] = cv[7].x + w00 + w01 + w10 + w11;
        return;
#endif
        // two channels per instruction (v_pk_fma_f32 with a broadcast weight), four independent chains advanced together
        const f32x2 W00 = {w00, w00}, W01 = {w01, w01}, W10 = {w10, w10}, W11 = {w11, w11};
        f32x2 r0 = W00 * cv[0].xy, r1 = W00 * cv[0].zw, r2 = W00 * cv[1].xy, r3 = W00 * cv[1].zw;
        r0 += W01 * cv[2].xy; r1 += W01 * cv[2].zw; r2 += W01 * cv[3].xy; r3 += W01 * cv[3].zw;
        r0 += W10 * cv[4].xy; r1 += W10 * cv[4].zw; r2 += W10 * cv[5].xy; r3 += W10 * cv[5].zw;
        r0 += W11 * cv[6].xy; r1 += W11 * cv[6].zw; r2 += W11 * cv[7].xy; r3 += W11 * cv[7].zw;
        a[0] = r0.x; a[1] = r0.y; a[2] = r1.x; a[3] = r1.y; a[4] = r2.x; a[5] = r2.y; a[6] = r3.x; a[7] = r3.y;
    };
    auto mfma_tap = [&](const f32x4 (&bb)[NQ]) {
#ifdef PP_NO_MFMA
        acc[0][0] += a[0] * bb[0].x + a[7] * bb[NQ - 1].w; acc[1][1] += a[3] * bb[1].y + a[5] * bb[0].z; acc[2][0] += a[1]; acc[3][0] += a[2];
        return;
#endif
#define PP_MM(ch, t, wa, av) acc[2 * (ch) + (t)] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, av, acc[2 * (ch) + (t)], 0, 0, 0);
        if constexpr (CG == 32 && BF3) {
            bf16x8 xh, xl;
            pp_split8(a, xh, xl);
            const bf16x8 w0h = __builtin_bit_cast(bf16x8, bb[0]), w0l = __builtin_bit_cast(bf16x8, bb[1]);
            const bf16x8 w1h = __builtin_bit_cast(bf16x8, bb[2]), w1l = __builtin_bit_cast(bf16x8, bb[3]);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0h, xh, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1h, xh, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0h, xl, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1h, xl, acc[3], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0l, xh, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1l, xh, acc[1], 0, 0, 0);
        } else if constexpr (CG == 32) {
            // k-step kk: channels 8 kq + kk of the lane's pixel; output tiles 0 / 1 = channels 0-15 / 16-31 of the group
            PP_MM(0, 0, bb[0].x, a[0]) PP_MM(0, 1, bb[2].x, a[0]) PP_MM(1, 0, bb[0].y, a[1]) PP_MM(1, 1, bb[2].y, a[1])
            PP_MM(0, 0, bb[0].z, a[2]) PP_MM(0, 1, bb[2].z, a[2]) PP_MM(1, 0, bb[0].w, a[3]) PP_MM(1, 1, bb[2].w, a[3])
            PP_MM(0, 0, bb[1].x, a[4]) PP_MM(0, 1, bb[3].x, a[4]) PP_MM(1, 0, bb[1].y, a[5]) PP_MM(1, 1, bb[3].y, a[5])
            PP_MM(0, 0, bb[1].z, a[6]) PP_MM(0, 1, bb[3].z, a[6]) PP_MM(1, 0, bb[1].w, a[7]) PP_MM(1, 1, bb[3].w, a[7])
        } else {
            // two groups of 16: a[0..3] = channels 4 kq + j of group A, a[4..7] of group B; output tile t = group t
            PP_MM(0, 0, bb[0].x, a[0]) PP_MM(0, 1, bb[1].x, a[4]) PP_MM(1, 0, bb[0].y, a[1]) PP_MM(1, 1, bb[1].y, a[5])
            PP_MM(0, 0, bb[0].z, a[2]) PP_MM(0, 1, bb[1].z, a[6]) PP_MM(1, 0, bb[0].w, a[3]) PP_MM(1, 1, bb[1].w, a[7])
        }
#undef PP_MM
    };

    auto pin_corners = [&]() {       // (a plain lambda: clang rejects captured variables as asm operands inside the generic `tap`)
        asm volatile("" : "+v"(cv[0]), "+v"(cv[1]), "+v"(cv[2]), "+v"(cv[3]), "+v"(cv[4]), "+v"(cv[5]), "+v"(cv[6]), "+v"(cv[7]));
    };
    // Per wave and tap k of a tile (the 9 taps are unrolled: no scalar bookkeeping in the loop):
    //   G(k): blend from the corners requested in G(k-1); weights / corner addresses of the NEXT tap; the LDS requests of the next tap
    //         (corners -> cv, weights -> b[(k+1) & 1]) and the table entry of the tap after it; the tile's one barrier (k = 7)
    //   M(k): 16 (8) MFMAs, nothing else
    //   S(k): one side job: epilogue of the previous tile (0), next tile's coordinates + patch / table DMA (1)
    TileXY cur = tile_xy(0), nxt = cur;                     // tile being computed / tile being prefetched (wave-uniform)
    unsigned base_cur = patch_b0, base_nxt = patch_b0 + PATCH_B;       // LDS byte offsets of the two patch buffers
    unsigned tab_cur = tab_b0, tab_nxt = tab_b0 + pp::TAB_B;               // ... of the two table buffers
    bool have_next = false;
    int it = 0;
    auto tap = [&](auto K, auto FARC) {
        constexpr int k = decltype(K)::value;
        constexpr int kn = (k + 1) % 9;
        // ---- G(k) ----
        if (k == 0) {                                             // previous tile's accumulators -> epilogue registers
            done[0] = acc[0] + acc[2]; done[1] = acc[1] + acc[3];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        blend();
        prep(k == 8 ? base_nxt : base_cur);                       // the next tap lives in the next tile's patch when k == 8
        if constexpr (kn < RW) issue_reads(FARC, kn, b[0], false, k == 8 ? nxt.tn : cur.tn);
        else issue_reads(FARC, kn, b[(k + 1) & 1], true, k == 8 ? nxt.tn : cur.tn);
        // side jobs: the compiler waits for ALL vector-memory operations right before the next tap's corner requests (the corner
        // registers may be the target of global loads), so stores and DMA requests are issued right behind this tap's and have a
        // whole tap to complete
#ifndef PP_NO_SIDE
#ifndef PP_NO_EPI
        if (k == 0 && it > 0) epilogue();
#endif
        if (k == 1) {                                             // next tile of this team: patch + table on their way for 6 taps
            have_next = it + 1 < n_items;
            if (have_next) {
                nxt = tile_xy(it + 1);
#ifndef PP_NO_DMA
                issue_patch(nxt, base_nxt);
                issue_table(nxt, tab_nxt);
#endif
            }
        }
#endif
        if (k == 7) {
            // the tile's only barrier.  Before it: every read of THIS tile's patch and table has been issued (tap 8's corners just
            // above, entry 8 in G(6)), so the DMA of tap 1 of the next tile may overwrite them; own DMA pieces (and zero-fill
            // stores of a border patch) have landed.  After it: the next tile's table / patch are readable by every wave.
            __builtin_amdgcn_s_waitcnt(0x0070);                   // vmcnt(0) lgkmcnt(0)
            PP_BARRIER();
        }
        read_entry(k >= 7 ? tab_nxt : tab_cur, (kn + 1) % 9);
        PP_SLOT();
        // ---- M(k) ----
        if constexpr (k < RW) mfma_tap(wres[k]);
        else mfma_tap(b[k & 1]);
        PP_SLOT();
        // the corner registers pass through an (empty) volatile asm behind the burst: the next tap's blend depends on its outputs, so
        // neither LLVM's IR passes nor the DAG scheduler can move that blend (and the wait for its loads) in front of the MFMAs
        pin_corners();
    };

    __syncthreads();
    // first tap of the first tile: entry, addresses, corner + weight reads in flight before the loop
    read_entry(tab_cur, 0);
    prep(base_cur);
    // 16 channels per group: does any tile of this workgroup hold a far sample?  (one word per tile behind the table, written by the
    // pre-pass; every wave reads them itself: no barrier)
    bool wg_far = true;
    if constexpr ((CG == 16 || PP_SPLIT32) && NT == 2) {
        const unsigned* tflag = reinterpret_cast<const unsigned*>(table + (size_t)ntiles * pp::NE);
        wg_far = false;
        for (int t = t0 + lane; t < t1; t += 64) {
            const TileXY T = tile_of(t);
            wg_far |= tflag[(T.tn * tiles_y + T.ty) * tiles_x + T.tx] != 0u;
        }
        wg_far = __ballot(wg_far) != 0;
    }
    if constexpr (RW > 0) issue_reads(std::true_type{}, 0, b[0], false, cur.tn);
    else issue_reads(std::true_type{}, 0, b[0], true, cur.tn);
    read_entry(tab_cur, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto items = [&](auto FARC) {
        for (it = 0; it < n_items; ++it) {
            tap(std::integral_constant<int, 0>{}, FARC); tap(std::integral_constant<int, 1>{}, FARC); tap(std::integral_constant<int, 2>{}, FARC);
            tap(std::integral_constant<int, 3>{}, FARC); tap(std::integral_constant<int, 4>{}, FARC); tap(std::integral_constant<int, 5>{}, FARC);
            tap(std::integral_constant<int, 6>{}, FARC); tap(std::integral_constant<int, 7>{}, FARC); tap(std::integral_constant<int, 8>{}, FARC);
            prev = cur; prev_valid = tile_valid(it); cur = nxt;
            unsigned tb = base_cur; base_cur = base_nxt; base_nxt = tb;
            tb = tab_cur; tab_cur = tab_nxt; tab_nxt = tb;
        }
    };
    // The far path's global loads into the corner registers make hipcc wait for ALL outstanding vector-memory operations (DMA,
    // epilogue stores) before every corner read, even when no lane takes it.  16 channels per group: a second copy of the loop without
    // the far path runs when none of the workgroup's tiles needs it (+8 % at offsets below 1 px).  32 channels per group: the shared
    // register allocation of two copies spills inside the far-capable one (2 px: +9 % time) - one loop there.
    if constexpr ((CG == 16 || PP_SPLIT32) && NT == 2) {
        if (wg_far) items(std::true_type{});
        else items(std::false_type{});
    } else {
        items(std::true_type{});
    }
    done[0] = acc[0] + acc[2]; done[1] = acc[1] + acc[3];
    epilogue();
}

}  // namespace

// scratch table for callers that pass offsets without a pre-built table (stride-2 layers, tests, tools, the training forward): grown on
// demand, ONE BUFFER PER STREAM - launches on different streams may be in flight together (two detector instances of one process), and
// the table is written by one launch and read by the next on the same stream only.  A buffer that has been handed out may be baked
// into a captured hipGraph, so it is never freed or moved: growing allocates a NEW buffer and keeps the old one alive (a few MB per
// distinct size class and stream, once).
static int scratch_table(size_t bytes, hipStream_t stream, void** out) {
    struct Slot { hipStream_t stream; void* buf; size_t cap; };
    static std::mutex mu;
    static std::vector<Slot> slots;
    std::lock_guard<std::mutex> lock(mu);
    Slot* sl = nullptr;
    for (Slot& c : slots) if (c.stream == stream) sl = &c;
    if (!sl) { slots.push_back(Slot{stream, nullptr, 0}); sl = &slots.back(); }
    if (bytes > sl->cap) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &st);
        if (st != hipStreamCaptureStatusNone) {
            wt::set_error("wd_deform_conv3x3: the sampling-table scratch would have to grow, which is not possible inside a stream capture: run "
                          "the shape once eagerly on this stream first, or pass a table (wd_deform_offsets_table_f32)");
            return WT_ERR_INVALID;
        }
        void* fresh = nullptr;
        const size_t want = bytes + bytes / 4;                   // head room: fewer size classes
        WT_HIP(hipMalloc(&fresh, want));
        sl->buf = fresh;                                         // the previous buffer stays allocated (see above)
        sl->cap = want;
    }
    *out = sl->buf;
    return WT_OK;
}

// launcher used by wd_deform_conv3x3_f32 (det_deform.hip); cg = channels per group (32 or 16), stride 1 or 2 (pad 1); `table` (stride 1
// only: built by wd_deform_offsets_table_f32 on the same h x w) or NULL = built here from `offset` into the scratch buffer
int wd_deform_pp_launch(const float* x, const float* offset, const float* packed_weight, const float* scale,
                        const float* bias, int relu, int batch, int h, int w, int c, int cg, int stride, hipStream_t stream, float* y,
                        const void* table) {
    static wt::OncePerDevice attr;
    if (const int dev = wt::device_index(); attr.needed(dev)) {
        WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deform_conv3x3_pp_kernel<32>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp::smem_bytes<32>()));
        WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deform_conv3x3_pp_kernel<16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp::smem_bytes<16>()));
        WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deform_conv3x3_pp_kernel<32, false, pp::PS_WIDE, 1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp::smem_bytes<32, pp::PS_WIDE, 1>()));
        WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deform_conv3x3_pp_kernel<16, false, pp::PS_WIDE, 1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp::smem_bytes<16, pp::PS_WIDE, 1>()));
        attr.mark(dev);
    }
    const int ho = (h + 2 - 3) / stride + 1, wo = (w + 2 - 3) / stride + 1;
    const int items = c / pp::CH;
    const int ntiles = batch * ((ho + 7) / 8) * ((wo + 7) / 8);
    // stride 2, WD_PP_S2=wide: 21 x 21 patches (no sample of an undeformed tap leaves the patch), one team per workgroup - the patches take
    // the LDS of the second team.  Measured on MI355X (tools/deform_s2_bench.py, profiles/r04_deform_stride2.txt): res3 205 vs 229 us at
    // 0.2 px but 261 vs 230 us at 2 px, res4 109 vs 108 / 132 vs 112 us: one wave per SIMD costs what the far path costs -> the default
    // stays the two-team kernel with 14 x 14 patches placed over the middle of the footprint (pp_origin)
    static const bool s2_wide = getenv("WD_PP_S2") && getenv("WD_PP_S2")[0] == 'w';
    const bool wide = stride == 2 && s2_wide;
    const int teams = wide ? 1 : 2;
    if (!table || stride != 1) {
        void* scratch = nullptr;
        WT_TRY(scratch_table(wd_deform_table_bytes(batch, ho, wo), stream, &scratch));
        hipLaunchKernelGGL(deform_table_kernel, dim3((unsigned)(ntiles < 65535 ? ntiles : 65535)), dim3(pp::NE), 0, stream, offset, batch, h, w,
                           ho, wo, stride, wide ? pp::PS_WIDE : pp::PS, (uint4*)scratch);
        WT_HIP(hipGetLastError());
        table = scratch;
    }
    const int n_cu = wt::device_cus() > 0 ? wt::device_cus() : 256;  // cached per device (the first call is not inside a stream capture)
    int nsplit = n_cu / items;
    static const int nsplit_env = getenv("WD_PP_NSPLIT") ? atoi(getenv("WD_PP_NSPLIT")) : 0;    // experiments: fewer workgroups per item
    if (nsplit_env > 0 && nsplit_env < nsplit) nsplit = nsplit_env;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > (ntiles + teams - 1) / teams) nsplit = (ntiles + teams - 1) / teams;       // at least one tile per team
    if (nsplit < 1) nsplit = 1;
    const float* wfrag = packed_weight + (size_t)c * cg * 9;       // lane-major fragment copy (pack_weight_kernel)
    // EXPERIMENT: not fp32 (see the kernel's header), and as an instruction stream (v_mfma_f32_16x16x32_bf16 with a repeated weight operand) the strongest
    // co-residency aggressor measured (profiles/r06_costream_victim_side.txt, burner kind 9): honoured only together with WT_EXPERIMENT=1
    static const bool bf3 = []() {
        const char *e = getenv("WD_DEFORM_BF16X3"), *x = getenv("WT_EXPERIMENT");
        if (!(e && e[0] == '1')) return false;
        if (x && x[0] == '1') return true;
        fprintf(stderr, "libwaymotrack: WD_DEFORM_BF16X3=1 ignored (laboratory switch; set WT_EXPERIMENT=1 to use it)\n");
        return false;
    }();
    const dim3 grid((unsigned)(items * nsplit));
    if (wide && cg == 32)
        hipLaunchKernelGGL((deform_conv3x3_pp_kernel<32, false, pp::PS_WIDE, 1>), grid, dim3(256), (pp::smem_bytes<32, pp::PS_WIDE, 1>()), stream, x,
                           wfrag, scale, bias, relu, batch, h, w, ho, wo, stride, c, c, nsplit, y, (const uint4*)table);
    else if (wide)
        hipLaunchKernelGGL((deform_conv3x3_pp_kernel<16, false, pp::PS_WIDE, 1>), grid, dim3(256), (pp::smem_bytes<16, pp::PS_WIDE, 1>()), stream, x,
                           wfrag, scale, bias, relu, batch, h, w, ho, wo, stride, c, c, nsplit, y, (const uint4*)table);
    else if (cg == 32 && bf3) {
        static bool attr3 = false;
        if (!attr3) {
            WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deform_conv3x3_pp_kernel<32, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp::smem_bytes<32>()));
            attr3 = true;
        }
        hipLaunchKernelGGL((deform_conv3x3_pp_kernel<32, true>), grid, dim3(512), pp::smem_bytes<32>(), stream, x, wfrag, scale, bias, relu,
                           batch, h, w, ho, wo, stride, c, c, nsplit, y, (const uint4*)table);
    } else if (cg == 32)
        hipLaunchKernelGGL(deform_conv3x3_pp_kernel<32>, grid, dim3(512), pp::smem_bytes<32>(), stream, x, wfrag, scale, bias, relu, batch,
                           h, w, ho, wo, stride, c, c, nsplit, y, (const uint4*)table);
    else
        hipLaunchKernelGGL(deform_conv3x3_pp_kernel<16>, grid, dim3(512), pp::smem_bytes<16>(), stream, x, wfrag, scale, bias, relu, batch,
                           h, w, ho, wo, stride, c, c, nsplit, y, (const uint4*)table);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

extern "C" size_t wd_deform_table_bytes(int batch, int h, int w) {
    const size_t ntiles = (size_t)batch * ((h + 7) / 8) * ((w + 7) / 8);
    return ntiles * pp::TAB_B + (ntiles * 4 + 15) / 16 * 16;          // entries + one far flag (u32) per tile
}

/* Offset conv epilogue + sampling table of the persistent deformable kernel in one launch: `partial` = the (pixels, ld >= 162) GEMM
 * output of the 18-channel offset conv (ops.conv3x3_few), stride 1, pad 1.  offsets: (batch, h, w, 18) NHWC; table:
 * wd_deform_table_bytes(batch, h, w) bytes, consumed by wd_deform_conv3x3_tab_f32 on the SAME (batch, h, w). */
extern "C" int wd_deform_offsets_table_f32(const float* partial, int ld, const float* bias, int batch, int h, int w, float* offsets,
                                           void* table, void* stream) {
    WT_TRY(wt::ensure_device());
    if (!partial || !offsets || !table || ld < 162 || (ld & 1) || batch < 1 || h < 1 || w < 1 || ((uintptr_t)partial & 7) ||
        ((uintptr_t)offsets & 7) || ((uintptr_t)table & 15)) {
        wt::set_error("wd_deform_offsets_table_f32: invalid arguments (ld=%d)", ld);
        return WT_ERR_INVALID;
    }
    const long ntiles = (long)batch * ((h + 7) / 8) * ((w + 7) / 8);
    hipLaunchKernelGGL(deform_offsets_table_kernel, dim3((unsigned)(ntiles < 65535 ? ntiles : 65535)), dim3(pp::NE), 0, (hipStream_t)stream,
                       partial, ld, bias, batch, h, w, offsets, (uint4*)table);
    WT_HIP(hipGetLastError());
    return WT_OK;
}
