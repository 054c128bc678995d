// Deformable 3x3 convolution, stride 1, 32 channels per group, no modulation mask: persistent LDS-resident kernel for gfx950
// (detectron2 DeformConv, logs/12442/job.log:412-415,538-541).  Same arithmetic as det_deform.hip's kernels.
//
// What round 2's experiments showed: v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate and, measured on MI355X,
// does not overlap with another wave's VALU work on the same SIMD - a two-team "ping-pong" schedule (one team's MFMA slot under
// the other team's blend slot, 18 workgroup barriers per tile) ran 104 us against 92 us for the same code with 2 barriers per
// tile.  So the kernel is bound by the SUM of matrix and vector instruction time plus whatever stalls are not covered by the other
// wave of the SIMD; this version minimises instructions per MFMA and keeps the waves independent:
//   * persistent: a 512-thread workgroup owns one group (32 channels) and a contiguous range of 8x8-pixel tiles; its two TEAMS
//     of 4 waves work on alternate tiles; the group's weights (all 9 taps, MFMA fragment order, 36 KiB) are loaded into LDS once
//     and never touch the vector-memory path again;
//   * per team two 14x14x32-channel input patches (+ one all-zero pixel; zero-filled outside the image, XOR-swizzled 16-byte
//     slots) filled by global_load_lds (no staging registers): the patch of the team's NEXT tile arrives during the current tile;
//     one sampling table per team, rewritten between the last read of this tile's entries and the first of the next tile's
//     (the only two workgroup barriers per tile);
//   * table entry (pixel, tap) = the 4 corner byte offsets + (lh, lw): a lane's per-tap address math is 4 XOR-adds; a sample
//     outside the image points at the zero pixel (no special case), a sample whose corners leave the patch (|offset| > ~2 px) is
//     flagged and blended from global memory by its own lane only;
//   * the 9 taps are unrolled (no scalar bookkeeping in the loop); every lane blends the MFMA fragment of ITS pixel / ITS 8
//     channels in registers (no im2col slab); the 16 pixels of a wave are rows (m, m + 4) of the tile with even columns on
//     fragment rows {0-3, 12-15} and odd columns on rows {4-11}: with the swizzle an undeformed tap reads the four corners
//     conflict-free (4 LDS cycles per ds_read_b128; simulated and confirmed by SQ_LDS_BANK_CONFLICT);
//   * four independent accumulator chains (a dependent MFMA pair 64 cycles apart measured 39 instead of 32 cycles per MFMA);
//   * D = W x samples (weights are the A operand): a lane ends up with 4 consecutive output channels of its pixel -> the
//     epilogue (FrozenBN affine + ReLU) writes 16 bytes per lane and 16-channel tile.
// LDS: 36 864 (weights) + 4 x 25 216 (patches) + 2 x 9 344 (tables) + 256 (affine) = 156 672 bytes.
#include <type_traits>
#include "common.h"
#include "../../include/waymodet.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

namespace pp {
constexpr int PS = 14;                 // patch side: 8 + 2 (3x3 footprint) + 2 * 2 (halo for the learned offsets)
constexpr int NPIX = PS * PS;          // 196 (+ pixel 196 = zeros)
constexpr int CG = 32;                 // channels per group = per item
constexpr int PATCH_F = (NPIX + 1) * CG;   // floats per patch buffer
constexpr int NE = 64 * 9;             // (pixel, tap) entries per tile
constexpr int BW_F = 9 * 4 * 64 * 4;   // weights of one group: [tap][j][lane][4]
constexpr float FAR = 2.0f;            // lh >= FAR flags a sample whose corners are outside the patch
constexpr int NT = NE + 8;              // table slots: the entries of tile rows 4-7 sit 8 slots (128 B) further, see tab_slot()
constexpr size_t SMEM = (size_t)BW_F * 4 + 4 * (size_t)PATCH_F * 4 + 2 * (size_t)NT * 16 + 256;
}  // namespace pp

// Workgroup barrier without the fence of __syncthreads(): the fence makes every wave wait for its outstanding LDS READS
// (lgkmcnt(0)), which need no ordering; the LDS writes that do (table rewrite, zero fill of edge patches, patch DMA) are
// followed by explicit waits.
#ifdef PP_NO_BARRIER      // experiments: upper bound of what the two workgroup barriers per tile cost (results are wrong)
#define PP_BARRIER() do { asm volatile("" ::: "memory"); } while (0)
#else
#define PP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif
// slot boundary: the "memory" clobber keeps LLVM's IR passes from sinking the LDS loads to their uses behind the MFMA burst, the
// sched_barrier keeps the machine scheduler from moving anything across
#define PP_SLOT() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef PP_DMA_K
#define PP_DMA_K 5
#endif
#ifndef PP_RES_TAPS
#define PP_RES_TAPS 2       // (measured: 0 -> 100.3 us, 2 -> 98.2, 3 -> 98.7, 4 -> 98.2 on one box) taps whose weights stay in registers for the whole kernel (0..9); the others stream from LDS per tile
#endif
#ifdef PP_PROF
__device__ unsigned long long pp_prof[8];      // experiments (reading the cycle counter drains the LDS queue: coarse only)
#define PP_T0 long long _t = __builtin_readcyclecounter(); unsigned long long _acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PP_TICK(i) { const long long _n = __builtin_readcyclecounter(); _acc[i] += (unsigned long long)(_n - _t); _t = _n; }
#define PP_FLUSH if ((threadIdx.x & 63) == 0) { for (int _i = 0; _i < 8; ++_i) atomicAdd(&pp_prof[_i], _acc[_i]); }
#else
#define PP_T0
#define PP_TICK(i)
#define PP_FLUSH
#endif

// fragment row r (0..15) of M tile m -> pixel (y, x) of the 8x8 tile (see header: even columns on rows {0-3, 12-15})
__device__ __forceinline__ int pp_row_pixel(int r, int m) {
    const int blk = r >> 2, i = r & 3;
    const int y = m + 4 * (blk >> 1);
    const int x = 2 * i + ((blk == 1 || blk == 2) ? 1 : 0);
    return y * 8 + x;
}

// Table slot of entry e = pixel * 9 + tap.  A wave reads the entries of tile rows m and m + 4 in one ds_read_b128: their pitch of 9
// slots maps both rows onto the same 8 of the 16 slots of a 256-byte bank row; 8 extra slots for rows 4-7 make it conflict-free.
#ifdef PP_TAB_PAD
__device__ __forceinline__ int pp_tab_slot(int e) { return e + (e >= 32 * 9 ? 8 : 0); }
#else
__device__ __forceinline__ int pp_tab_slot(int e) { return e; }      // measured: the padded table removes the 2-way conflict of the entry read (10 % -> 4 % of LDS cycles) and changes nothing in time
#endif

// 16-byte slot index of quad q (0..7) of patch pixel p
__device__ __forceinline__ int pp_slot(int p, int q) { return (p << 3) + (q ^ ((p >> 1) & 7)); }

// Sampling-table entry of (pixel, tap): the four corner slots (byte offsets inside a patch buffer) + the bilinear fractions.
// t_fy / t_fx = undeformed sample position in patch coordinates (pixel row + kh + 2, pixel column + kw + 2); (ty, tx) = tile.  One
// function for the in-kernel table build and for the per-layer pre-pass (deform_offsets_table_kernel): identical entries.
__device__ __forceinline__ uint4 pp_make_entry(bool pixel_in_image, float t_fy, float t_fx, float2 ov, int ty, int tx, int H, int W) {
    const float py0 = (float)(ty * 8 - 3), px0 = (float)(tx * 8 - 3);
    const float fH = (float)H, fW = (float)W;
    unsigned s0 = pp_slot(pp::NPIX, 0), s1 = s0, s2 = s0, s3 = s0;        // the zero pixel: contributes nothing
    float lh = 0.f, lw = 0.f;
    if (pixel_in_image) {
        const float ry = t_fy + ov.x, rx = t_fx + ov.y;                   // patch coordinates
        const float h_im = ry + py0, w_im = rx + px0;
        if (h_im > -1.f && w_im > -1.f && h_im < fH && w_im < fW) {
            const float fy = floorf(ry), fx = floorf(rx);
            const int hl = (int)fy, wl = (int)fx;
            if ((unsigned)hl <= (unsigned)(pp::PS - 2) && (unsigned)wl <= (unsigned)(pp::PS - 2)) {
                const int u = hl * pp::PS + wl;
                lh = ry - fy; lw = rx - fx;
                s0 = pp_slot(u, 0); s1 = pp_slot(u + 1, 0); s2 = pp_slot(u + pp::PS, 0); s3 = pp_slot(u + pp::PS + 1, 0);
            } else {
                lh = pp::FAR;                                             // the lane recomputes this sample from global memory
            }
        }
    }
    uint4 e;
    e.x = (s0 << 4) | (s1 << 20); e.y = (s2 << 4) | (s3 << 20);           // byte offsets inside the patch buffer (< 2^16)
    e.z = __float_as_uint(lh); e.w = __float_as_uint(lw);
    return e;
}

// Per-layer pre-pass, fused with the offset conv's tap gather (det_misc.hip tap_shift_add_kernel, same arithmetic and order):
// thread = (tile, pixel, tap).  Writes the two offsets of the tap (NHWC, 18 per pixel) and the table entry the persistent kernel
// would otherwise rebuild in EVERY one of its 32 channel-group workgroups (7 us of 93 on res4).  table[tile][pixel * 9 + tap],
// tile = (n * tiles_y + ty) * tiles_x + tx; slots of pixels outside the image hold the zero-pixel entry.
__global__ __launch_bounds__(256) void deform_offsets_table_kernel(const float* __restrict__ partial, int ld,
                                                                  const float* __restrict__ bias, int batch, int H, int W,
                                                                  float* __restrict__ offsets, uint4* __restrict__ table) {
    const int tiles_x = (W + 7) >> 3, tiles_y = (H + 7) >> 3;
    const long total = (long)batch * tiles_y * tiles_x * pp::NE;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int e = (int)(i % pp::NE);
        long tile = i / pp::NE;
        const int tx = (int)(tile % tiles_x);
        tile /= tiles_x;
        const int ty = (int)(tile % tiles_y), tn = (int)(tile / tiles_y);
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
        table[i] = pp_make_entry(in, (float)(yy + kh + 2), (float)(xx + kw + 2), ov, ty, tx, H, W);
    }
}

__global__ __launch_bounds__(512, 2) void deform_conv3x3_pp_kernel(
    const float* __restrict__ x, const float* __restrict__ offset, const float* __restrict__ wfrag,
    const float* __restrict__ scale, const float* __restrict__ bias, int relu,
    int batch, int H, int W, int C, int Cout, int nsplit, float* __restrict__ y, const uint4* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bw = reinterpret_cast<float*>(smem);
    float* patch_all = bw + pp::BW_F;
    uint4* tab_all = reinterpret_cast<uint4*>(patch_all + 4 * pp::PATCH_F);
    float* affine = reinterpret_cast<float*>(tab_all + 2 * pp::NT);          // [scale 32][bias 32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform: scalar branches on team / slot
    const int team = wave >> 2, m = wave & 3, tt = tid & 255;
    const int r16 = lane & 15, kq = lane >> 4;
    const int tiles_x = (W + 7) >> 3, tiles_y = (H + 7) >> 3;
    const int ntiles = batch * tiles_y * tiles_x;
    const int g = blockIdx.x / nsplit, sidx = blockIdx.x - g * nsplit;
    // contiguous tile range of this workgroup
    const int tq = ntiles / nsplit, trm = ntiles - tq * nsplit;
    const int t0 = sidx * tq + (sidx < trm ? sidx : trm);
    const int t1 = t0 + tq + (sidx < trm ? 1 : 0);
    const int n_items = (t1 - t0 + 1) >> 1;                 // per team (the last one of team 1 may be a dummy)
    float* patch_t = patch_all + team * 2 * pp::PATCH_F;
    uint4* tab_t = tab_all + team * pp::NT;
    const int c0 = g * pp::CG;
    const long HW = (long)H * W;
    const char* xb = reinterpret_cast<const char*>(x);
    const char* ob = reinterpret_cast<const char*>(offset);

    // ---- per-thread constants of the patch / table fill (no division inside the loop) ---------------------------
    // The patch goes global -> LDS without passing through registers (global_load_lds: the LDS image of one wave
    // instruction is lane-linear, 64 x 16 bytes; the XOR swizzle is applied on the SOURCE side).  Wave m of a team issues
    // instructions j = 0..6 covering LDS slots (7 m + j) * 64 + lane.
    int p_rel[7], p_rc[7];                  // byte offset from the patch origin pixel, (row << 8 | col) or 0xFFFF
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int sl = (7 * m + j) * 64 + lane;
        const int pxl = sl >> 3, q = (sl & 7) ^ ((pxl >> 1) & 7);
        const int r = pxl / pp::PS, cc = pxl - r * pp::PS;
        p_rel[j] = ((r * W + cc) * C + q * 4) * 4;
        p_rc[j] = (sl < pp::NPIX * 8) ? ((r << 8) | cc) : 0xFFFF;
    }
    int t_rel[3], t_yx[3];                  // byte offset of the entry's (dy, dx) from the tile's first pixel; y << 8 | x | kh << 16 | kw << 20
    float t_fy[3], t_fx[3];                 // undeformed sample position in patch coordinates
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int e = tt + 256 * j;
        const int pxl = e / 9, k = e - 9 * pxl;
        const int kh = k / 3, kw = k - 3 * kh;
        t_rel[j] = (((pxl >> 3) * W + (pxl & 7)) * 18 + 2 * k) * 4;
        t_yx[j] = (e < pp::NE) ? (((pxl >> 3) << 8) | (pxl & 7) | (kh << 16) | (kw << 20)) : 0xFFFF;
        t_fy[j] = (float)((pxl >> 3) + kh + 2);
        t_fx[j] = (float)((pxl & 7) + kw + 2);
    }

    struct TileXY { int tn, ty, tx; };
    auto tile_xy = [&](int it) {
        int t = t0 + team + 2 * it;
        t = t < t1 ? t : t1 - 1;
        TileXY r;
        r.tn = t / (tiles_y * tiles_x);
        const int trem = t - r.tn * tiles_y * tiles_x;
        // tiles are numbered in bands of two tile rows, column by column inside a band: the two teams (even / odd t) work on
        // vertically adjacent tiles at the same time and the next pair is the horizontal neighbour, so 3 of the 4 halo sides are
        // re-read from L2 while still hot (PMC: 79 MB fetched per res4 launch with plain row-major numbering)
        const int band = trem / (2 * tiles_x), rb = trem - band * 2 * tiles_x;
        if (2 * band + 1 < tiles_y) { r.tx = rb >> 1; r.ty = 2 * band + (rb & 1); }
        else { r.tx = rb; r.ty = 2 * band; }
        return r;
    };
    auto tile_valid = [&](int it) { return t0 + team + 2 * it < t1; };

    // offsets of a tile (registers) and its patch (straight into LDS buffer `buf`)
    float2 ov[3];
    uint4 tent[3];
    auto issue_offsets = [&](const TileXY& T) {
        if (table) {                                            // per-layer pre-pass: this tile's entries, ready-made (3 loads per thread)
            const uint4* tb = table + ((size_t)(T.tn * tiles_y + T.ty) * tiles_x + T.tx) * pp::NE;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (t_yx[j] != 0xFFFF) tent[j] = tb[tt + 256 * j];
            return;
        }
        const char* obase = ob + ((long)T.tn * HW + (long)(T.ty * 8) * W + T.tx * 8) * 72;
        const bool full = (T.ty * 8 + 8 <= H) && (T.tx * 8 + 8 <= W);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            ov[j] = make_float2(0.f, 0.f);
            const int yy = (t_yx[j] >> 8) & 255, xx = t_yx[j] & 255;
            if (t_yx[j] != 0xFFFF && (full || (T.ty * 8 + yy < H && T.tx * 8 + xx < W)))
                ov[j] = *reinterpret_cast<const float2*>(obase + t_rel[j]);
        }
    };
    // one LDS-DMA instruction: 64 lanes x 16 bytes from (sbase + voff) to LDS byte offset `dst` + 16 * lane.  Inline asm, not
    // __builtin_amdgcn_global_load_lds: hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of every LDS-DMA that follows another
    // one (7 serialized memory round trips per tile); the wait before the first read of the patch is explicit (vmcnt(0) in G(7)
    // + the barriers)
    auto dma16 = [&](const char* sbase, int voff, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
    };
    auto issue_patch = [&](const TileXY& T, unsigned base) {
        const int py0 = T.ty * 8 - 3, px0 = T.tx * 8 - 3;
        const char* pbase = xb + (((long)T.tn * HW + (long)py0 * W + px0) * C + c0) * 4;
        const bool inner = py0 >= 0 && px0 >= 0 && py0 + pp::PS <= H && px0 + pp::PS <= W;
        if (inner) {
            // interior tile (wave-uniform): scalar base + the per-lane constant offset, no bounds logic
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int s0 = (7 * m + j) * 64;                     // wave-uniform first slot of this instruction
                if (s0 + 64 <= pp::NPIX * 8) dma16(pbase, p_rel[j], base + s0 * 16);
                else if (s0 < pp::NPIX * 8) { if (p_rc[j] != 0xFFFF) dma16(pbase, p_rel[j], base + s0 * 16); }
            }
            return;
        }
        float* pb = reinterpret_cast<float*>(smem + base);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int s0 = (7 * m + j) * 64;
            if (s0 >= pp::NPIX * 8) continue;
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
    // table entries of the next tile are computed into registers one per G slot (the table itself can only be rewritten in
    // G(7): entry 8 of the current tile is read in M(6), entry 0 of the next tile in M(7))
    auto compute_entry = [&](const TileXY& T, int j) {
        if (table) return;                                      // entries of this tile were loaded by issue_offsets()
        const int yy = (t_yx[j] >> 8) & 255, xx = t_yx[j] & 255;
        const bool in = t_yx[j] != 0xFFFF && T.ty * 8 + yy < H && T.tx * 8 + xx < W;
        tent[j] = pp_make_entry(in, t_fy[j], t_fx[j], ov[j], T.ty, T.tx, H, W);
    };
    auto write_table = [&]() {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (t_yx[j] != 0xFFFF) tab_t[pp_tab_slot(tt + 256 * j)] = tent[j];
    };
    auto build_table = [&](const TileXY& T) {
        compute_entry(T, 0); compute_entry(T, 1); compute_entry(T, 2);
        write_table();
    };

    // ---- prologue: group weights + affine + zero pixels -> LDS, first tile of each team --------------------------
    {
        // wfrag: [group][tap][lane][nt*8 + kk] (pack_weight_kernel's fragment copy) -> bw[tap][j][lane][4]
        const float4* src = reinterpret_cast<const float4*>(wfrag + (size_t)g * 9 * 64 * 16);
        for (int e = tid; e < 9 * 64 * 4; e += 512) {
            const int j = e & 3, ln = (e >> 2) & 63, k = e >> 8;
            *reinterpret_cast<float4*>(bw + ((k * 4 + j) * 64 + ln) * 4) = src[e];
        }
        if (tid < 32) {
            affine[tid] = scale ? scale[c0 + tid] : 1.f;
            affine[32 + tid] = bias ? bias[c0 + tid] : 0.f;
        }
        if (tid < 4 * pp::CG) patch_all[(tid >> 5) * pp::PATCH_F + pp::NPIX * pp::CG + (tid & 31)] = 0.f;     // the zero pixels
        const TileXY T0 = tile_xy(0);
        issue_offsets(T0);
        issue_patch(T0, (unsigned)((char*)patch_t - smem));
        build_table(T0);
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the patch has landed
    }
    // accumulators of the current tile: [even k-steps, odd k-steps] x [16-channel tile] - four independent chains (a dependent
    // v_mfma_f32_16x16x4_f32 pair 64 cycles apart measured 39 cycles per MFMA instead of 32); done = finished tile awaiting its epilogue
    f32x4 acc[4], done[2];
    const int my_p = pp_row_pixel(r16, m);
    const int my_y = my_p >> 3, my_x = my_p & 7;

    TileXY prev = {0, 0, 0};
    bool prev_valid = false;
    float4 aff_sc[2], aff_bi[2];                            // FrozenBN scale / shift of this lane's 2 x 4 output channels
    auto epilogue = [&]() {
        if (!prev_valid) return;
        const int ho = prev.ty * 8 + my_y, wo = prev.tx * 8 + my_x;
        if (ho < H && wo < W) {
            float* dst = y + ((long)prev.tn * HW + (long)ho * W + wo) * Cout + c0 + 4 * kq;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                // the affine comes from LDS ONCE (registers afterwards): global loads inside the loop would make the compiler wait for
                // ALL vector memory operations (patch DMA, earlier stores) in front of every store
#ifdef PP_AFFINE_LDS
                const float4 sc = *reinterpret_cast<const float4*>(affine + 16 * nt + 4 * kq);
                const float4 bi = *reinterpret_cast<const float4*>(affine + 32 + 16 * nt + 4 * kq);
#else
                const float4 sc = aff_sc[nt], bi = aff_bi[nt];
#endif
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
    float4 b[2][4];                // weights of a streamed tap k in b[k & 1] (MFMA A operand)
    constexpr int RW = PP_RES_TAPS;
    float4 wres[RW > 0 ? RW : 1][4];   // weights of taps 0 .. RW-1: resident (fewer LDS reads per tap, no register move at tap 0)
    float4 cv[8];                  // the 4 corners x 2 quads of the tap being gathered
    float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;      // weights of the tap whose corners are in cv
    float cur_lh = 0.f, blend_lh = 0.f;                     // lh of the prepared / of the blended tap (>= FAR: global memory)
    unsigned ad0 = 0, ad1 = 0, ad2 = 0, ad3 = 0;            // LDS byte addresses of the prepared tap's corners (first quad)
    uint4 ent = make_uint4(0, 0, 0, 0);                     // table entry read one tap ahead
    const unsigned patch_b0 = (unsigned)((char*)patch_t - smem);          // LDS byte offset of this team's buffer 0
    const unsigned kx = (unsigned)(2 * kq) << 4;            // byte XOR selecting this lane's first quad
    const char* lds = smem;
#ifdef PP_LIN_ENTRY
    const uint4* my_tab = tab_t + r16 * 9;      // experiments: conflict-free entry reads (wrong pixels)
#else
    const uint4* my_tab = tab_t + pp_tab_slot(my_p * 9);
#endif
    const float* my_bw = bw + lane * 4;

    auto read_entry = [&](int k) { ent = my_tab[k]; };                    // LDS
    auto prep = [&](unsigned base) {                                      // VALU: weights + corner addresses from `ent`
        const float lh = __uint_as_float(ent.z), lw = __uint_as_float(ent.w);
        cur_lh = lh;
        const float uh = 1.f - lh, uw = 1.f - lw;
        w00 = uh * uw; w01 = uh * lw; w10 = lh * uw; w11 = lh * lw;
        ad0 = ((ent.x & 0xFFFFu) ^ kx) + base;
        ad1 = ((ent.x >> 16) ^ kx) + base;
        ad2 = ((ent.y & 0xFFFFu) ^ kx) + base;
        ad3 = ((ent.y >> 16) ^ kx) + base;
#ifdef PP_LIN_CORNER
        ad0 = base + lane * 32; ad1 = ad0 + 2048; ad2 = ad0 + 4096; ad3 = ad0 + 6144;      // experiments: lane-linear corner reads
#endif
    };
    auto issue_reads = [&](int k, float4 (&bb)[4], bool want_w) {          // LDS: corners of the prepared tap (+ weights of tap k)
        cv[0] = *reinterpret_cast<const float4*>(lds + ad0); cv[1] = *reinterpret_cast<const float4*>(lds + (ad0 ^ 16));
        cv[2] = *reinterpret_cast<const float4*>(lds + ad1); cv[3] = *reinterpret_cast<const float4*>(lds + (ad1 ^ 16));
        cv[4] = *reinterpret_cast<const float4*>(lds + ad2); cv[5] = *reinterpret_cast<const float4*>(lds + (ad2 ^ 16));
        cv[6] = *reinterpret_cast<const float4*>(lds + ad3); cv[7] = *reinterpret_cast<const float4*>(lds + (ad3 ^ 16));
        if (want_w) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bb[j] = *reinterpret_cast<const float4*>(my_bw + (k * 4 + j) * 256);
        }
    };
    auto blend = [&]() {
#ifdef PP_NO_BLEND
        a[0] = cv[0].x; a[1] = cv[1].y; a[2] = cv[2].x; a[3] = cv[3].y; a[4] = cv[4].x; a[5] = cv[5].x; a[6] = cv[6].x; a[7] = cv[7].x + w00 + w01 + w10 + w11;
        return;
#endif
        a[0] = w00 * cv[0].x + w01 * cv[2].x + w10 * cv[4].x + w11 * cv[6].x;
        a[1] = w00 * cv[0].y + w01 * cv[2].y + w10 * cv[4].y + w11 * cv[6].y;
        a[2] = w00 * cv[0].z + w01 * cv[2].z + w10 * cv[4].z + w11 * cv[6].z;
        a[3] = w00 * cv[0].w + w01 * cv[2].w + w10 * cv[4].w + w11 * cv[6].w;
        a[4] = w00 * cv[1].x + w01 * cv[3].x + w10 * cv[5].x + w11 * cv[7].x;
        a[5] = w00 * cv[1].y + w01 * cv[3].y + w10 * cv[5].y + w11 * cv[7].y;
        a[6] = w00 * cv[1].z + w01 * cv[3].z + w10 * cv[5].z + w11 * cv[7].z;
        a[7] = w00 * cv[1].w + w01 * cv[3].w + w10 * cv[5].w + w11 * cv[7].w;
    };
    // corners outside the patch: the lane blends its sample from global memory (offsets re-read; rare); the result replaces a[]
    // of the flagged lanes.
    TileXY cur = tile_xy(0), nxt = cur;                     // tile being computed / tile being prefetched (wave-uniform)
    auto far_fix = [&](int k) {
        float fx8[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) fx8[kk] = 0.f;
        const bool far = blend_lh >= pp::FAR;
        if (far) {
            const int ho = cur.ty * 8 + my_y, wo = cur.tx * 8 + my_x;
            const int kh = k / 3, kw = k - 3 * kh;
            const float2 o2 = *reinterpret_cast<const float2*>(offset + ((long)cur.tn * HW + (long)ho * W + wo) * 18 + 2 * k);
            // same arithmetic as the table: patch coordinates first, then the image offset
            const float ry = (float)(my_y + kh + 2) + o2.x, rx = (float)(my_x + kw + 2) + o2.y;
            const float fy = floorf(ry), fx = floorf(rx);
            const int ih = (int)fy + cur.ty * 8 - 3, iw = (int)fx + cur.tx * 8 - 3;
            const float lh = ry - fy, lw = rx - fx, uh = 1.f - lh, uw = 1.f - lw;
            const float wq[4] = {uh * uw, uh * lw, lh * uw, lh * lw};
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int yy = ih + (qd >> 1), xx = iw + (qd & 1);
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const float* gb = x + ((long)cur.tn * HW + (long)yy * W + xx) * C + c0 + kq * 8;
                    const float4 v0 = *reinterpret_cast<const float4*>(gb), v1 = *reinterpret_cast<const float4*>(gb + 4);
                    fx8[0] += wq[qd] * v0.x; fx8[1] += wq[qd] * v0.y; fx8[2] += wq[qd] * v0.z; fx8[3] += wq[qd] * v0.w;
                    fx8[4] += wq[qd] * v1.x; fx8[5] += wq[qd] * v1.y; fx8[6] += wq[qd] * v1.z; fx8[7] += wq[qd] * v1.w;
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) a[kk] = far ? fx8[kk] : a[kk];
    };
    auto mfma_tap = [&](const float4 (&bb)[4]) {
#ifdef PP_NO_MFMA
        acc[0][0] += a[0] * bb[0].x + a[7] * bb[3].w; acc[1][1] += a[3] * bb[1].y + a[5] * bb[2].z; acc[2][0] += a[1]; acc[3][0] += a[2];
        return;
#endif
#define PP_MM(kk, wa, wb)                                                                             \
    acc[2 * ((kk) & 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, a[kk], acc[2 * ((kk) & 1)], 0, 0, 0);           \
    acc[2 * ((kk) & 1) + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb, a[kk], acc[2 * ((kk) & 1) + 1], 0, 0, 0);
        PP_MM(0, bb[0].x, bb[2].x) PP_MM(1, bb[0].y, bb[2].y) PP_MM(2, bb[0].z, bb[2].z) PP_MM(3, bb[0].w, bb[2].w)
        PP_MM(4, bb[1].x, bb[3].x) PP_MM(5, bb[1].y, bb[3].y) PP_MM(6, bb[1].z, bb[3].z) PP_MM(7, bb[1].w, bb[3].w)
#undef PP_MM
    };

    // Per wave and tap k of a tile (the 9 taps are unrolled: no scalar bookkeeping in the loop):
    //   G(k): blend from the corners requested in M(k-1); prep = weights / corner addresses of the NEXT tap; one small side job
    //         per k: epilogue of the previous tile (0), next tile's coordinates + offsets (1), patch DMA (2), its table entries
    //         into registers (3-5), table rewrite (7): entry 8 of this tile was read in M(6), entry 0 of the next tile is read in
    //         M(7) - the two barriers of a tile; far-sample fix-ups
    //   M(k): 16 MFMAs; then the LDS requests of the next tap (corners -> cv, weights -> b[(k+1) & 1]) and the table entry of
    //         the tap after it
    unsigned base_cur = patch_b0, base_nxt = patch_b0 + pp::PATCH_F * 4;   // LDS byte offsets of the two patch buffers
    bool have_next = false;
    int it = 0;
    auto tap = [&](auto K) {
        constexpr int k = decltype(K)::value;
        // ---- G(k) ----
        if (k == 0) {                                             // previous tile's accumulators -> epilogue registers
            done[0] = acc[0] + acc[2]; done[1] = acc[1] + acc[3];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (RW == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[0][j] = b[1][j];    // tap 0's weights were requested into b[1] by M(8)
            }
        }
        blend_lh = cur_lh;
        blend();
        prep(k == 8 ? base_nxt : base_cur);                       // the next tap lives in the next tile's patch when k == 8
#ifndef PP_NO_SIDE
#ifndef PP_NO_FAR
        if (__ballot(blend_lh >= pp::FAR)) far_fix(k);
#endif
#ifndef PP_NO_EPI
        if (k == 0 && it > 0) epilogue();
#endif
        // next tile of this team, one small job per G slot (a G slot has ~300 cycles of slack against its partner's MFMA slot)
        if (k == 1) {
            have_next = it + 1 < n_items;
            if (have_next) {
                nxt = tile_xy(it + 1);
#ifndef PP_NO_TAB
                issue_offsets(nxt);
#endif
            }
        }
#ifndef PP_NO_DMA
        if (k == 2 && have_next) issue_patch(nxt, base_nxt);
#endif
#ifndef PP_NO_TAB
        if (k == 3 && have_next) compute_entry(nxt, 0);
        if (k == 4 && have_next) compute_entry(nxt, 1);
        if (k == 5 && have_next) compute_entry(nxt, 2);
#endif
#ifndef PP_NO_TAB
        if (k == 7 && have_next) { write_table(); __builtin_amdgcn_s_waitcnt(0x0070); }      // vmcnt(0): next patch landed; lgkmcnt(0): table / zero-fill stores done
#endif
#endif
        // the slot boundaries are also scheduling barriers: hipcc otherwise moves the next tap's blend (and its LDS waits)
        // up into the MFMA slot, which serialises the wave on LDS latency and defeats the two-team phase structure
        if (k == 7) PP_BARRIER();                                 // table of the next tile written -> readable (entry 0 is read below)
        constexpr int kn = (k + 1) % 9;
#ifndef PP_LATE_READS
        // round 3: the LDS requests of the NEXT tap (corners -> cv, weights -> b[(k+1) & 1]) and the table entry of the tap after it
        // are issued BEFORE this tap's MFMA burst and the slot boundaries are scheduling barriers.  hipcc used to sink them behind
        // the burst and start the next blend right after them: ~200 cycles of LDS latency exposed per tap (SQ_WAIT_ANY 32 % of the
        // wave cycles).  cv is dead once blend() has produced a[], so no extra registers are live across the burst.
        issue_reads(kn, b[(k + 1) & 1], kn >= RW);
        read_entry((kn + 1) % 9);
        PP_SLOT();
#endif
        // ---- M(k) ----
        if constexpr (k < RW) mfma_tap(wres[k]);
        else mfma_tap(b[k & 1]);
#ifndef PP_LATE_READS
        PP_SLOT();
#else
        issue_reads(kn, b[(k + 1) & 1], kn >= RW);
        read_entry((kn + 1) % 9);
#endif
        if (k == 6) PP_BARRIER();                                 // every wave has read entry 8 -> the table may be rewritten in G(7)
    };

    __syncthreads();
#ifndef PP_AFFINE_LDS
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        aff_sc[nt] = *reinterpret_cast<const float4*>(affine + 16 * nt + 4 * kq);
        aff_bi[nt] = *reinterpret_cast<const float4*>(affine + 32 + 16 * nt + 4 * kq);
    }
#endif
    // first tap of the first tile: entry, addresses, corner + weight reads in flight before the loop
    read_entry(0);
    prep(base_cur);
#pragma unroll
    for (int k = 0; k < RW; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) wres[k][j] = *reinterpret_cast<const float4*>(my_bw + (k * 4 + j) * 256);
    issue_reads(0, b[1], RW == 0);
    read_entry(1);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (it = 0; it < n_items; ++it) {
        tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
        tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{}); tap(std::integral_constant<int, 5>{});
        tap(std::integral_constant<int, 6>{}); tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
        prev = cur; prev_valid = tile_valid(it); cur = nxt;
        const unsigned tb = base_cur; base_cur = base_nxt; base_nxt = tb;
    }
    done[0] = acc[0] + acc[2]; done[1] = acc[1] + acc[3];
    epilogue();
    PP_FLUSH
}

}  // namespace

#ifdef PP_PROF
extern "C" int wd_debug_pp_prof(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(pp_prof), 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(pp_prof), z, 64); }
    return 0;
}
#endif

// launcher used by wd_deform_conv3x3_f32 (det_deform.hip)
int wd_deform_pp_launch(const float* x, const float* offset, const float* packed_weight, const float* scale,
                        const float* bias, int relu, int batch, int h, int w, int c, hipStream_t stream, float* y, const void* table) {
    static bool attr_set = false;
    if (!attr_set) {
        WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deform_conv3x3_pp_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp::SMEM));
        attr_set = true;
    }
    const int groups = c / pp::CG;
    const int ntiles = batch * ((h + 7) / 8) * ((w + 7) / 8);
    static int n_cu = 0;                                            // queried once (not inside a stream capture)
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    int nsplit = n_cu / groups;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > (ntiles + 1) / 2) nsplit = (ntiles + 1) / 2;       // at least two tiles (one per team) per workgroup
    if (nsplit < 1) nsplit = 1;
    const float* wfrag = packed_weight + (size_t)c * pp::CG * 9;   // lane-major fragment copy (pack_weight_kernel)
    hipLaunchKernelGGL(deform_conv3x3_pp_kernel, dim3((unsigned)(groups * nsplit)), dim3(512), pp::SMEM, stream, x, offset,
                       wfrag, scale, bias, relu, batch, h, w, c, c, nsplit, y, (const uint4*)table);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

extern "C" size_t wd_deform_table_bytes(int batch, int h, int w) {
    return (size_t)batch * ((h + 7) / 8) * ((w + 7) / 8) * pp::NE * sizeof(uint4);
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
    const long total = (long)batch * ((h + 7) / 8) * ((w + 7) / 8) * pp::NE;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(deform_offsets_table_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, partial, ld, bias, batch,
                       h, w, offsets, (uint4*)table);
    WT_HIP(hipGetLastError());
    return WT_OK;
}
