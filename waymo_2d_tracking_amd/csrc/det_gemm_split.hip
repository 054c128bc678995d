// fp32-equivalent GEMM / implicit-GEMM convolution on the bf16 matrix cores of gfx950 ("split-operand" GEMM).
//
//   out (M, N) = act(A (M, K) . W (N, K)^T + bias [+ residual])            the 1x1 convolutions of the NHWC backbone / FPN
//                                                                          (detectron2 BottleneckBlock conv1 / conv3 / shortcut,
//                                                                          logs/12442/job.log:534-546) and, as an implicit GEMM
//                                                                          over (tap, channel), the dense 3x3 convolutions of
//                                                                          FPN / RPN / box heads (job.log:1126-1160)
//
// gfx950 has no reduced-precision f32 matrix instruction: v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate (157 TFLOP/s), 1/16 of
// the bf16 rate.  Here every f32 operand x is carried EXACTLY as three bfloat16 planes x = hi + mid + lo (successive round-to-nearest
// subtraction: hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid; 8 + 8 + 8 significand bits, the last difference is exact), and
// a.b = sum of the six cross terms with i + j <= 2 (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi) on v_mfma_f32_32x32x16_bf16 with
// f32 accumulation.  Every bf16 x bf16 product is exact in f32; the three dropped terms are <= 2^-26 |a.b| (the f32 product rounding
// of an fmaf chain is 2^-25); the only roundings are the accumulator's, one per MFMA (6 K / 16 of them against K for the fmaf chain
// of the f32 MFMA) - measured error against float64 BELOW the exact-f32 kernel's (tests/test_gpu_gemm_split.py,
// profiles/r05_split_gemm_error.txt).  Six bf16 MFMAs replace sixteen f32 MFMAs' worth of time: 0.375 of the f32 matrix time.
// Not carried: +-inf / NaN operands (x - hi = NaN -> the output element is NaN where the f32 GEMM gives +-inf) and operands below
// 2^-110 (their lo plane underflows bfloat16).
//
// Structure (one workgroup = 8 waves = one (32 MT) x 256 output tile; MT = 5 -> 160 x 256: 9600 x 1024 is 240 tiles on 256 CUs):
//   * W is static: packed once (wd_gemm_split_pack_weight) into MFMA B-fragment order, [N / 32][K / 16][plane][lane] 16-byte
//     entries.  Wave w owns columns 32 w .. 32 w + 31 of the tile and streams ITS fragments straight from L2 into registers
//     (1 KiB contiguous per wave-load, one K step ahead) - no wave shares them, so they never touch LDS.
//   * A (activations, f32 in HBM) is shared by all 8 waves: a K step of 64 is loaded by the workgroup (float4 per thread, 256 B
//     contiguous per row), split in registers (v_cvt_pk_bf16_f32 + shift / and + subtract: 5.5 VALU per element, once per
//     workgroup) and written as three bf16 planes [row][64 k] to LDS (128-byte rows, 16-byte slots XOR-swizzled with
//     (row >> 1) & 7: conflict-free for the ds_read_b128 lane groups of gfx950).  Double-buffered: ONE workgroup barrier per K step.
//   * per 16-deep sub-step a wave reads 3 MT A fragments (ds_read_b128) and issues 6 MT MFMAs: 2 fragment reads per MFMA less than
//     a plain bf16 GEMM, LDS is at a fifth of its bandwidth.
//   * the A row of an output row is a pointer: plain (m * lda), or the NHWC pixel of an output pixel for a (strided) 1x1 or 3x3
//     convolution (K step -> (tap, channel block); rows whose tap leaves the image contribute zeros).
#include "common.h"
#include "../../include/waymodet.h"
#include <cstdlib>
#include <cstring>

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 64;            // K step (floats); one 128-byte row per plane
constexpr int BN = 256;           // 8 waves x 32 columns
constexpr int NTHREADS = 512;

struct SplitArgs {
    const float* a;               // activations
    const uint4* w;               // packed weight planes
    const float* bias;
    const float* residual;
    float* out;
    long lda, ldc;                // row strides (floats) of a (plain mode) and of out / residual
    int M, N, K, relu;
    int tiles_m, tiles_n, xmap;
    int splitk;                   // K slices per output tile (ring kernel); > 1: raw partial tiles go to part[slice][M][N]
    float* part;
    long long* stamps;            // debug: per-workgroup s_memtime stamps (start, main loop, epilogue, end) or nullptr
    // convolution mode: a = NHWC (batch, H, W, C); output pixel grid (Ho, Wo); K = taps * C
    int H, W, C, Ho, Wo, stride, pad, ksize;
};

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32: round to nearest even
}

// (x0, x1) -> packed bf16 pairs of the three planes; hi + mid + lo == x exactly
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = pk_bf16(s0, s1);
}

// ---- epilogue (both kernels) ----
// C/D layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a lane owns ONE column.  The tile goes
// through LDS (free after the main loop's last barrier) so that global memory sees whole 1-KiB output rows: per pass up to 3 row blocks
// (96 rows x 256 floats); every wave writes its 32-column strip, then reads whole rows as float4 per lane and adds bias / residual / ReLU.
template <int MT>
__device__ __forceinline__ void split_epilogue(const SplitArgs& p, unsigned char* smem, const f32x16 (&acc)[MT], int m0, int n0, int wave, int lane) {
    const int rr = lane & 31, rg = lane >> 5;
    float* ct = reinterpret_cast<float*>(smem);
    const int ncols = p.N - n0 < BN ? p.N - n0 : BN;            // valid columns of this tile (multiple of 32)
    const int c4 = 4 * lane;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && c4 < ncols) bv = *reinterpret_cast<const float4*>(p.bias + n0 + c4);
    constexpr int PASS = 3;                                     // row blocks per pass
    constexpr int RPW = 32 * PASS / 8;                          // rows per wave and pass
#pragma unroll
    for (int i0 = 0; i0 < MT; i0 += PASS) {
        const int nrows = 32 * ((MT - i0) < PASS ? (MT - i0) : PASS);
        // the residual rows of this pass are requested FIRST, all at once: they are the only reads of the epilogue that come from beyond L2
        // (~2 us away), and twelve 1-KiB row loads per wave in flight are what it takes to pull them at more than ~10 B/clk per CU
        float4 rv[RPW];
        if (p.residual) {
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int r = wave + 8 * j, row = m0 + 32 * i0 + r;
                rv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < nrows && row < p.M && c4 < ncols) rv[j] = *reinterpret_cast<const float4*>(p.residual + (size_t)row * p.ldc + n0 + c4);
            }
        }
        if (i0 > 0) __builtin_amdgcn_s_barrier();              // the previous pass has been read
#pragma unroll
        for (int i = i0; i < i0 + PASS && i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                ct[((i - i0) * 32 + (e & 3) + 8 * (e >> 2) + 4 * rg) * BN + 32 * wave + rr] = acc[i][e];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int r = wave + 8 * j, row = m0 + 32 * i0 + r;
            if (r < nrows && row < p.M && c4 < ncols) {
                float4 v = *reinterpret_cast<const float4*>(ct + r * BN + c4);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (p.residual) { v.x += rv[j].x; v.y += rv[j].y; v.z += rv[j].z; v.w += rv[j].w; }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(p.out + (size_t)row * p.ldc + n0 + c4) = v;
            }
        }
    }
}

// MODE 0: plain row-major A.  MODE 1: NHWC convolution source (ksize 1 or 3, any stride / pad).
template <int MT, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_split_k64_kernel(const SplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 32 * MT;
    constexpr int PLANE = BM * 128;              // bytes of one bf16 plane of a K step
    constexpr int BUF = 3 * PLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // workgroup b runs on XCD b % 8: every XCD takes a contiguous run of tiles (N fastest), so the tiles_n workgroups that share an A row
    // block meet in one L2
    const int total = p.tiles_m * p.tiles_n;
    int id;
    {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = total >> 3, r = total & 7;
        if (j >= q + (x < r ? 1 : 0)) return;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    int tm, tn;
    if (p.xmap == 0) {                              // N fastest: the tiles_n workgroups sharing an A row block are neighbours on one XCD
        tm = id / p.tiles_n; tn = id - tm * p.tiles_n;
    } else {                                        // M fastest: an XCD walks down ONE column block of W (its L2 holds that block's planes)
        tn = id / p.tiles_m; tm = id - tn * p.tiles_m;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = p.K / BK;

    // ---- A staging: thread -> (row srow + 32 i, float4 sk4 of the 64-float K step) ----
    const int srow = tid >> 4, sk4 = tid & 15;
    int aoff[MT];                                 // element offset of the thread's float4 at K step 0 (MODE 1: at tap (0, 0), channel 0)
    unsigned vmask[MT];                           // MODE 1: bit t = tap t of this row lies inside the image
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = m0 + srow + 32 * i;
        m = m < p.M ? m : p.M - 1;
        if (MODE == 0) {
            aoff[i] = (int)(m * p.lda) + 4 * sk4;
            vmask[i] = 1u;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
            const int y0 = yo * p.stride - p.pad, x0 = xo * p.stride - p.pad;
            aoff[i] = ((b * p.H + y0) * p.W + x0) * p.C + 4 * sk4;
            unsigned vm = 0;
            for (int t = 0; t < p.ksize * p.ksize; ++t) {
                const int yy = y0 + t / p.ksize, xx = x0 + t % p.ksize;
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) vm |= 1u << t;
            }
            vmask[i] = vm;
        }
    }
    const int kc = (MODE == 1) ? p.C / BK : 1;    // K steps per tap
    float4 araw[MT];
    auto a_load_row = [&](int kt, int i) {          // row block i of K step kt (clamped) -> araw[i]
        kt = kt < nk ? kt : nk - 1;
        if (MODE == 0) {
            araw[i] = *reinterpret_cast<const float4*>(p.a + (aoff[i] + kt * BK));
        } else {
            const int tap = kt / kc, cb = kt - tap * kc;
            const int dy = tap / p.ksize, dx = tap - dy * p.ksize;
            const int delta = (dy * p.W + dx) * p.C + cb * BK;
            const bool ok = ((vmask[i] >> tap) & 1u) != 0;
            const int off = ok ? aoff[i] + delta : 4 * sk4;               // always a valid address; zeroed below
            const float4 v = *reinterpret_cast<const float4*>(p.a + off);
            araw[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int wofs = srow * 128 + ((((sk4 >> 1) ^ ((srow >> 1) & 7)) << 4) | ((sk4 & 1) << 3));
    auto a_store_row = [&](int buf, int i) {       // split row block i of araw and write its three planes
        unsigned char* base = smem + buf * BUF + wofs + i * 4096;
        unsigned h0, m0_, l0, h1, m1, l1;
        split_pair(araw[i].x, araw[i].y, h0, m0_, l0);
        split_pair(araw[i].z, araw[i].w, h1, m1, l1);
        *reinterpret_cast<uint2*>(base) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(base + PLANE) = make_uint2(m0_, m1);
        *reinterpret_cast<uint2*>(base + 2 * PLANE) = make_uint2(l0, l1);
    };

    // ---- W fragments: this wave's 32 columns, [K / 16][3 planes][64 lanes] uint4; a ring of two sub-steps in registers ----
    const int nt32 = (n0 >> 5) + wave;
    const bool active = nt32 * 32 < p.N;          // waves past N (N % 256 != 0) compute on tile 0 and store nothing
    const uint4* wbase = p.w + (size_t)(active ? nt32 : 0) * (size_t)(p.K / 16) * 192 + lane;
    const int nsub = nk * 4;
    bf16x8 wf[2][3];
    auto w_load = [&](int sub, int slot) {         // sub = global sub-step index (K / 16 of them), clamped at the end
        sub = sub < nsub ? sub : nsub - 1;
        const uint4* q = wbase + (size_t)sub * 192;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[slot][pl] = __builtin_bit_cast(bf16x8, q[pl * 64]);
    };

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // fragment read offsets: lane (row rr = lane & 31, k half rg = lane >> 5), sub-step s -> slot (2 s + rg) ^ ((rr >> 1) & 7)
    const int rr = lane & 31, rg = lane >> 5;
    int rofs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) rofs[s] = rr * 128 + ((((2 * s + rg) ^ ((rr >> 1) & 7))) << 4);

    // ---- prologue ----
#pragma unroll
    for (int i = 0; i < MT; ++i) a_load_row(0, i);
    w_load(0, 0);
    w_load(1, 1);
#pragma unroll
    for (int i = 0; i < MT; ++i) { a_store_row(0, i); a_load_row(1, i); }
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this wave's LDS writes are done
    __builtin_amdgcn_s_barrier();

    // Steady state, one K step (4 sub-steps of 6 MT MFMAs) per iteration, ONE barrier.  The instruction stream is laid out by hand in
    // slots of MT MFMAs (sched_barrier between slots): the three planes of a sub-step's A fragments are consumed lo -> mid -> hi, and
    // the reads of the NEXT sub-step's plane are issued into the same registers right behind the last MFMA that used it (>= 3 slots of
    // MFMAs ahead of their first use); the W ring is refilled two sub-steps ahead; the split of the next K step's A rows (VALU) and its
    // LDS writes sit in the read-free slots of sub-steps 2 and 3, each row block's global load for the K step after that right behind its split
    // (a full K step of MFMAs ahead of its use).
    bf16x8 af[MT][3];
using F4_ = __attribute__((ext_vector_type(4))) float;
#define SB __builtin_amdgcn_sched_barrier(0)
#define MF(pa, pb, slot)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], wf[slot][pb], acc[i], 0, 0, 0);
#define RD(s, pl)                                                                                             \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
        af[i][pl] = *reinterpret_cast<const bf16x8*>(rbase + (pl) * PLANE + i * 4096 + rofs[s]);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const unsigned char* rbase = smem + cur * BUF;
        RD(0, 2) RD(0, 1) RD(0, 0)
        SB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int slot = s & 1;
            MF(2, 0, slot)
            if (s < 3) { RD(s + 1, 2) }
            SB;
            MF(1, 1, slot)
            if (s >= 2 && 3 * (s - 2) + 0 < MT) { a_store_row(cur ^ 1, 3 * (s - 2) + 0); a_load_row(kt + 2, 3 * (s - 2) + 0); }
            SB;
            MF(1, 0, slot)
            if (s < 3) { RD(s + 1, 1) }
            SB;
            MF(0, 2, slot)
            if (s >= 2 && 3 * (s - 2) + 1 < MT) { a_store_row(cur ^ 1, 3 * (s - 2) + 1); a_load_row(kt + 2, 3 * (s - 2) + 1); }
            SB;
            MF(0, 1, slot)
            if (s >= 2 && 3 * (s - 2) + 2 < MT) { a_store_row(cur ^ 1, 3 * (s - 2) + 2); a_load_row(kt + 2, 3 * (s - 2) + 2); }
            SB;
            MF(0, 0, slot)
            if (s < 3) { RD(s + 1, 0) }
            w_load(kt * 4 + s + 2, slot);
            SB;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    }
#undef SB
#undef MF
#undef RD

    split_epilogue<MT>(p, smem, acc, m0, n0, wave, lane);
}

// ---- round 5, second structure: K steps of 32 through a ring of THREE LDS buffers -------------------------------------------------
// Same tile, same fragment order, same split; what changes is the hand-off.  The K-64 kernel above double-buffers: the buffer a K step
// reads is written during the step before, so the first fragments of a K step can only be requested behind the barrier and every wave's
// matrix pipe drains once per K step (counters: 25 % of the main loop's wave cycles).  Here step j reads buffer j % 3 while the split of
// step j + 2 is written into buffer (j + 2) % 3 - which nobody has touched since the barrier of step j - 1 - so buffer (j + 1) % 3 is
// complete and visible for the whole of step j: the last slots of a step prefetch the next step's fragments ACROSS the barrier, whose
// wait is a counted lgkmcnt (the LDS writes are older than the MT outstanding prefetch reads).  One barrier per 2 x 6 MT MFMAs, no drain.
// A rows are staged as float2 per thread (160 rows x 32 floats / 512 threads = 5 float2), three ds_write_b32 per row block.
// A rows fetched two K steps ahead of their split (a second register set; MT = 6 still fits 256 VGPRs without scratch).  Measured on MI355X,
// graph replay, same box (profiles/r05_split_adeep.txt): res4 1x1 116.0 -> 109.3 us, box-head 3x3 347.1 -> 340.6 us, FPN p2 3x3 998 -> 965 us, res3 equal.
// -DWD_SPLIT_ADEEP=0 builds the one-step-ahead loop it replaced.
#ifndef WD_SPLIT_ADEEP
#define WD_SPLIT_ADEEP 1
#endif
#ifndef WD_ABL
#define WD_ABL 0
#endif
#ifndef WD_SPLIT_PRIO
#define WD_SPLIT_PRIO 0
#endif
#ifndef WD_SPLIT_CURSOR
#define WD_SPLIT_CURSOR 1       // conv mode: (tap, channel block) of the fetched K step kept as a cursor (0: divided out of kt per row block)
#endif
#ifndef WD_SPLIT_RESPF
#define WD_SPLIT_RESPF 0        // experiment, off: K steps between a residual prefetch into L2 and the epilogue (measured with 3 and 6: 2 - 4 % SLOWER)
#endif
template <int MT, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_split_kernel(const SplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 32 * MT;
    constexpr int PLANE = BM * 64;               // bytes of one bf16 plane of a 32-deep K step
    constexpr int BUF = 3 * PLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = p.tiles_m * p.tiles_n * p.splitk;
    int id;
    {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = total >> 3, r = total & 7;
        if (j >= q + (x < r ? 1 : 0)) return;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    long long t_start = 0, r_start = 0;
    if (p.stamps) { t_start = __builtin_amdgcn_s_memtime(); r_start = __builtin_amdgcn_s_memrealtime(); }
    // split-K: the slices of one tile are neighbours (same XCD); slice kz covers K steps [k0, k0 + nk)
    const int kz = id % p.splitk;
    id /= p.splitk;
    int tm, tn;
    if (p.xmap == 0) { tm = id / p.tiles_n; tn = id - tm * p.tiles_n; } else { tn = id / p.tiles_m; tm = id - tn * p.tiles_m; }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk_all = p.K / 32;
    const int per = (nk_all + p.splitk - 1) / p.splitk;
    const int k0 = kz * per;
    const int nk = (k0 + per <= nk_all) ? per : nk_all - k0;

    // ---- A staging: thread -> (row srow + 32 i, float2 sk2 of the 32-float K step) ----
    const int srow = tid >> 4, sk2 = tid & 15;
    unsigned aoff[MT];                            // 32-bit element offsets against the scalar base pointer (MODE 1: may wrap below 0 for padded taps)
    unsigned vmask[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = m0 + srow + 32 * i;
        m = m < p.M ? m : p.M - 1;
        if (MODE == 0) {
            aoff[i] = (unsigned)(m * p.lda) + 2u * sk2;
            vmask[i] = 1u;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
            const int y0 = yo * p.stride - p.pad, x0 = xo * p.stride - p.pad;
            aoff[i] = (unsigned)(((b * p.H + y0) * p.W + x0) * p.C + 2 * sk2);
            unsigned vm = 0;
            for (int t = 0; t < p.ksize * p.ksize; ++t) {
                const int yy = y0 + t / p.ksize, xx = x0 + t % p.ksize;
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) vm |= 1u << t;
            }
            vmask[i] = vm;
        }
    }
    const int kc = (MODE == 1) ? p.C / 32 : 1;    // K steps per tap
    float2 araw[MT];
    auto a_fetch = [&](int kt, int i) -> float2 {   // row block i of K step kt (clamped)
        kt = k0 + (kt < nk ? kt : nk - 1);
#if WD_ABL & 128
        kt = k0;
#endif
        if (MODE == 0) {
            return *reinterpret_cast<const float2*>(p.a + (size_t)(aoff[i] + (unsigned)(kt * 32)));
        } else {
            const int tap = kt / kc, cb = kt - tap * kc;
            const int dy = tap / p.ksize, dx = tap - dy * p.ksize;
            const unsigned delta = (unsigned)((dy * p.W + dx) * p.C + cb * 32);
            const bool ok = ((vmask[i] >> tap) & 1u) != 0;
            const unsigned off = ok ? aoff[i] + delta : 2u * sk2;         // always a valid address; zeroed below
            const float2 v = *reinterpret_cast<const float2*>(p.a + (size_t)off);
            return ok ? v : make_float2(0.f, 0.f);
        }
    };
    auto a_load_row = [&](int kt, int i) { araw[i] = a_fetch(kt, i); };
    // MODE 1, main loop: the (tap, channel block) of the K step being fetched is a CURSOR advanced once per iteration - a_fetch divides it out of kt for
    // every row block (217 scalar instructions per wave and K step in the box-head convolution, as many as its vector instructions: profiles/
    // r05_split_pmc_conv.txt)
    int f_idx = 0, f_tap = 0, f_cb = 0;
    unsigned f_delta = 0;
    auto f_place = [&]() {
        const int dy = p.ksize == 3 ? (f_tap * 11) >> 5 : 0, dx = f_tap - dy * p.ksize;      // tap / 3 for tap < 9
        f_delta = (unsigned)((dy * p.W + dx) * p.C + f_cb * 32);
    };
    auto f_set = [&](int kt) {
        f_idx = kt < nk ? kt : nk - 1;
        const int ks = k0 + f_idx;
        f_tap = ks / kc; f_cb = ks - f_tap * kc;
        f_place();
    };
    auto f_advance = [&](int kt) {                  // -> K step min(kt, nk - 1), one ahead of the current one at most
        if (kt < nk && kt > f_idx) {
            ++f_idx;
            if (++f_cb == kc) { f_cb = 0; ++f_tap; }
            f_place();
        }
    };
    auto a_fetch_cur = [&](int i) -> float2 {
        const bool ok = ((vmask[i] >> f_tap) & 1u) != 0;
        const unsigned off = ok ? aoff[i] + f_delta : 2u * sk2;
        const float2 v = *reinterpret_cast<const float2*>(p.a + (size_t)off);
        return ok ? v : make_float2(0.f, 0.f);
    };
    const int wofs = srow * 64 + ((((sk2 >> 2) ^ ((srow >> 2) & 3))) << 4) + ((sk2 & 3) << 2);
    auto a_store_val = [&](int bufoff, int i, float2 v) {     // split a row block's float2 and write its three planes (one bf16 pair each)
        unsigned char* base = smem + bufoff + wofs + i * 2048;
        unsigned h, m, l;
        split_pair(v.x, v.y, h, m, l);
        *reinterpret_cast<unsigned*>(base) = h;
        *reinterpret_cast<unsigned*>(base + PLANE) = m;
        *reinterpret_cast<unsigned*>(base + 2 * PLANE) = l;
    };
    auto a_store_row = [&](int bufoff, int i) { a_store_val(bufoff, i, araw[i]); };

    // ---- W fragments: this wave's 32 columns, [K / 16][3 planes][64 lanes] uint4; one K step (two sub-steps) in registers ----
    const int nt32 = (n0 >> 5) + wave;
    const bool active = nt32 * 32 < p.N;          // waves past N (N % 256 != 0) compute on tile 0 and store nothing
    const uint4* wbase = p.w + (size_t)(active ? nt32 : 0) * (size_t)(p.K / 16) * 192 + lane;
    const int nsub = nk * 2;
    bf16x8 wf[2][3];
    auto w_load = [&](int sub, int slot) {
        sub = 2 * k0 + (sub < nsub ? sub : nsub - 1);
#if WD_ABL & 64
        sub = 2 * k0 + (sub & 1);
#endif
        const uint4* q = wbase + (size_t)sub * 192;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[slot][pl] = __builtin_bit_cast(bf16x8, q[pl * 64]);
    };

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // fragment read offsets: lane (row rr, k half rg), sub-step s -> 16-byte slot (2 s + rg) ^ ((rr >> 2) & 3) of the 64-byte row
    const int rr = lane & 31, rg = lane >> 5;
    const int rofs0 = rr * 64 + ((((0 + rg) ^ ((rr >> 2) & 3))) << 4);
    const int rofs1 = rr * 64 + ((((2 + rg) ^ ((rr >> 2) & 3))) << 4);

    // ---- prologue: K steps 0, 1, 2 requested at once (one HBM round trip, not three); 0 and 1 go to buffers 0 and 1, 2 stays in registers ----
    {
        float2 p0[MT], p1[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) p0[i] = a_fetch(0, i);
        w_load(0, 0);
        w_load(1, 1);
#pragma unroll
        for (int i = 0; i < MT; ++i) p1[i] = a_fetch(1, i);
#pragma unroll
        for (int i = 0; i < MT; ++i) a_load_row(2, i);
#pragma unroll
        for (int i = 0; i < MT; ++i) a_store_val(0, i, p0[i]);
#pragma unroll
        for (int i = 0; i < MT; ++i) a_store_val(BUF, i, p1[i]);
    }
#if WD_SPLIT_ADEEP
    float2 aahead[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) aahead[i] = a_fetch(3, i);
#endif
    long long t_main = 0;
    if (p.stamps) t_main = __builtin_amdgcn_s_memtime();      // (an SMEM op: kept in front of the lgkmcnt(0) below, see the loop's counted waits)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();

    bf16x8 af[MT][3];
#define SB __builtin_amdgcn_sched_barrier(0)
    // compile-time ablations for tools/split_ablation.sh (timing only, results wrong): WD_ABL bit 0 no A global loads, 1 no split / LDS
    // writes, 2 no W loads, 3 no barrier, 4 no MFMAs, 5 no fragment reads, 6 W loads always from K step 0 (cache hits), 7 A loads always from K step 0
#if WD_ABL & 256      /* diagnostics: the MFMA's issue time as s_nop (no matrix-pipe work) */
#define MF(pa, pb, slot)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" :: "v"(af[i][pa]), "v"(wf[slot][pb])); }
#elif WD_ABL & 512    /* diagnostics: an f32 MFMA (32x32x2, 64 cycles) in place of every bf16 MFMA */
#define MF(pa, pb, slot)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(F4_, af[i][pa])[0], __builtin_bit_cast(F4_, wf[slot][pb])[0], acc[i], 0, 0, 0);
#elif WD_ABL & 1024   /* diagnostics: the bf16 MFMA on constant-zero operands (same pipe occupancy, no data toggling) */
#define MF(pa, pb, slot)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                        \
        bf16x8 z_ = {}; asm volatile("" : "+v"(z_) : "v"(af[i][pa]), "v"(wf[slot][pb]));                      \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z_, z_, acc[i], 0, 0, 0); }
#elif WD_ABL & 16
#define MF(pa, pb, slot)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) asm volatile("" :: "v"(af[i][pa]), "v"(wf[slot][pb]));
#else
#define MF(pa, pb, slot)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], wf[slot][pb], acc[i], 0, 0, 0);
#endif
#if WD_ABL & 32
#define RD(addr, pl)
#else
#define RD(addr, pl)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
        af[i][pl] = *reinterpret_cast<const bf16x8*>(smem + (addr) + (pl) * PLANE + i * 2048);
#endif
#if WD_ABL & 2
#define SPLIT_ROW(i_)
#elif WD_ABL & 1
#define SPLIT_ROW(i_)                                                                                         \
    if ((i_) < MT) { a_store_row(wr, (i_)); }
#elif WD_SPLIT_ADEEP
    // A rows two K steps ahead (a second register set): araw = step kt + 2 (split now), aahead = step kt + 3, reload with step kt + 4
#define SPLIT_ROW(i_)                                                                                         \
    if ((i_) < MT) { a_store_row(wr, (i_)); araw[(i_)] = aahead[(i_)]; aahead[(i_)] = (MODE == 1 && WD_SPLIT_CURSOR && !(WD_ABL & 128)) ? a_fetch_cur((i_)) : a_fetch(kt + 4, (i_)); }
#else
#define SPLIT_ROW(i_)                                                                                         \
    if ((i_) < MT) { a_store_row(wr, (i_)); a_load_row(kt + 3, (i_)); }
#endif
#if WD_ABL & 4
#define WLOAD(sub, slot)
#else
#define WLOAD(sub, slot) w_load(sub, slot);
#endif
    RD(rofs0, 2) RD(rofs0, 1) RD(rofs0, 0)
    SB;
#if WD_SPLIT_PRIO
    // experiment: the second wave of every SIMD (waves 4-7) at a higher issue priority - the two waves of a SIMD then stop marching in lockstep
    if (wave >= 4) __builtin_amdgcn_s_setprio(WD_SPLIT_PRIO);
#endif
    // EXPERIMENT (-DWD_SPLIT_RESPF=3|6, off by default): residual tile -> L2 a few K steps before the epilogue asks for it, one dword per 128-byte
    // line, results never read (inline asm: older than any load the compiler tracks, so its vmcnt waits only get more conservative; the destination
    // registers stay allocated until the epilogue is over).  The idea: every workgroup starts its epilogue with a cold HBM round trip for 1 KiB x BM
    // of residual, all at the same moment (8 k cycles of epilogue without a residual, 24 k with one).  Measured (three alternating rounds, res4 / res3 /
    // res2 with the full epilogue, profiles/r05_split_respf.txt): 128 -> 132, 144 -> 149, 192 -> 197 us; e2e 37.8 -> 37.5 frames/s on that box: the
    // epilogue is bound by the HBM burst itself, not by its latency, and the early requests only compete with the A / W stream.
    constexpr int RESPF_N = (WD_SPLIT_RESPF && MT <= 5) ? (BM * 8 + NTHREADS - 1) / NTHREADS : 0;
    float respf[RESPF_N > 0 ? RESPF_N : 1];
    const int respf_at = (nk > WD_SPLIT_RESPF) ? nk - WD_SPLIT_RESPF : 0;
    const bool respf_on = RESPF_N > 0 && p.residual != nullptr && p.splitk == 1;
    int cur = 0, nxt = BUF, wr = 2 * BUF;
    if (MODE == 1) f_set(4);
    for (int kt = 0; kt < nk; ++kt) {
        const int a1 = cur + rofs1, a0n = nxt + rofs0;
        if constexpr (RESPF_N > 0) {
            if (respf_on && kt == respf_at) {
                const int ncols = p.N - n0 < BN ? p.N - n0 : BN;
#pragma unroll
                for (int j = 0; j < RESPF_N; ++j) {
                    const int L = tid + NTHREADS * j, row = L >> 3, seg = L & 7;
                    const bool ok = row < BM && m0 + row < p.M && seg * 32 < ncols;
                    const float* q = p.residual + (ok ? (size_t)(m0 + row) * p.ldc + n0 + seg * 32 : (size_t)0);
                    asm volatile("global_load_dword %0, %1, off" : "=v"(respf[j]) : "v"(q) : "memory");
                }
            }
        }
        // sub-step 0 (fragments in registers); its slots request sub-step 1's fragments of the same buffer.  The W fragments of a sub-step are
        // reloaded right behind its last MFMA, a whole K step ahead of their next use (measured: spreading the eight waves' reloads over
        // different slots of the other sub-step - no burst in the vector-memory path, but 2-5 slots of lead - costs 9 % of the main loop)
        MF(2, 0, 0) RD(a1, 2) SB;
        MF(1, 1, 0) SPLIT_ROW(0) SB;
        MF(1, 0, 0) RD(a1, 1) SB;
        MF(0, 2, 0) SPLIT_ROW(1) SB;
        MF(0, 1, 0) SPLIT_ROW(2) SB;
        MF(0, 0, 0) RD(a1, 0) WLOAD(2 * kt + 2, 0) SB;
        // sub-step 1; its slots request sub-step 0 of the NEXT K step's buffer (complete since the previous barrier)
        MF(2, 0, 1) RD(a0n, 2) SB;
        MF(1, 1, 1) SPLIT_ROW(3) SB;
        MF(1, 0, 1) RD(a0n, 1) SB;
        MF(0, 2, 1) SPLIT_ROW(4) SB;
        MF(0, 1, 1) SPLIT_ROW(5) SB;
        MF(0, 0, 1) RD(a0n, 0) WLOAD(2 * kt + 3, 1) SB;
        // this wave's LDS writes (older than the MT prefetch reads just issued) are done; the prefetch stays in flight across the barrier
#if !(WD_ABL & 8)
        __builtin_amdgcn_s_waitcnt(0xC07F | (MT << 8));
        __builtin_amdgcn_s_barrier();
#endif
        SB;
        const int t = cur; cur = nxt; nxt = wr; wr = t;
        if (MODE == 1) f_advance(kt + 5);         // the next iteration fetches K step kt + 5
    }
#undef SB
#undef MF
#undef RD
#undef SPLIT_ROW
#undef WLOAD
#if WD_SPLIT_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    long long t_epi = 0;
    if (p.stamps) t_epi = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);          // the dangling prefetch of the step behind the last one
    __builtin_amdgcn_s_barrier();
    if (p.splitk > 1) {                            // raw partial tile of this K slice; bias / residual / ReLU in the reduce launch
        SplitArgs q = p;
        q.out = p.part + (size_t)kz * p.M * p.N;
        q.ldc = p.N;
        q.bias = nullptr; q.residual = nullptr; q.relu = 0;
        split_epilogue<MT>(q, smem, acc, m0, n0, wave, lane);
    } else {
        split_epilogue<MT>(p, smem, acc, m0, n0, wave, lane);
    }
    if constexpr (RESPF_N > 0) {
#pragma unroll
        for (int j = 0; j < RESPF_N; ++j) asm volatile("" :: "v"(respf[j]));       // the prefetch targets: allocated until here
    }
    if (p.stamps && tid == 0) {
        long long* o = p.stamps + 8 * (long)blockIdx.x;
        o[0] = t_start; o[1] = t_main; o[2] = t_epi; o[3] = __builtin_amdgcn_s_memtime();
        o[4] = id; o[5] = __builtin_amdgcn_s_getreg(20 /* XCC_ID */ | (0 << 6) | (3 << 11));
        o[6] = r_start; o[7] = __builtin_amdgcn_s_memrealtime();            // constant 100 MHz clock, common to the chip
    }
}

// ---- round 5, third structure: FOUR waves (one per SIMD, up to 512 registers each), wave tile (32 MT) x 64 ------------------------------
// The ring of the kernel above, but a wave owns TWO 32-column tiles: every A fragment read from LDS feeds 4 MFMAs instead of 2 (half the
// ds_read traffic per MFMA - the ablations priced the fragment reads at 16 % of the kernel), and with the whole register file of a SIMD to
// itself the wave can afford a SECOND accumulator set: DUAL = the five small cross terms accumulate apart from hi.hi, so the large
// accumulator is rounded 2 K / 16 times instead of 12 K / 16 (error against float64 ~ 1/3 of an f32 fmaf chain's); the two sets are added in
// the epilogue.  Everything is software-pipelined inside the one wave (fragment reads >= 3 slots of 2 MT MFMAs ahead, W fragments and A rows
// a whole K step ahead), so no second wave is needed to hide latency.  A rows are staged as float4 per thread (256 threads: 8 x 16 bytes per
// 32-float row, 5 row blocks), three ds_write_b64 per row block.
constexpr int W4_THREADS = 256;
// row blocks per epilogue pass of the four-wave kernel: 3 = 96 KB of LDS; 1 = 32 KB, so that the ring (18 KB per row block) sizes the workgroup and two
// workgroups of MT <= 3 share a CU (their prologues / epilogues then run under each other's main loop: the short-K shapes)
#ifndef WD_W4_PASS
#define WD_W4_PASS 3
#endif

template <int MT, bool DUAL>
__device__ __forceinline__ void split_epilogue_w4(const SplitArgs& p, unsigned char* smem, f32x16 (&acc)[2][MT], const f32x16 (&accs)[2][DUAL ? MT : 1], int m0, int n0,
                                                  int wave, int lane) {
    const int rr = lane & 31, rg = lane >> 5;
    float* ct = reinterpret_cast<float*>(smem);
    const int ncols = p.N - n0 < BN ? p.N - n0 : BN;
    const int c4 = 4 * lane;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && c4 < ncols) bv = *reinterpret_cast<const float4*>(p.bias + n0 + c4);
#pragma unroll
    for (int i0 = 0; i0 < MT; i0 += WD_W4_PASS) {
        constexpr int PASS = WD_W4_PASS;
        if (i0 > 0) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = i0; i < i0 + PASS && i < MT; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[c][i][e];
                    if (DUAL) v += accs[c][DUAL ? i : 0][e];
                    ct[((i - i0) * 32 + (e & 3) + 8 * (e >> 2) + 4 * rg) * BN + 64 * wave + 32 * c + rr] = v;
                }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        const int nrows = 32 * ((MT - i0) < PASS ? (MT - i0) : PASS);
#pragma unroll 4
        for (int r = wave; r < nrows; r += 4) {
            const int row = m0 + 32 * i0 + r;
            if (row < p.M && c4 < ncols) {
                float4 v = *reinterpret_cast<const float4*>(ct + r * BN + c4);
                const size_t o = (size_t)row * p.ldc + n0 + c4;
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (p.residual) {
                    const float4 q = *reinterpret_cast<const float4*>(p.residual + o);
                    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(p.out + o) = v;
            }
        }
    }
}

template <int MT, int MODE, bool DUAL>
__global__ __launch_bounds__(W4_THREADS, 1) void gemm_split_w4_kernel(const SplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 32 * MT;
    constexpr int PLANE = BM * 64;
    constexpr int BUF = 3 * PLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = p.tiles_m * p.tiles_n;
    int id;
    {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = total >> 3, r = total & 7;
        if (j >= q + (x < r ? 1 : 0)) return;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    long long t_start = 0, r_start = 0;
    if (p.stamps) { t_start = __builtin_amdgcn_s_memtime(); r_start = __builtin_amdgcn_s_memrealtime(); }
    int tm, tn;
    if (p.xmap == 0) { tm = id / p.tiles_n; tn = id - tm * p.tiles_n; } else { tn = id / p.tiles_m; tm = id - tn * p.tiles_m; }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = p.K / 32;

    // ---- A staging: thread -> (row srow + 32 i, float4 sk4 of the 32-float K step) ----
    const int srow = tid >> 3, sk4 = tid & 7;
    unsigned aoff[MT];                            // element offset of the thread's float4 at K step 0 (MODE 1: tap (0, 0), channel 0; may wrap below 0)
    unsigned vmask[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = m0 + srow + 32 * i;
        m = m < p.M ? m : p.M - 1;
        if (MODE == 0) {
            aoff[i] = (unsigned)(m * p.lda) + 4u * sk4;
            vmask[i] = 1u;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
            const int y0 = yo * p.stride - p.pad, x0 = xo * p.stride - p.pad;
            aoff[i] = (unsigned)(((b * p.H + y0) * p.W + x0) * p.C + 4 * sk4);
            unsigned vm = 0;
            for (int t = 0; t < p.ksize * p.ksize; ++t) {
                const int yy = y0 + t / p.ksize, xx = x0 + t % p.ksize;
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) vm |= 1u << t;
            }
            vmask[i] = vm;
        }
    }
    const int kc = (MODE == 1) ? p.C / 32 : 1;
    float4 araw[MT];
    auto a_load_row = [&](int kt, int i) {
        kt = kt < nk ? kt : nk - 1;
        if (MODE == 0) {
            araw[i] = *reinterpret_cast<const float4*>(p.a + (size_t)(aoff[i] + (unsigned)(kt * 32)));       // scalar base + 32-bit offset
        } else {
            const int tap = kt / kc, cb = kt - tap * kc;
            const int dy = tap / p.ksize, dx = tap - dy * p.ksize;
            const unsigned delta = (unsigned)((dy * p.W + dx) * p.C + cb * 32);
            const bool ok = ((vmask[i] >> tap) & 1u) != 0;
            const unsigned off = ok ? aoff[i] + delta : 4u * sk4;
            const float4 v = *reinterpret_cast<const float4*>(p.a + (size_t)off);
            araw[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int wofs = srow * 64 + ((((sk4 >> 1) ^ ((srow >> 2) & 3))) << 4) + ((sk4 & 1) << 3);
    auto a_store_row = [&](int bufoff, int i) {
        unsigned char* base = smem + bufoff + wofs + i * 2048;
        unsigned h0, m0_, l0, h1, m1, l1;
        split_pair(araw[i].x, araw[i].y, h0, m0_, l0);
        split_pair(araw[i].z, araw[i].w, h1, m1, l1);
        *reinterpret_cast<uint2*>(base) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(base + PLANE) = make_uint2(m0_, m1);
        *reinterpret_cast<uint2*>(base + 2 * PLANE) = make_uint2(l0, l1);
    };

    // ---- W fragments: this wave's two 32-column tiles ----
    const int nt32 = (n0 >> 5) + 2 * wave;
    const int ntiles = p.N >> 5;
    const size_t wstride = (size_t)(p.K / 16) * 192;
    const uint4* wbase0 = p.w + (size_t)(nt32 < ntiles ? nt32 : 0) * wstride + lane;
    const uint4* wbase1 = p.w + (size_t)(nt32 + 1 < ntiles ? nt32 + 1 : 0) * wstride + lane;
    const int nsub = nk * 2;
    bf16x8 wf[2][2][3];
    auto w_load = [&](int sub, int slot) {
        sub = sub < nsub ? sub : nsub - 1;
        const uint4* q0 = wbase0 + (size_t)sub * 192;
        const uint4* q1 = wbase1 + (size_t)sub * 192;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            wf[slot][0][pl] = __builtin_bit_cast(bf16x8, q0[pl * 64]);
            wf[slot][1][pl] = __builtin_bit_cast(bf16x8, q1[pl * 64]);
        }
    };

    f32x16 acc[2][MT], accs[2][DUAL ? MT : 1];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[c][i][e] = 0.f;
                if (DUAL) accs[c][DUAL ? i : 0][e] = 0.f;
            }

    const int rr = lane & 31, rg = lane >> 5;
    const int rofs0 = rr * 64 + ((((0 + rg) ^ ((rr >> 2) & 3))) << 4);
    const int rofs1 = rr * 64 + ((((2 + rg) ^ ((rr >> 2) & 3))) << 4);

#pragma unroll
    for (int i = 0; i < MT; ++i) a_load_row(0, i);
    w_load(0, 0);
    w_load(1, 1);
#pragma unroll
    for (int i = 0; i < MT; ++i) { a_store_row(0, i); a_load_row(1, i); }
#pragma unroll
    for (int i = 0; i < MT; ++i) { a_store_row(BUF, i); a_load_row(2, i); }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();

    bf16x8 af[MT][3];
#define SB __builtin_amdgcn_sched_barrier(0)
    // MFS: a small cross term (goes to the second accumulator set under DUAL); MFH: hi.hi
#define MFS(pa, pb, slot)                                                                                     \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) _Pragma("unroll") for (int i = 0; i < MT; ++i) {           \
        if (DUAL) accs[c][DUAL ? i : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], wf[slot][c][pb], accs[c][DUAL ? i : 0], 0, 0, 0); \
        else acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], wf[slot][c][pb], acc[c][i], 0, 0, 0);       \
    }
#define MFH(slot)                                                                                             \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) _Pragma("unroll") for (int i = 0; i < MT; ++i)             \
        acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], wf[slot][c][0], acc[c][i], 0, 0, 0);
#define RD(addr, pl)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
        af[i][pl] = *reinterpret_cast<const bf16x8*>(smem + (addr) + (pl) * PLANE + i * 2048);
#define SPLIT_ROW(i_)                                                                                         \
    if ((i_) < MT) { a_store_row(wr, (i_)); a_load_row(kt + 3, (i_)); }
    RD(rofs0, 2) RD(rofs0, 1) RD(rofs0, 0)
    SB;
    long long t_main = 0;
    if (p.stamps) t_main = __builtin_amdgcn_s_memtime();
    int cur = 0, nxt = BUF, wr = 2 * BUF;
    for (int kt = 0; kt < nk; ++kt) {
        const int a1 = cur + rofs1, a0n = nxt + rofs0;
        MFS(2, 0, 0) RD(a1, 2) SB;
        MFS(1, 1, 0) SPLIT_ROW(0) SB;
        MFS(1, 0, 0) RD(a1, 1) SB;
        MFS(0, 2, 0) SPLIT_ROW(1) SB;
        MFS(0, 1, 0) SPLIT_ROW(2) SB;
        MFH(0) RD(a1, 0) w_load(2 * kt + 2, 0); SB;
        MFS(2, 0, 1) RD(a0n, 2) SB;
        MFS(1, 1, 1) SPLIT_ROW(3) SB;
        MFS(1, 0, 1) RD(a0n, 1) SB;
        MFS(0, 2, 1) SPLIT_ROW(4) SB;
        MFS(0, 1, 1) SPLIT_ROW(5) SB;
        MFH(1) RD(a0n, 0) w_load(2 * kt + 3, 1); SB;
        __builtin_amdgcn_s_waitcnt(0xC07F | (MT << 8));
        __builtin_amdgcn_s_barrier();
        SB;
        const int t = cur; cur = nxt; nxt = wr; wr = t;
    }
#undef SB
#undef MFS
#undef MFH
#undef RD
#undef SPLIT_ROW
    long long t_epi = 0;
    if (p.stamps) t_epi = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    split_epilogue_w4<MT, DUAL>(p, smem, acc, accs, m0, n0, wave, lane);
    if (p.stamps && tid == 0) {
        long long* o = p.stamps + 8 * (long)blockIdx.x;
        o[0] = t_start; o[1] = t_main; o[2] = t_epi; o[3] = __builtin_amdgcn_s_memtime();
        o[4] = id; o[5] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        o[6] = r_start; o[7] = __builtin_amdgcn_s_memrealtime();
    }
}

// W (N, K) f32 -> packed planes [N32 / 32][K / 16][3][64] x 16 bytes; one thread per (column, 8 consecutive k)
// element (n, k) of the matrix = w[n * sn + k * sk]: (K, 1) for a row-major (N, K) weight, (1, N) for the transpose of a row-major (K, N) one
__global__ __launch_bounds__(256) void gemm_split_pack_kernel(const float* __restrict__ w, int N, int K, long sn, long sk, uint4* __restrict__ out, long total) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int k8 = K / 8;
    const int n = (int)(t / k8), kq = (int)(t - (long)n * k8);          // k = 8 kq .. 8 kq + 7
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = n < N ? w[(size_t)n * sn + (size_t)(8 * kq + e) * sk] : 0.f;
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_pair(v[2 * e], v[2 * e + 1], h[e], m[e], l[e]);
    const int nt = n >> 5, ks = kq >> 1, ln = (n & 31) + 32 * (kq & 1);
    uint4* dst = out + ((size_t)nt * (K / 16) + ks) * 192 + ln;
    dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
    dst[64] = make_uint4(m[0], m[1], m[2], m[3]);
    dst[128] = make_uint4(l[0], l[1], l[2], l[3]);
}

// Many weights in ONE launch (the training step: every trainable 1x1 / 3x3 / FC weight is re-packed after the optimizer step, forward and
// backward-data orientation - ~500 launches of ~10 us otherwise).  A descriptor addresses its source through strides, so convolution weights
// (N, C, ks, ks) are read where they lie (no permute / flip copies): element (n, k), k = (kh * ks + kw) * C + c, = src[n s_n + c s_c + kh' s_kh + kw' s_kw]
// with (kh', kw') = (ks - 1 - kh, ks - 1 - kw) when `flip` (the backward-data convolution).  Workgroup -> descriptor by binary search over first_block.
__global__ __launch_bounds__(256) void gemm_split_pack_batch_kernel(const WdSplitPackDesc* __restrict__ descs, int count) {
    int lo = 0, hi = count - 1;
    const long blk = blockIdx.x;
    while (lo < hi) {                                   // last descriptor with first_block <= blk (uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const WdSplitPackDesc d = descs[lo];
    const long t = (blk - d.first_block) * 256 + threadIdx.x;
    const int k8 = d.K / 8;
    const long total = (long)((d.N + 31) / 32) * 32 * k8;
    if (t >= total) return;
    const int n = (int)(t / k8), kq = (int)(t - (long)n * k8);
    const int k = 8 * kq, tap = k / d.C, c = k - tap * d.C;          // 8 consecutive k share a tap (C % 8 == 0)
    int kh = tap / d.ksize, kw = tap - kh * d.ksize;
    if (d.flip) { kh = d.ksize - 1 - kh; kw = d.ksize - 1 - kw; }
    const float* src = d.src + (size_t)n * d.s_n + (size_t)kh * d.s_kh + (size_t)kw * d.s_kw + (size_t)c * d.s_c;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = n < d.N ? src[(size_t)e * d.s_c] : 0.f;
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_pair(v[2 * e], v[2 * e + 1], h[e], m[e], l[e]);
    const int nt = n >> 5, ks = kq >> 1, ln = (n & 31) + 32 * (kq & 1);
    uint4* dst = reinterpret_cast<uint4*>(d.dst) + ((size_t)nt * (d.K / 16) + ks) * 192 + ln;
    dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
    dst[64] = make_uint4(m[0], m[1], m[2], m[3]);
    dst[128] = make_uint4(l[0], l[1], l[2], l[3]);
}

// out = act(sum over the K slices in slice order + bias + residual): deterministic, one float4 per thread
__global__ __launch_bounds__(256) void gemm_split_reduce_kernel(const float4* __restrict__ part, int splitk, long mn4, int n4, const float4* __restrict__ bias,
                                                                const float* __restrict__ residual, long ldc, int relu, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= mn4) return;
    float4 v = part[i];
    for (int z = 1; z < splitk; ++z) {
        const float4 q = part[(size_t)z * mn4 + i];
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    const long row = i / n4;
    const int c4 = (int)(i - row * n4);
    if (bias) { const float4 b = bias[c4]; v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    const size_t o = (size_t)row * ldc + 4 * c4;
    if (residual) { const float4 q = *reinterpret_cast<const float4*>(residual + o); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(out + o) = v;
}

// Tile height (32 MT rows) and K slices for a shape.  Cost model in microseconds from the measured ring kernel (MI355X, profiles/r05_split_*):
// a K step of 32 costs ~0.52 us per 32-row block of the tile, prologue + epilogue ~(2 + MT) us per workgroup, 256 workgroups run at once;
// a split adds the reduce launch (3 us + the partial tiles through HBM at ~4 TB/s).  Shapes with few tiles (FPN p5 / p6, the box-head FC)
// fill the chip through K slices; large ones pick the tile height with the fewest idle CU-rounds.
struct Plan { int mt, splitk; };
int kernel_choice();

Plan pick_plan(long M, int N, int K, bool allow_split) {
    static const int forced_mt = []() { const char* e = getenv("WD_SPLIT_MT"); return e ? atoi(e) : 0; }();
    static const int forced_sk = []() { const char* e = getenv("WD_SPLIT_SPLITK"); return e ? atoi(e) : 0; }();
    const long tn = (N + BN - 1) / BN;
    const int nk_all = K / 32;
    Plan best{5, 1};
    double best_t = 1e30;
    // measured cost per row block and K step relative to MT = 4 / 5 (box-head conv 49000 x 256 x 2304, tools/conv_split_one.py): MT = 3 and MT = 6 +24 %
    const int mt_max = kernel_choice() == 0 ? 6 : 5;            // the ring kernel fits 6 row blocks in 256 registers
    // Tile heights of 2 and 3 row blocks (64 / 96 rows) are NOT offered (only when forced with WD_SPLIT_MT).  Round-5 finding (tools/diag_victim.py,
    // tools/diag_two_models.py, tools/diag_canary.py): while such a launch is in flight, launches of the round-1 deformable kernel
    // (deform_conv3x3_kernel<64, true>) and of the grouped 3x3 kernel on ANOTHER stream return a few hundred slightly wrong outputs each (196 of 200
    // launches at MT = 2, 38 of 200 at MT = 3, none at MT = 4 / 5; none next to idle workgroups holding the same LDS; a canary workgroup - LDS
    // pattern, 200 live registers, VALU and f32-MFMA chains - next to the same launches stays clean, the split kernel's own results are never
    // affected, reserving 112 KiB of LDS does not help).  Narrowed down (profiles/r05_costream_interference.txt): it takes co-residency on a CU AND
    // bf16 MFMAs on real operand data in the small-tile workgroups (zero operands, s_nop in place of the MFMAs, dirty LDS, busy canaries: all clean) -
    // no software state is shared; with >= 4 row blocks no SIMD has room for those 230 / 256-VGPR kernels next to two split waves.  The two-pipeline test
    // (tests/test_gpu_e2e.py::test_two_pipelines_in_flight_on_different_streams_equal_serial_runs) is bit-identical with >= 4 row blocks.
    const int mt_min = (forced_mt == 2 || forced_mt == 3) ? forced_mt : 4;
    for (int mt = mt_max; mt >= mt_min; --mt) {
        if (forced_mt >= 2 && forced_mt <= mt_max && mt != forced_mt) continue;
        const long tiles = ((M + 32 * mt - 1) / (32 * mt)) * tn;
        for (int sk = 1; sk <= 32; ++sk) {
            if (sk > 1 && !allow_split) break;
            if (forced_sk >= 1 && sk != forced_sk && allow_split) continue;
            const int per = (nk_all + sk - 1) / sk;
            if (sk > 1 && (per < 3 || (long)(sk - 1) * per >= nk_all)) continue;
            const long rounds = (tiles * sk + 255) / 256;
            const double eff = (mt == 4 || mt == 5) ? 1.0 : 1.24;
            double t = rounds * (mt * per * 0.52 * eff + 2.0 + mt);
            if (sk > 1) t += 3.0 + (double)(sk + 1) * M * N * 4.0 / 4.0e6;
            if (t < best_t * 0.97) { best_t = t; best = Plan{mt, sk}; }      // ties to the larger tile / fewer slices
        }
    }
    return best;
}

long long* g_stamps = nullptr;           // diagnostics (wd_gemm_split_debug_stamps)

// WD_SPLIT_KERNEL = ring (8 waves, wave tile 32 MT x 32) | k64 (double-buffered K-64 structure) | w4 (4 waves, wave tile 32 MT x 64) |
// w4d (w4 with the second accumulator set)
int kernel_choice() {
    static const int v = []() {
        const char* e = getenv("WD_SPLIT_KERNEL");
        if (!e) return 0;
        return !strcmp(e, "k64") ? 1 : !strcmp(e, "w4") ? 2 : !strcmp(e, "w4d") ? 3 : 0;
    }();
    return v;
}

template <int MT, int MODE>
int launch(const SplitArgs& a, hipStream_t stream) {
    const int kc = MT <= 5 ? kernel_choice() : 0;           // the A/B structures exist for up to 5 row blocks
    constexpr size_t lds_epi = 32u * (MT < 3 ? MT : 3) * BN * 4u;
    constexpr size_t lds_k64 = 2u * 3u * 32u * MT * 128u, lds_ring = 3u * 3u * 32u * MT * 64u;
    const size_t lds_main = kc == 1 ? lds_k64 : lds_ring;
    constexpr size_t lds_epi_w4 = 32u * (MT < WD_W4_PASS ? MT : WD_W4_PASS) * BN * 4u;
    const size_t lds_e = kc >= 2 ? lds_epi_w4 : lds_epi;
    const size_t lds = lds_main > lds_e ? lds_main : lds_e;
    const void* fn = reinterpret_cast<const void*>(gemm_split_kernel<MT, MODE>);
    if constexpr (MT <= 5) {
        if (kc == 1) fn = reinterpret_cast<const void*>(gemm_split_k64_kernel<MT, MODE>);
        if (kc == 2) fn = reinterpret_cast<const void*>(gemm_split_w4_kernel<MT, MODE, false>);
        if (kc == 3) fn = reinterpret_cast<const void*>(gemm_split_w4_kernel<MT, MODE, true>);
    }
    static bool attr_set[16] = {};
    int dev = 0;
    WT_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        WT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    const int total = a.tiles_m * a.tiles_n * a.splitk;
    const dim3 grid((unsigned)((total + 7) / 8 * 8));
    if constexpr (MT <= 5) {
        if (kc == 1) hipLaunchKernelGGL((gemm_split_k64_kernel<MT, MODE>), grid, dim3(NTHREADS), lds, stream, a);
        else if (kc == 2) hipLaunchKernelGGL((gemm_split_w4_kernel<MT, MODE, false>), grid, dim3(W4_THREADS), lds, stream, a);
        else if (kc == 3) hipLaunchKernelGGL((gemm_split_w4_kernel<MT, MODE, true>), grid, dim3(W4_THREADS), lds, stream, a);
    }
    if (kc == 0) hipLaunchKernelGGL((gemm_split_kernel<MT, MODE>), grid, dim3(NTHREADS), lds, stream, a);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

// EXPERIMENT (WD_SPLIT_SERIALIZE=1): launches of this unit on different streams are chained through one event, so two of them are never
// in flight together
struct CrossStream {
    hipEvent_t ev = nullptr;
    bool recorded = false;
};
CrossStream g_chain;

template <int MODE>
int dispatch(SplitArgs& a, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    static const bool serialize = []() { const char* e = getenv("WD_SPLIT_SERIALIZE"); return e && e[0] == '1'; }();
    struct Guard {
        hipStream_t st; bool on;
        ~Guard() { if (on) { (void)hipEventRecord(g_chain.ev, st); g_chain.recorded = true; } }
    } guard{stream, serialize};
    if (serialize) {
        if (!g_chain.ev) WT_HIP(hipEventCreateWithFlags(&g_chain.ev, hipEventDisableTiming));
        if (g_chain.recorded) WT_HIP(hipStreamWaitEvent(stream, g_chain.ev, 0));
    }
    a.stamps = g_stamps;
    const bool can_split = kernel_choice() == 0 && workspace != nullptr && (a.N % 4) == 0;
    Plan pl = pick_plan(a.M, a.N, a.K, can_split);
    if (pl.splitk > 1 && workspace_bytes < (size_t)pl.splitk * a.M * a.N * sizeof(float)) pl = pick_plan(a.M, a.N, a.K, false);
    const int mt = pl.mt;
    a.splitk = pl.splitk;
    a.part = pl.splitk > 1 ? (float*)workspace : nullptr;
    a.tiles_m = (a.M + 32 * mt - 1) / (32 * mt);
    a.tiles_n = (a.N + BN - 1) / BN;
    static const int xmap = []() { const char* e = getenv("WD_SPLIT_XMAP"); return e ? atoi(e) : 0; }();
    a.xmap = xmap;
    int rc;
    switch (mt) {
        case 2: rc = launch<2, MODE>(a, stream); break;
        case 3: rc = launch<3, MODE>(a, stream); break;
        case 4: rc = launch<4, MODE>(a, stream); break;
        case 6: rc = launch<6, MODE>(a, stream); break;
        default: rc = launch<5, MODE>(a, stream); break;
    }
    if (rc != WT_OK || a.splitk == 1) return rc;
    const long mn4 = (long)a.M * a.N / 4;
    hipLaunchKernelGGL(gemm_split_reduce_kernel, dim3((unsigned)((mn4 + 255) / 256)), dim3(256), 0, stream, (const float4*)a.part, a.splitk, mn4, a.N / 4,
                       (const float4*)a.bias, a.residual, a.ldc, a.relu, a.out);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // namespace

extern "C" {

/* Diagnostics: following launches write eight int64 per workgroup (s_memtime at start / main loop / epilogue / end, tile id, XCC id,
 * s_memrealtime at start / end) to `buf` (device memory, 8 x grid size entries; nullptr switches it off).  tools/gemm_split_stamps.py */
int wd_gemm_split_debug_stamps(long long* buf) {
    g_stamps = buf;
    return WT_OK;
}

size_t wd_gemm_split_packed_bytes(int N, int K) {
    if (N <= 0 || K <= 0 || (K % BK)) return 0;
    return (size_t)((N + 31) / 32) * 32 * (size_t)K * 6;
}

int wd_gemm_split_pack_weight_strided(const float* w, int N, int K, long stride_n, long stride_k, void* packed, void* stream_) {
    WT_TRY(wt::ensure_device());
    if (!w || !packed || N <= 0 || K <= 0 || (K % BK)) {
        wt::set_error("wd_gemm_split_pack_weight: K must be a positive multiple of %d (N=%d K=%d)", BK, N, K);
        return WT_ERR_INVALID;
    }
    const long total = (long)((N + 31) / 32) * 32 * (K / 8);
    hipLaunchKernelGGL(gemm_split_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, w, N, K, stride_n, stride_k,
                       (uint4*)packed, total);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

int wd_gemm_split_pack_weight(const float* w, int N, int K, void* packed, void* stream_) {
    return wd_gemm_split_pack_weight_strided(w, N, K, (long)K, 1, packed, stream_);
}

int wd_gemm_split_pack_batch(const WdSplitPackDesc* descs_device, int count, long total_blocks, void* stream_) {
    WT_TRY(wt::ensure_device());
    if (count <= 0 || total_blocks <= 0) return WT_OK;
    if (!descs_device || total_blocks >= (1l << 31)) {
        wt::set_error("wd_gemm_split_pack_batch: NULL descriptor array or too many workgroups (%ld)", total_blocks);
        return WT_ERR_INVALID;
    }
    hipLaunchKernelGGL(gemm_split_pack_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream_, descs_device, count);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

/* Bytes of scratch the K-sliced form of a shape wants (0: the shape runs unsliced).  Passing less (or NULL) is legal: the call then runs unsliced. */
size_t wd_gemm_split_workspace(long M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % BK) || (N % 32) || kernel_choice() != 0) return 0;
    const Plan pl = pick_plan(M, N, K, true);
    return pl.splitk > 1 ? (size_t)pl.splitk * M * N * sizeof(float) : 0;
}

int wd_gemm_split_f32(const float* a, long lda, const void* packed_w, const float* bias, const float* residual, float* out, long ldc,
                      int M, int N, int K, int relu, void* workspace, size_t workspace_bytes, void* stream_) {
    WT_TRY(wt::ensure_device());
    if (M <= 0 || N <= 0) return WT_OK;
    if (!a || !packed_w || !out || K <= 0 || (K % BK) || (N % 32) || (lda & 3) || (ldc & 3) || ((uintptr_t)a & 15) || ((uintptr_t)packed_w & 15) ||
        ((uintptr_t)out & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)residual & 15) || (long)M * lda >= (1l << 31)) {
        wt::set_error("wd_gemm_split_f32: needs K %% %d == 0, N %% 32 == 0, 16-byte aligned rows and M * lda < 2^31 (M=%d N=%d K=%d lda=%ld)", BK, M, N,
                      K, lda);
        return WT_ERR_INVALID;
    }
    SplitArgs s{};
    s.a = a; s.w = (const uint4*)packed_w; s.bias = bias; s.residual = residual; s.out = out;
    s.lda = lda; s.ldc = ldc; s.M = M; s.N = N; s.K = K; s.relu = relu;
    return dispatch<0>(s, workspace, workspace_bytes, (hipStream_t)stream_);
}

int wd_conv_split_f32(const float* x, int batch, int H, int W, int C, const void* packed_w, int ksize, int stride, int pad, const float* bias,
                      const float* residual, float* out, int N, int relu, void* workspace, size_t workspace_bytes, void* stream_) {
    WT_TRY(wt::ensure_device());
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const long M = (long)batch * Ho * Wo;
    if (M <= 0 || N <= 0) return WT_OK;
    if (!x || !packed_w || !out || (ksize != 1 && ksize != 3) || stride < 1 || pad < 0 || C <= 0 || (C % BK) || (N % 32) || ((uintptr_t)x & 15) ||
        ((uintptr_t)out & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)residual & 15) ||
        (long)batch * H * W * C >= (1l << 31) || M >= (1l << 31)) {
        wt::set_error("wd_conv_split_f32: needs ksize 1 or 3, C %% %d == 0, N %% 32 == 0 and fewer than 2^31 input elements (C=%d N=%d k=%d)", BK, C, N, ksize);
        return WT_ERR_INVALID;
    }
    SplitArgs s{};
    s.a = x; s.w = (const uint4*)packed_w; s.bias = bias; s.residual = residual; s.out = out;
    s.lda = C; s.ldc = N; s.M = (int)M; s.N = N; s.K = ksize * ksize * C; s.relu = relu;
    s.H = H; s.W = W; s.C = C; s.Ho = Ho; s.Wo = Wo; s.stride = stride; s.pad = pad; s.ksize = ksize;
    return dispatch<1>(s, workspace, workspace_bytes, (hipStream_t)stream_);
}

}  // extern "C"

// ---- diagnostics: a "canary" workgroup for co-residency experiments (tools/diag_canary.py) ----------------------------------------------
// 256 threads fill `lds_bytes` of LDS and 16 registers with a pattern, keep an f32 FMA chain and an f32 MFMA chain going for `spins` rounds
// and count, per kind, how often a value comes back different: flags[0] LDS, [1] registers, [2] VALU chain, [3] MFMA chain, [4] workgroups run.
namespace {
template <int NR>
__global__ __launch_bounds__(256, 2) void canary_kernel(int lds_bytes, int spins, unsigned* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char csm[];
    unsigned* l = reinterpret_cast<unsigned*>(csm);
    const int tid = threadIdx.x, n = lds_bytes / 4;
    for (int i = tid; i < n; i += 256) l[i] = 0x9E3779B9u * (unsigned)(i + 1) + blockIdx.x;
    unsigned r[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) { r[j] = 0x85EBCA6Bu * (unsigned)(tid * 16 + j + 1); asm volatile("" : "+v"(r[j])); }
    __syncthreads();
    unsigned bad_l = 0, bad_r = 0, bad_v = 0, bad_m = 0;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    for (int s = 0; s < spins; ++s) {
        // VALU chain with a known closed form: x <- x * 1 + 0 keeps x; (x + 1) - 1 exact for small integers
        float x = (float)(tid & 63);
#pragma unroll 8
        for (int j = 0; j < 64; ++j) x = __builtin_fmaf(x, 1.0f, 1.0f);
        if (x != (float)((tid & 63) + 64)) ++bad_v;
        // f32 MFMA chain with DISTINCT small-integer operands per lane and per step (all-ones operands cannot show an operand mix-up):
        // A_j[i][k] = i + 2 k + j, B_j[k][n] = n + 3 k + 1 + j (lane l holds A[l % 16][l / 16] and B[l / 16][l % 16]);
        // D[i][n] = sum_j sum_k A_j[i][k] B_j[k][n], exact in float32; lane l holds D[4 (l / 16) + r][l % 16], r = 0..3
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int ln = tid & 63, li = ln & 15, lk = ln >> 4;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)(li + 2 * lk + j + (s & 3)), (float)(li + 3 * lk + 1 + j), acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * lk + r, nn = li;
            int want = 0;
            for (int j = 0; j < 8; ++j)
                for (int k = 0; k < 4; ++k) want += (i + 2 * k + j + (s & 3)) * (nn + 3 * k + 1 + j);
            if (acc[r] != (float)want) ++bad_m;
        }
        for (int i = tid; i < n; i += 256)
            if (l[i] != 0x9E3779B9u * (unsigned)(i + 1) + blockIdx.x) ++bad_l;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            asm volatile("" : "+v"(r[j]));
            if (r[j] != 0x85EBCA6Bu * (unsigned)(tid * 16 + j + 1)) ++bad_r;
        }
    }
    if (bad_l) atomicAdd(flags + 0, bad_l);
    if (bad_r) atomicAdd(flags + 1, bad_r);
    if (bad_v) atomicAdd(flags + 2, bad_v);
    if (bad_m) atomicAdd(flags + 3, bad_m);
    if (tid == 0) atomicAdd(flags + 4, 1u);
}
}  // namespace

extern "C" int wd_debug_canary(int workgroups, int lds_bytes, int spins, unsigned* flags, void* stream_) {
    WT_TRY(wt::ensure_device());
    static const bool big = getenv("WD_CANARY_BIG") != nullptr;          // 200 live registers per lane (two waves per SIMD, like the old deformable kernel)
    if (big) hipLaunchKernelGGL(canary_kernel<200>, dim3((unsigned)workgroups), dim3(256), (size_t)lds_bytes, (hipStream_t)stream_, lds_bytes, spins, flags);
    else hipLaunchKernelGGL(canary_kernel<16>, dim3((unsigned)workgroups), dim3(256), (size_t)lds_bytes, (hipStream_t)stream_, lds_bytes, spins, flags);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

// diagnostics: workgroups that only occupy a CU slot (512 threads, `lds_bytes` of LDS) for ~`ticks` s_memtime ticks
namespace {
__global__ __launch_bounds__(512) void occupy_kernel(long long ticks, unsigned* __restrict__ sink) {
    extern __shared__ unsigned char osm[];
    if (ticks < 0) {                               // "dirty" occupant: leaves its whole LDS allocation full of NaN bit patterns
        ticks = -ticks;
        unsigned* w = reinterpret_cast<unsigned*>(osm);
        for (int i = threadIdx.x; i < (int)(sink[1] / 4); i += 512) w[i] = 0x7FC01234u;
        __syncthreads();
    }
    const long long t0 = __builtin_amdgcn_s_memtime();
    unsigned acc = 0;
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); acc += osm[threadIdx.x]; }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
}  // namespace

extern "C" int wd_debug_occupy(int workgroups, int lds_bytes, long long ticks, unsigned* sink, void* stream_) {
    WT_TRY(wt::ensure_device());
    WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)workgroups), dim3(512), (size_t)lds_bytes, (hipStream_t)stream_, ticks, sink);
    WT_HIP(hipGetLastError());
    return WT_OK;
}
