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

// MODE 0: plain row-major A.  MODE 1: NHWC convolution source (ksize 1 or 3, any stride / pad).
template <int MT, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_split_kernel(const SplitArgs p) {
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

    // ---- epilogue ----
    // C/D layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a lane owns ONE column.  The tile goes
    // through LDS (free now) so that global memory sees whole 1-KiB output rows: per pass up to 3 row blocks (96 rows x 256 floats); every
    // wave writes its 32-column strip, then reads whole rows as float4 per lane and adds bias / residual / ReLU on the way out.
    float* ct = reinterpret_cast<float*>(smem);
    const int ncols = p.N - n0 < BN ? p.N - n0 : BN;            // valid columns of this tile (multiple of 32)
    const int c4 = 4 * lane;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && c4 < ncols) bv = *reinterpret_cast<const float4*>(p.bias + n0 + c4);
#pragma unroll
    for (int i0 = 0; i0 < MT; i0 += 3) {
        constexpr int PASS = 3;
        if (i0 > 0) __builtin_amdgcn_s_barrier();              // the previous pass has been read
#pragma unroll
        for (int i = i0; i < i0 + PASS && i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                ct[((i - i0) * 32 + (e & 3) + 8 * (e >> 2) + 4 * rg) * BN + 32 * wave + rr] = acc[i][e];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        const int nrows = 32 * ((MT - i0) < PASS ? (MT - i0) : PASS);
#pragma unroll 4
        for (int r = wave; r < nrows; r += 8) {
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

// W (N, K) f32 -> packed planes [N32 / 32][K / 16][3][64] x 16 bytes; one thread per (column, 8 consecutive k)
__global__ __launch_bounds__(256) void gemm_split_pack_kernel(const float* __restrict__ w, int N, int K, uint4* __restrict__ out, long total) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int k8 = K / 8;
    const int n = (int)(t / k8), kq = (int)(t - (long)n * k8);          // k = 8 kq .. 8 kq + 7
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = n < N ? w[(size_t)n * K + 8 * kq + e] : 0.f;
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_pair(v[2 * e], v[2 * e + 1], h[e], m[e], l[e]);
    const int nt = n >> 5, ks = kq >> 1, ln = (n & 31) + 32 * (kq & 1);
    uint4* dst = out + ((size_t)nt * (K / 16) + ks) * 192 + ln;
    dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
    dst[64] = make_uint4(m[0], m[1], m[2], m[3]);
    dst[128] = make_uint4(l[0], l[1], l[2], l[3]);
}

int pick_mt(long M, int N) {
    // rows per tile = 32 MT: the fewest idle CU-rounds on 256 CUs wins, ties to the larger tile (fewer passes over W)
    static const int forced = []() { const char* e = getenv("WD_SPLIT_MT"); return e ? atoi(e) : 0; }();
    if (forced >= 2 && forced <= 5) return forced;
    const long tn = (N + BN - 1) / BN;
    int best = 5;
    double best_cost = 1e30;
    for (int mt = 5; mt >= 2; --mt) {            // MT = 6 needs more than 256 registers
        const long tiles = ((M + 32 * mt - 1) / (32 * mt)) * tn;
        const long rounds = (tiles + 255) / 256;
        const double cost = (double)rounds * mt;                  // time ~ rounds x rows per tile
        if (cost < best_cost * 0.999) { best_cost = cost; best = mt; }
    }
    return best;
}

template <int MT, int MODE>
int launch(const SplitArgs& a, hipStream_t stream) {
    constexpr size_t lds_main = 2u * 3u * 32u * MT * 128u, lds_epi = 32u * (MT < 3 ? MT : 3) * BN * 4u;
    constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    static bool attr_set[16] = {};
    int dev = 0;
    WT_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<MT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    const int total = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL((gemm_split_kernel<MT, MODE>), dim3((unsigned)((total + 7) / 8 * 8)), dim3(NTHREADS), lds, stream, a);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

template <int MODE>
int dispatch(SplitArgs& a, hipStream_t stream) {
    const int mt = pick_mt(a.M, a.N);
    a.tiles_m = (a.M + 32 * mt - 1) / (32 * mt);
    a.tiles_n = (a.N + BN - 1) / BN;
    static const int xmap = []() { const char* e = getenv("WD_SPLIT_XMAP"); return e ? atoi(e) : 0; }();
    a.xmap = xmap;
    switch (mt) {
        case 2: return launch<2, MODE>(a, stream);
        case 3: return launch<3, MODE>(a, stream);
        case 4: return launch<4, MODE>(a, stream);
        default: return launch<5, MODE>(a, stream);
    }
}

}  // namespace

extern "C" {

size_t wd_gemm_split_packed_bytes(int N, int K) {
    if (N <= 0 || K <= 0 || (K % BK)) return 0;
    return (size_t)((N + 31) / 32) * 32 * (size_t)K * 6;
}

int wd_gemm_split_pack_weight(const float* w, int N, int K, void* packed, void* stream_) {
    WT_TRY(wt::ensure_device());
    if (!w || !packed || N <= 0 || K <= 0 || (K % BK)) {
        wt::set_error("wd_gemm_split_pack_weight: K must be a positive multiple of %d (N=%d K=%d)", BK, N, K);
        return WT_ERR_INVALID;
    }
    const long total = (long)((N + 31) / 32) * 32 * (K / 8);
    hipLaunchKernelGGL(gemm_split_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, w, N, K, (uint4*)packed, total);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

int wd_gemm_split_f32(const float* a, long lda, const void* packed_w, const float* bias, const float* residual, float* out, long ldc,
                      int M, int N, int K, int relu, void* stream_) {
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
    return dispatch<0>(s, (hipStream_t)stream_);
}

int wd_conv_split_f32(const float* x, int batch, int H, int W, int C, const void* packed_w, int ksize, int stride, int pad, const float* bias,
                      const float* residual, float* out, int N, int relu, void* stream_) {
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
    return dispatch<1>(s, (hipStream_t)stream_);
}

}  // extern "C"
