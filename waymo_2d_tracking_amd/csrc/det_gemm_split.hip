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
// f32 accumulation.  Every bf16 x bf16 product is exact in f32.  The three dropped terms (mid.lo, lo.mid, lo.lo): with round-to-nearest
// planes |mid| <= 2^-9 |x| and |lo| <= 2^-17 |x| (half an ulp of the plane above), so they are bounded by ~2^-25 |a.b| per product in the
// worst case - the same order as ONE f32 product rounding, not below it; the other roundings are the accumulator's, one per MFMA (6 K / 16
// of them against K for the fmaf chain of the f32 MFMA).  What the tests hold is the MEASURED error against float64: not above the exact-f32
// kernels' on the tested distributions, including operands chosen to maximise mid and lo (tests/test_gpu_gemm_split.py,
// profiles/r05_split_gemm_error.txt).  Six bf16 MFMAs replace sixteen f32 MFMAs' worth of time: 0.375 of the f32 matrix time.
// Not carried: +-inf / NaN operands (x - hi = NaN -> the output element is NaN where the f32 GEMM gives +-inf) and operands below
// 2^-110 (their lo plane underflows bfloat16).
//
// Structure (one workgroup = 8 waves = one (32 MT) x 256 output tile; MT = 5 -> 160 x 256: 9600 x 1024 is 240 tiles on 256 CUs):
//   * W is static: packed once (wd_gemm_split_pack_weight) into MFMA B-fragment order, [N / 32][K / 16][plane][lane] 16-byte
//     entries.  Wave w owns columns 32 w .. 32 w + 31 of the tile and streams ITS fragments straight from L2 into registers
//     (1 KiB contiguous per wave-load, one K step ahead) - no wave shares them, so they never touch LDS.
//   * A is shared by all 8 waves through LDS as three bf16 planes [row][32 k] per K step (64-byte rows, 16-byte slots XOR-swizzled with
//     (row >> 2) & 3: conflict-free for the ds_read_b128 lane groups of gfx950).  Two ways in:
//       - gemm_split_kernel (MODE 0 / 1): A is f32 in HBM; a K step is loaded by the workgroup (float2 per thread), split in registers
//         (v_cvt_pk_bf16_f32 + shift / and + subtract: 5.5 VALU per element) and written with ds_write_b32 into a ring of three buffers;
//       - gemm_split_planes_kernel (round 6): A arrives PRE-SPLIT ("activation planes": the three bf16 planes of every 32-row x 32-k block
//         stored as the 6 KiB LDS image itself, written by the producing kernel's epilogue); the workgroup pulls a K step with
//         global_load_lds_dwordx4 (LDS-DMA: no staging registers, no VALU, no ds_write) into a ring of FOUR buffers, three K steps ahead.
//   * per 16-deep sub-step a wave reads 3 MT A fragments (ds_read_b128) and issues 6 MT MFMAs.
//   * MODE 1: the A row of an output row is the NHWC pixel of an output pixel for a (strided) 1x1 or 3x3 convolution (K step -> (tap,
//     channel block); rows whose tap leaves the image contribute zeros).
#include "common.h"
#include "../../include/waymodet.h"
#include <cstdlib>
#include <cstring>

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int BK = 64;            // granularity of K the entry points accept (two 32-deep K steps)
constexpr int BN = 256;           // 8 waves x 32 columns
constexpr int NTHREADS = 512;
constexpr int CHUNK = 6144;       // activation planes: bytes of one (32 rows x 32 k) block = 3 planes x 2048-byte LDS image

struct SplitArgs {
    const float* a;               // activations, f32 (MODE 0 / 1)
    const unsigned char* ap;      // activation planes (MODE 2): [ceil(M / 32)][K / 32][3][2048]
    const uint4* w;               // packed weight planes
    const float* bias;
    const float* residual;        // f32 residual (M, N; row stride ldc) or nullptr
    const unsigned char* resp;    // residual as activation planes of an (M, N) matrix, or nullptr (exact: hi + mid + lo == x)
    float* out;                   // f32 output or nullptr
    unsigned char* outp;          // output as activation planes of an (M, N) matrix (the next GEMM's A), or nullptr
    long lda, ldc;                // row strides (floats) of a (plain mode) and of out / residual
    int M, N, K, relu;
    int tiles_m, tiles_n, xmap;
    int splitk;                   // K slices per output tile; > 1: raw partial tiles go to part[slice][M][N]
    float* part;
    long long* stamps;            // WD_DEBUG builds: per-workgroup s_memtime stamps (start, main loop, epilogue, end) or nullptr
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

// the two floats of a packed bf16 pair
__device__ __forceinline__ float bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

// byte offset of the 8-byte group (row, columns col .. col + 3; col % 4 == 0) inside the hi plane of an (M, ncols) activation-planes matrix;
// mid / lo: + 2048 / + 4096.  The 2048-byte block of a plane is the LDS image the consumer reads: 64-byte rows, 16-byte slots XOR (row >> 2) & 3.
__device__ __forceinline__ size_t planes_offset(int row, int col, int ncols) {
    return ((size_t)(row >> 5) * (size_t)(ncols >> 5) + (size_t)(col >> 5)) * CHUNK +
           (size_t)((row & 31) * 64 + ((((col >> 3) & 3) ^ ((row >> 2) & 3)) << 4) + ((col >> 2) & 1) * 8);
}

// ---- epilogue ----
// C/D layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a lane owns ONE column.  The tile goes
// through LDS (free after the main loop's last barrier) so that global memory sees whole 1-KiB runs: per pass up to 3 row blocks
// (96 rows x 256 floats, row stride LDC = 264 floats: the 32-byte skew per row keeps both read patterns below conflict-free).
//   * f32 output only (row-major pass): every wave reads whole rows as float4 per lane, adds bias / residual / ReLU, stores 1 KiB per row;
//   * planes output (block-major pass): the unit of work is half a (32 x 32) block of the output = ONE 1-KiB piece of each plane: lane -> (row
//     16 h + lane / 4, the 8 columns whose bf16 values share a 16-byte slot of the LDS image), two ds_read_b128, bias / residual / ReLU, the
//     three-way split of the FINAL f32 value and three global_store_dwordx4 that are contiguous across the wave (and, if an f32 copy is wanted
//     too, two float4 stores that fill 128-byte lines per lane quad).  The residual comes in as f32 or as planes (three 1-KiB wave loads; summed
//     hi + mid + lo: exactly the f32 value the planes were split from).
constexpr int LDC = BN + 8;

// Row mapping (round 6, position-major convolution tiles): tile row r stands for matrix row m0 + r, valid while < mlim, and lives at output row
// (m0 + r) * row_mul + row_add (plain tiles: mlim = M, row_mul = 1, row_add = 0).
template <int MT>
__device__ __forceinline__ void split_epilogue(const SplitArgs& p, unsigned char* smem, const f32x16 (&acc)[MT], int m0, int n0, int wave, int lane, int mlim,
                                               int row_mul = 1, int row_add = 0) {
    const int rr = lane & 31, rg = lane >> 5;
    float* ct = reinterpret_cast<float*>(smem);
    const int ncols = p.N - n0 < BN ? p.N - n0 : BN;            // valid columns of this tile (multiple of 32)
    constexpr int PASS = 3;                                     // row blocks per pass
    auto stage = [&](int i0) {                                  // accumulators of row blocks i0 .. i0 + PASS - 1 -> ct
        if (i0 > 0) __builtin_amdgcn_s_barrier();              // the previous pass has been read
#pragma unroll
        for (int i = i0; i < i0 + PASS && i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                ct[((i - i0) * 32 + (e & 3) + 8 * (e >> 2) + 4 * rg) * LDC + 32 * wave + rr] = acc[i][e];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    };
    if (p.outp == nullptr) {
        // ---- row-major pass: f32 output ----
        const int c4 = 4 * lane;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && c4 < ncols) bv = *reinterpret_cast<const float4*>(p.bias + n0 + c4);
        constexpr int RPW = 32 * PASS / 8;                      // rows per wave and pass
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
                    if (r < nrows && row < mlim && c4 < ncols)
                        rv[j] = *reinterpret_cast<const float4*>(p.residual + (size_t)(row * row_mul + row_add) * p.ldc + n0 + c4);
                }
            } else if (p.resp) {
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    const int r = wave + 8 * j, row = m0 + 32 * i0 + r;
                    rv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (r < nrows && row < mlim && c4 < ncols) {
                        const unsigned char* q = p.resp + planes_offset(row * row_mul + row_add, n0 + c4, p.N);
                        const uint2 h = *reinterpret_cast<const uint2*>(q), m = *reinterpret_cast<const uint2*>(q + 2048),
                                    l = *reinterpret_cast<const uint2*>(q + 4096);
                        rv[j] = make_float4((bf_lo(h.x) + bf_lo(m.x)) + bf_lo(l.x), (bf_hi(h.x) + bf_hi(m.x)) + bf_hi(l.x),
                                            (bf_lo(h.y) + bf_lo(m.y)) + bf_lo(l.y), (bf_hi(h.y) + bf_hi(m.y)) + bf_hi(l.y));
                    }
                }
            }
            stage(i0);
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int r = wave + 8 * j, row = m0 + 32 * i0 + r;
                if (r < nrows && row < mlim && c4 < ncols) {
                    float4 v = *reinterpret_cast<const float4*>(ct + r * LDC + c4);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (p.residual || p.resp) { v.x += rv[j].x; v.y += rv[j].y; v.z += rv[j].z; v.w += rv[j].w; }
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *reinterpret_cast<float4*>(p.out + (size_t)(row * row_mul + row_add) * p.ldc + n0 + c4) = v;
                }
            }
        }
        return;
    }
    // ---- block-major pass: planes output (+ optional f32 copy) ----
    const int ur = lane >> 2;                                   // row inside the half block
    constexpr int UPW = PASS * 8 * 2 / 8;                       // units per wave and pass: (3 row blocks x 8 column blocks x 2 halves) / 8 waves
    const int ncb = ncols >> 5;
#pragma unroll
    for (int i0 = 0; i0 < MT; i0 += PASS) {
        const int nrb = (MT - i0) < PASS ? (MT - i0) : PASS;
        const int nunits = nrb * ncb * 2;
        // unit u -> (row block u / (2 ncb), column block (u / 2) % ncb, half u & 1): consecutive units of a wave... are 8 apart: spread over the tile
        uint4 rq[UPW][3];                                       // residual of the unit: planes (3 x 16 bytes) or f32 (2 x 16 bytes)
#pragma unroll
        for (int j = 0; j < UPW; ++j) {
            const int u = wave + 8 * j;
            rq[j][0] = rq[j][1] = rq[j][2] = make_uint4(0u, 0u, 0u, 0u);
            if (u < nunits && m0 + 32 * (i0 + u / (2 * ncb)) < mlim) {       // (row blocks past the matrix do not exist in the planes buffers)
                const int rbi = u / (2 * ncb), cb = (u >> 1) - rbi * ncb, h = u & 1;
                const int r = 16 * h + ur, row = m0 + 32 * (i0 + rbi) + r;
                const int c8 = 8 * ((lane & 3) ^ ((r >> 2) & 3)), col = n0 + 32 * cb + c8;
                if (p.resp) {
                    const unsigned char* q = p.resp + ((size_t)(row >> 5) * (size_t)(p.N >> 5) + (size_t)(col >> 5)) * CHUNK + h * 1024 + lane * 16;
                    rq[j][0] = *reinterpret_cast<const uint4*>(q);
                    rq[j][1] = *reinterpret_cast<const uint4*>(q + 2048);
                    rq[j][2] = *reinterpret_cast<const uint4*>(q + 4096);
                } else if (p.residual && row < mlim) {
                    const uint4* q = reinterpret_cast<const uint4*>(p.residual + (size_t)row * p.ldc + col);
                    rq[j][0] = q[0];
                    rq[j][1] = q[1];
                }
            }
        }
        stage(i0);
#pragma unroll
        for (int j = 0; j < UPW; ++j) {
            const int u = wave + 8 * j;
            if (u < nunits && m0 + 32 * (i0 + u / (2 * ncb)) < mlim) {
                const int rbi = u / (2 * ncb), cb = (u >> 1) - rbi * ncb, h = u & 1;
                const int r = 16 * h + ur, row = m0 + 32 * (i0 + rbi) + r;
                const int c8 = 8 * ((lane & 3) ^ ((r >> 2) & 3)), col = n0 + 32 * cb + c8;
                const float* src = ct + (32 * rbi + r) * LDC + 32 * cb + c8;
                float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
                if (p.bias) {
                    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col), b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
                    v0.x += b0.x; v0.y += b0.y; v0.z += b0.z; v0.w += b0.w; v1.x += b1.x; v1.y += b1.y; v1.z += b1.z; v1.w += b1.w;
                }
                if (p.resp) {
                    const uint4 hh = rq[j][0], mm = rq[j][1], ll = rq[j][2];
                    v0.x += (bf_lo(hh.x) + bf_lo(mm.x)) + bf_lo(ll.x); v0.y += (bf_hi(hh.x) + bf_hi(mm.x)) + bf_hi(ll.x);
                    v0.z += (bf_lo(hh.y) + bf_lo(mm.y)) + bf_lo(ll.y); v0.w += (bf_hi(hh.y) + bf_hi(mm.y)) + bf_hi(ll.y);
                    v1.x += (bf_lo(hh.z) + bf_lo(mm.z)) + bf_lo(ll.z); v1.y += (bf_hi(hh.z) + bf_hi(mm.z)) + bf_hi(ll.z);
                    v1.z += (bf_lo(hh.w) + bf_lo(mm.w)) + bf_lo(ll.w); v1.w += (bf_hi(hh.w) + bf_hi(mm.w)) + bf_hi(ll.w);
                } else if (p.residual) {
                    const uint4 q0 = rq[j][0], q1 = rq[j][1];
                    v0.x += __uint_as_float(q0.x); v0.y += __uint_as_float(q0.y); v0.z += __uint_as_float(q0.z); v0.w += __uint_as_float(q0.w);
                    v1.x += __uint_as_float(q1.x); v1.y += __uint_as_float(q1.y); v1.z += __uint_as_float(q1.z); v1.w += __uint_as_float(q1.w);
                }
                if (p.relu) {
                    v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                    v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
                }
                if (p.out && row < mlim) {
                    float4* d = reinterpret_cast<float4*>(p.out + (size_t)row * p.ldc + col);
                    d[0] = v0; d[1] = v1;
                }
                unsigned hv[4], mv[4], lv[4];
                split_pair(v0.x, v0.y, hv[0], mv[0], lv[0]); split_pair(v0.z, v0.w, hv[1], mv[1], lv[1]);
                split_pair(v1.x, v1.y, hv[2], mv[2], lv[2]); split_pair(v1.z, v1.w, hv[3], mv[3], lv[3]);
                // rows >= M of the last row block are written too (finite garbage from the clamped A rows): they exist in the planes buffer and
                // only ever feed output rows >= M of the consumer
                unsigned char* q = p.outp + ((size_t)(row >> 5) * (size_t)(p.N >> 5) + (size_t)(col >> 5)) * CHUNK + h * 1024 + lane * 16;
                *reinterpret_cast<uint4*>(q) = make_uint4(hv[0], hv[1], hv[2], hv[3]);
                *reinterpret_cast<uint4*>(q + 2048) = make_uint4(mv[0], mv[1], mv[2], mv[3]);
                *reinterpret_cast<uint4*>(q + 4096) = make_uint4(lv[0], lv[1], lv[2], lv[3]);
            }
        }
    }
}

// workgroup b runs on XCD b % 8: every XCD takes a contiguous run of tiles, so the tiles_n workgroups that share an A row block meet in
// one L2.  Returns the tile id or -1 for a surplus workgroup.
__device__ __forceinline__ int xcd_tile_id(int total) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int q = total >> 3, r = total & 7;
    if (j >= q + (x < r ? 1 : 0)) return -1;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// ---- A in f32: K steps of 32 through a ring of THREE LDS buffers (round 5) ----------------------------------------------------------
// Step j reads buffer j % 3 while the split of step j + 2 is written into buffer (j + 2) % 3 - which nobody has touched since the barrier
// of step j - 1 - so buffer (j + 1) % 3 is complete and visible for the whole of step j: the last slots of a step prefetch the next step's
// fragments ACROSS the barrier, whose wait is a counted lgkmcnt (the LDS writes are older than the MT outstanding prefetch reads).  One
// barrier per 2 x 6 MT MFMAs, no drain.  A rows are staged as float2 per thread (160 rows x 32 floats / 512 threads = 5 float2), three
// ds_write_b32 per row block, and fetched two K steps ahead of their split (a second register set; MT = 6 still fits 256 VGPRs without
// scratch; profiles/r05_split_adeep.txt).  The instruction stream is laid out by hand in slots of MT MFMAs (sched_barrier between slots):
// the three planes of a sub-step's A fragments are consumed lo -> mid -> hi, and the reads of the NEXT sub-step's plane are issued into the
// same registers right behind the last MFMA that used it.  What was measured and not kept (a double-buffered K-64 structure, four-wave
// workgroups with 64-column wave tiles and a second accumulator set, W reloads spread over the slots, priorities, residual prefetch) is in
// DESIGN.md section 4 and profiles/r05_split_*.txt; the code lives in the history of this file (round 5).
template <int MT, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_split_kernel(const SplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 32 * MT;
    constexpr int PLANE = BM * 64;               // bytes of one bf16 plane of a 32-deep K step
    constexpr int BUF = 3 * PLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool CONV = MODE == 1 || MODE == 3;       // A rows are NHWC pixels
    constexpr bool PM = MODE == 3;                       // position-major tiles (see below)
    int id = PM ? (int)blockIdx.x : xcd_tile_id(p.tiles_m * p.tiles_n * p.splitk);
    if (id < 0 || (PM && id >= p.tiles_m * p.tiles_n)) return;
#ifdef WD_DEBUG
    long long t_start = 0, r_start = 0, t_main = 0, t_epi = 0;
    if (p.stamps) { t_start = __builtin_amdgcn_s_memtime(); r_start = __builtin_amdgcn_s_memrealtime(); }
    const int id_stamp = id;
#endif
    // split-K: the slices of one tile are neighbours (same XCD); slice kz covers K steps [k0, k0 + nk)
    const int kz = PM ? 0 : id % p.splitk;
    if (!PM) id /= p.splitk;
    int tm, tn;
    if (p.xmap == 0 || PM) { tm = id / p.tiles_n; tn = id - tm * p.tiles_n; } else { tn = id / p.tiles_m; tm = id - tn * p.tiles_m; }
    int m0 = tm * BM;
    const int n0 = tn * BN;
    const int kc = CONV ? p.C / 32 : 1;           // K steps per tap
    // MODE 3 (round 6): POSITION-MAJOR tiles of a 3x3 / stride 1 / pad 1 convolution over many small maps (the box heads: 1000 ROIs x 7 x 7).  A tile = ONE
    // output position (y, x) of up to BM consecutive maps, so whether a tap falls into the zero padding is the same for every row of the tile and the K
    // steps of such taps are SKIPPED (18 % of the (row, tap) pairs of a 7 x 7 map; the plain row order computes them on zero operands).  Tiles are
    // numbered longest first - the (H-2)(W-2) interior positions (9 taps), then the edges (6), then the corners (4), chunk-major inside a class - and are
    // dispatched in that order without the per-XCD ranges of the other modes, so the short tiles fill the tail.  The remaining taps are summed in the same
    // order as ever: bit-identical results.  m0 = first map of the tile; matrix row of (map b, position) = b * H * W + y * W + x.
    int pm_y = 0, pm_x = 0, pm_nvt = 9;
    unsigned long long pm_taps = 0x876543210ull;  // the valid taps, one nibble each, ascending
    if (PM) {
        const int nch = (p.M / (p.H * p.W) + BM - 1) / BM;                 // map chunks per position (M = maps x H x W)
        const int wi = p.W - 2, hi = p.H - 2, ni = wi * hi, ne = 2 * wi + 2 * hi;
        int t = tm, c, q;
        if (t < ni * nch) { c = t / ni; q = t - c * ni; pm_y = 1 + q / wi; pm_x = 1 + q - (q / wi) * wi; }
        else if (t < (ni + ne) * nch) {
            t -= ni * nch; c = t / ne; q = t - c * ne;
            if (q < wi) { pm_y = 0; pm_x = 1 + q; }
            else if (q < 2 * wi) { pm_y = p.H - 1; pm_x = 1 + q - wi; }
            else if (q < 2 * wi + hi) { pm_x = 0; pm_y = 1 + q - 2 * wi; }
            else { pm_x = p.W - 1; pm_y = 1 + q - 2 * wi - hi; }
        } else { t -= (ni + ne) * nch; c = t >> 2; q = t & 3; pm_y = (q >> 1) ? p.H - 1 : 0; pm_x = (q & 1) ? p.W - 1 : 0; }
        m0 = c * BM;
        pm_taps = 0; pm_nvt = 0;
        for (int tp = 0; tp < 9; ++tp) {
            const int yy = pm_y + tp / 3 - 1, xx = pm_x + tp % 3 - 1;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) { pm_taps |= (unsigned long long)tp << (4 * pm_nvt); ++pm_nvt; }
        }
    }
    auto pm_tap = [&](int i) { return (int)((pm_taps >> (4 * i)) & 15ull); };
    const int nk_all = PM ? pm_nvt * kc : p.K / 32;
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
        if (PM) {
            const int maps = p.M / (p.H * p.W);
            m = m < maps ? m : maps - 1;                 // m = map index
            aoff[i] = (unsigned)(((m * p.H + pm_y - 1) * p.W + pm_x - 1) * p.C + 2 * sk2);
            vmask[i] = 0x1ffu;                           // the taps that are walked are valid for every row
            continue;
        }
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
    float2 araw[MT];
    auto a_fetch = [&](int kt, int i) -> float2 {   // row block i of K step kt (clamped)
        kt = k0 + (kt < nk ? kt : nk - 1);
        if (MODE == 0) {
            return *reinterpret_cast<const float2*>(p.a + (size_t)(aoff[i] + (unsigned)(kt * 32)));
        } else {
            const int ti = kt / kc, cb = kt - ti * kc;
            const int tap = PM ? pm_tap(ti) : ti;
            const int dy = tap / p.ksize, dx = tap - dy * p.ksize;
            const unsigned delta = (unsigned)((dy * p.W + dx) * p.C + cb * 32);
            const bool ok = ((vmask[i] >> tap) & 1u) != 0;
            const unsigned off = ok ? aoff[i] + delta : 2u * sk2;         // always a valid address; zeroed below
            const float2 v = *reinterpret_cast<const float2*>(p.a + (size_t)off);
            return ok ? v : make_float2(0.f, 0.f);
        }
    };
    auto a_load_row = [&](int kt, int i) { araw[i] = a_fetch(kt, i); };
    // MODE 1, main loop: the (tap, channel block) of the K step being fetched is a CURSOR advanced once per iteration - a_fetch divides it out of
    // kt for every row block (217 scalar instructions per wave and K step in the box-head convolution, as many as its vector instructions:
    // profiles/r05_split_pmc_conv.txt, r05_split_conv_cursor.txt)
    int f_idx = 0, f_tap = 0, f_cb = 0;
    unsigned f_delta = 0;
    int f_act = 0;                                  // the tap f_tap stands for (MODE 3: f_tap counts the WALKED taps)
    auto f_place = [&]() {
        f_act = PM ? pm_tap(f_tap) : f_tap;
        const int dy = p.ksize == 3 ? (f_act * 11) >> 5 : 0, dx = f_act - dy * p.ksize;      // tap / 3 for tap < 9
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
        const bool ok = ((vmask[i] >> f_act) & 1u) != 0;
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
        if (PM) {                                   // walked K step -> the weight's K step: (walked tap index, rest) -> (tap, rest)
            const int ti = sub / (2 * kc);
            sub = pm_tap(ti) * 2 * kc + (sub - ti * 2 * kc);
        }
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
    float2 aahead[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) aahead[i] = a_fetch(3, i);
#ifdef WD_DEBUG
    if (p.stamps) t_main = __builtin_amdgcn_s_memtime();      // (an SMEM op: kept in front of the lgkmcnt(0) below, see the loop's counted waits)
#endif
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();

    bf16x8 af[MT][3];
#define SB __builtin_amdgcn_sched_barrier(0)
#define MF(pa, pb, slot)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], wf[slot][pb], acc[i], 0, 0, 0);
#define RD(addr, pl)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
        af[i][pl] = *reinterpret_cast<const bf16x8*>(smem + (addr) + (pl) * PLANE + i * 2048);
    // A rows two K steps ahead (a second register set): araw = step kt + 2 (split now), aahead = step kt + 3, reload with step kt + 4
#define SPLIT_ROW(i_)                                                                                         \
    if ((i_) < MT) { a_store_row(wr, (i_)); araw[(i_)] = aahead[(i_)]; aahead[(i_)] = CONV ? a_fetch_cur((i_)) : a_fetch(kt + 4, (i_)); }
    RD(rofs0, 2) RD(rofs0, 1) RD(rofs0, 0)
    SB;
    int cur = 0, nxt = BUF, wr = 2 * BUF;
    if (CONV) f_set(4);
    for (int kt = 0; kt < nk; ++kt) {
        const int a1 = cur + rofs1, a0n = nxt + rofs0;
        // sub-step 0 (fragments in registers); its slots request sub-step 1's fragments of the same buffer.  The W fragments of a sub-step are
        // reloaded right behind its last MFMA, a whole K step ahead of their next use (measured: spreading the eight waves' reloads over
        // different slots of the other sub-step - no burst in the vector-memory path, but 2-5 slots of lead - costs 9 % of the main loop)
        MF(2, 0, 0) RD(a1, 2) SB;
        MF(1, 1, 0) SPLIT_ROW(0) SB;
        MF(1, 0, 0) RD(a1, 1) SB;
        MF(0, 2, 0) SPLIT_ROW(1) SB;
        MF(0, 1, 0) SPLIT_ROW(2) SB;
        MF(0, 0, 0) RD(a1, 0) w_load(2 * kt + 2, 0); SB;
        // sub-step 1; its slots request sub-step 0 of the NEXT K step's buffer (complete since the previous barrier)
        MF(2, 0, 1) RD(a0n, 2) SB;
        MF(1, 1, 1) SPLIT_ROW(3) SB;
        MF(1, 0, 1) RD(a0n, 1) SB;
        MF(0, 2, 1) SPLIT_ROW(4) SB;
        MF(0, 1, 1) SPLIT_ROW(5) SB;
        MF(0, 0, 1) RD(a0n, 0) w_load(2 * kt + 3, 1); SB;
        // this wave's LDS writes (older than the MT prefetch reads just issued) are done; the prefetch stays in flight across the barrier
        __builtin_amdgcn_s_waitcnt(0xC07F | (MT << 8));
        __builtin_amdgcn_s_barrier();
        SB;
        const int t = cur; cur = nxt; nxt = wr; wr = t;
        if (CONV) f_advance(kt + 5);              // the next iteration fetches K step kt + 5
    }
#undef SPLIT_ROW
#ifdef WD_DEBUG
    if (p.stamps) t_epi = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_waitcnt(0xC07F);          // the dangling prefetch of the step behind the last one
    __builtin_amdgcn_s_barrier();
    if (p.splitk > 1) {                            // raw partial tile of this K slice; bias / residual / ReLU in the reduce launch
        SplitArgs q = p;
        q.out = p.part + (size_t)kz * p.M * p.N;
        q.ldc = p.N;
        q.bias = nullptr; q.residual = nullptr; q.resp = nullptr; q.outp = nullptr; q.relu = 0;
        split_epilogue<MT>(q, smem, acc, m0, n0, wave, lane, p.M);
    } else if (PM) {
        split_epilogue<MT>(p, smem, acc, m0, n0, wave, lane, p.M / (p.H * p.W), p.H * p.W, pm_y * p.W + pm_x);
    } else {
        split_epilogue<MT>(p, smem, acc, m0, n0, wave, lane, p.M);
    }
#ifdef WD_DEBUG
    if (p.stamps && tid == 0) {
        long long* o = p.stamps + 8 * (long)blockIdx.x;
        o[0] = t_start; o[1] = t_main; o[2] = t_epi; o[3] = __builtin_amdgcn_s_memtime();
        o[4] = id_stamp; o[5] = __builtin_amdgcn_s_getreg(20 /* XCC_ID */ | (0 << 6) | (3 << 11));
        o[6] = r_start; o[7] = __builtin_amdgcn_s_memrealtime();            // constant 100 MHz clock, common to the chip
    }
#endif
}

#undef MF
#undef RD

// ---- round 6: A arrives pre-split (activation planes), pulled by LDS-DMA into a ring of FOUR buffers --------------------------------------
// The producer (this kernel's own epilogue, the deformable-conv epilogue, wd_split_planes_pack_f32) stores every (32 rows x 32 k) block of the
// activation matrix as the 6 KiB LDS image the fragment reads expect (3 planes x 32 rows x 64 bytes, slots swizzled).  A K step of the tile is
// then 6 MT pieces of 1 KiB, each ONE global_load_lds_dwordx4 wave instruction (64 lanes x 16 bytes, contiguous on both sides); the eight waves
// share them (ceil(6 MT / 8) each; surplus instructions repeat the last piece - same bytes to the same place).  No staging registers, no split
// VALU, no ds_write: what the round-5 ablations priced at 20 % (split + LDS writes) + 11 % (A loads through registers) of the f32-A kernel, and
// what its four N tiles per row block did four times over.
//
// Ordering.  vmcnt counts a wave's vector-memory operations in issue order and loads return in order, so the LDS-DMA and the W fragment loads
// share one queue.  hipcc does not know about the inline-asm DMA and would wait for it with every W fragment; therefore the W loads are inline
// asm too and every wait is written out:
//     iteration kt :  slot 1   s_waitcnt vmcnt(ND + 3)   W sub-step 0 of step kt is here (younger: the ND DMA pieces of step kt + 2, W sub-step 1)
//                     slot 6   W sub-step 0 of step kt + 1  (3 loads)
//                     slot 7   s_waitcnt vmcnt(3)        W sub-step 1 of step kt is here - and with it every OLDER operation: the DMA of step
//                                                        kt + 2, issued in iteration kt - 1, has landed in LDS
//                     slots 8, 10, 11   the ND DMA pieces of step kt + 3 into buffer (kt + 3) % 4 (last read in iteration kt - 1)
//                     slot 12  W sub-step 1 of step kt + 1  (3 loads)
//                     s_barrier                          buffer (kt + 2) % 4 is visible to everybody: iteration kt + 1 prefetches its first
//                                                        fragments in its last slots, as the f32-A kernel does
// A DMA piece has from slot 8 of iteration kt to slot 7 of iteration kt + 1 to arrive (one full K step, ~2 us at MT = 5), and no MFMA wave ever
// waits on LDS writes: the barrier needs no lgkmcnt wait (the prefetch reads stay in flight across it).
template <int MT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_split_planes_kernel(const SplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 32 * MT;
    constexpr int PLANE = BM * 64;
    constexpr int BUF = 3 * PLANE;
    constexpr int NPIECE = 6 * MT;                // 1-KiB pieces per K step: (row block i, plane pl, half h) -> piece (3 i + pl) * 2 + h
    constexpr int ND = (NPIECE + 7) / 8;          // per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int id = xcd_tile_id(p.tiles_m * p.tiles_n * p.splitk);
    if (id < 0) return;
#ifdef WD_DEBUG
    long long t_start = 0, r_start = 0, t_main = 0, t_epi = 0;
    if (p.stamps) { t_start = __builtin_amdgcn_s_memtime(); r_start = __builtin_amdgcn_s_memrealtime(); }
    const int id_stamp = id;
#endif
    const int kz = id % p.splitk;
    id /= p.splitk;
    int tm, tn;
    if (p.xmap == 0) { tm = id / p.tiles_n; tn = id - tm * p.tiles_n; } else { tn = id / p.tiles_m; tm = id - tn * p.tiles_m; }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk_all = p.K / 32;
    const int per = (nk_all + p.splitk - 1) / p.splitk;
    const int k0 = kz * per;
    const int nk = (k0 + per <= nk_all) ? per : nk_all - k0;

    // ---- this wave's DMA pieces (wave-uniform: scalar registers) ----
    const int nrb = (p.M + 31) >> 5;              // row blocks of the planes matrix (the last one may hold rows >= M: never stored)
    const size_t rb_stride = (size_t)nk_all * CHUNK;
    const unsigned char* dsrc[ND];                // source of the piece at K step k0
    unsigned ddst[ND];                            // LDS offset of the piece inside a buffer
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        int c = wave + 8 * j;
        c = c < NPIECE ? c : NPIECE - 1;
        const int i = c / 6, pl = (c - 6 * i) >> 1, h = c & 1;
        int rb = tm * MT + i;
        rb = rb < nrb ? rb : nrb - 1;
        dsrc[j] = p.ap + (size_t)rb * rb_stride + (size_t)k0 * CHUNK + pl * 2048 + h * 1024;
        ddst[j] = (unsigned)(pl * PLANE + i * 2048 + h * 1024);
    }
    const unsigned lane16 = (unsigned)lane * 16u;
    // one LDS-DMA instruction: 64 lanes x 16 bytes from sbase + 16 lane to LDS byte offset dst + 16 lane (M0 carries dst; the dynamic LDS
    // segment starts at LDS address 0: this kernel has no static __shared__)
    auto dma16 = [&](const unsigned char* sbase, unsigned dst) {
        unsigned keep;                         // M0 is a reserved register for hipcc: saved and restored around the instruction
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane16), "s"(sbase), "s"(dst) : "memory");
    };
    auto dma_piece = [&](int kt, unsigned bufoff, int j) {      // piece j of K step kt (clamped: past the end the last step is re-read into a free buffer)
        kt = kt < nk ? kt : nk - 1;
        dma16(dsrc[j] + (size_t)kt * CHUNK, bufoff + ddst[j]);
    };

    // ---- W fragments: this wave's 32 columns; one K step (two sub-steps) in registers, loaded by inline asm (see "Ordering") ----
    const int nt32 = (n0 >> 5) + wave;
    const bool active = nt32 * 32 < p.N;
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w + (size_t)(active ? nt32 : 0) * (size_t)(p.K / 16) * 192);
    const int nsub = nk * 2;
    u32x4 wf[2][3];
    auto w_load = [&](int sub, int slot) {
        sub = 2 * k0 + (sub < nsub ? sub : nsub - 1);
        const unsigned char* q = wbase + (size_t)sub * 3072;
        asm volatile("global_load_dwordx4 %0, %3, %4\n\tglobal_load_dwordx4 %1, %3, %4 offset:1024\n\tglobal_load_dwordx4 %2, %3, %4 offset:2048"
                     : "=&v"(wf[slot][0]), "=&v"(wf[slot][1]), "=&v"(wf[slot][2]) : "v"(lane16), "s"(q) : "memory");
    };
    // the wait that makes a W sub-step usable: n = vector-memory operations issued after its three loads
#define WAITW(slot, n) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]), "+v"(wf[slot][2]) : "n"(n) : "memory")

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    const int rr = lane & 31, rg = lane >> 5;
    const int rofs0 = rr * 64 + ((((0 + rg) ^ ((rr >> 2) & 3))) << 4);
    const int rofs1 = rr * 64 + ((((2 + rg) ^ ((rr >> 2) & 3))) << 4);

    // ---- prologue: K steps 0, 1, 2 -> buffers 0, 1, 2; both W sub-steps of step 0 ----
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < ND; ++j) dma_piece(s, (unsigned)(s * BUF), j);
    w_load(0, 0);
    w_load(1, 1);
#ifdef WD_DEBUG
    if (p.stamps) t_main = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[0][2]), "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[1][2]) :: "memory");
    __builtin_amdgcn_s_barrier();

    bf16x8 af[MT][3];
#define SB __builtin_amdgcn_sched_barrier(0)
#define MF(pa, pb, slot)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], __builtin_bit_cast(bf16x8, wf[slot][pb]), acc[i], 0, 0, 0);
#define RD(addr, pl)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
        af[i][pl] = *reinterpret_cast<const bf16x8*>(smem + (addr) + (pl) * PLANE + i * 2048);
#define DMA(j_)                                                                                               \
    if ((j_) < ND) { dma_piece(kt + 3, (unsigned)wr, (j_)); }
    RD(rofs0, 2) RD(rofs0, 1) RD(rofs0, 0)
    SB;
    int cur = 0, nxt = BUF, nx2 = 2 * BUF, wr = 3 * BUF;
    for (int kt = 0; kt < nk; ++kt) {
        const int a1 = cur + rofs1, a0n = nxt + rofs0;
        WAITW(0, ND + 3);
        MF(2, 0, 0) RD(a1, 2) SB;
        MF(1, 1, 0) SB;
        MF(1, 0, 0) RD(a1, 1) SB;
        MF(0, 2, 0) SB;
        MF(0, 1, 0) SB;
        MF(0, 0, 0) RD(a1, 0) w_load(2 * kt + 2, 0); SB;
        WAITW(1, 3);
        MF(2, 0, 1) RD(a0n, 2) SB;
        MF(1, 1, 1) DMA(0) DMA(3) SB;
        MF(1, 0, 1) RD(a0n, 1) SB;
        MF(0, 2, 1) DMA(1) DMA(4) SB;
        MF(0, 1, 1) DMA(2) SB;
        MF(0, 0, 1) RD(a0n, 0) w_load(2 * kt + 3, 1); SB;
        __builtin_amdgcn_s_barrier();
        SB;
        const int t = cur; cur = nxt; nxt = nx2; nx2 = wr; wr = t;
    }
#undef DMA
#undef WAITW
#ifdef WD_DEBUG
    if (p.stamps) t_epi = __builtin_amdgcn_s_memtime();
#endif
    // the dangling fragment prefetch and the re-read pieces of the last iterations (they land in free ring buffers - which the epilogue reuses)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.splitk > 1) {
        SplitArgs q = p;
        q.out = p.part + (size_t)kz * p.M * p.N;
        q.ldc = p.N;
        q.bias = nullptr; q.residual = nullptr; q.resp = nullptr; q.outp = nullptr; q.relu = 0;
        split_epilogue<MT>(q, smem, acc, m0, n0, wave, lane, p.M);
    } else {
        split_epilogue<MT>(p, smem, acc, m0, n0, wave, lane, p.M);
    }
#ifdef WD_DEBUG
    if (p.stamps && tid == 0) {
        long long* o = p.stamps + 8 * (long)blockIdx.x;
        o[0] = t_start; o[1] = t_main; o[2] = t_epi; o[3] = __builtin_amdgcn_s_memtime();
        o[4] = id_stamp; o[5] = __builtin_amdgcn_s_getreg(20 /* XCC_ID */ | (0 << 6) | (3 << 11));
        o[6] = r_start; o[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}
#undef SB
#undef MF
#undef RD

// f32 activations (M, K; row stride lda) -> activation planes: the stand-alone producer (inputs that no epilogue of ours wrote, tests, tools).
// One thread per (row, 8 consecutive k): two float4 in, one 16-byte slot per plane out; rows >= M of the last row block are zero-filled.
__global__ __launch_bounds__(256) void split_planes_pack_kernel(const float* __restrict__ a, long lda, int M, int K, unsigned char* __restrict__ out, long total) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int k8 = K / 8;
    const int row = (int)(t / k8), kq = (int)(t - (long)row * k8);
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (row < M) {
        const float4* s = reinterpret_cast<const float4*>(a + (size_t)row * lda + 8 * kq);
        v0 = s[0]; v1 = s[1];
    }
    unsigned h[4], m[4], l[4];
    split_pair(v0.x, v0.y, h[0], m[0], l[0]); split_pair(v0.z, v0.w, h[1], m[1], l[1]);
    split_pair(v1.x, v1.y, h[2], m[2], l[2]); split_pair(v1.z, v1.w, h[3], m[3], l[3]);
    unsigned char* q = out + planes_offset(row, 8 * kq, K);
    *reinterpret_cast<uint4*>(q) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(q + 2048) = make_uint4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<uint4*>(q + 4096) = make_uint4(l[0], l[1], l[2], l[3]);
}

// activation planes -> f32 (M, K; row stride ldo): exact (hi + mid + lo); for consumers outside this unit and for the tests
__global__ __launch_bounds__(256) void split_planes_unpack_kernel(const unsigned char* __restrict__ planes, int M, int K, float* __restrict__ out, long ldo, long total) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int k8 = K / 8;
    const int row = (int)(t / k8), kq = (int)(t - (long)row * k8);
    const unsigned char* q = planes + planes_offset(row, 8 * kq, K);
    const uint4 h = *reinterpret_cast<const uint4*>(q), m = *reinterpret_cast<const uint4*>(q + 2048), l = *reinterpret_cast<const uint4*>(q + 4096);
    float4* d = reinterpret_cast<float4*>(out + (size_t)row * ldo + 8 * kq);
    d[0] = make_float4((bf_lo(h.x) + bf_lo(m.x)) + bf_lo(l.x), (bf_hi(h.x) + bf_hi(m.x)) + bf_hi(l.x), (bf_lo(h.y) + bf_lo(m.y)) + bf_lo(l.y),
                       (bf_hi(h.y) + bf_hi(m.y)) + bf_hi(l.y));
    d[1] = make_float4((bf_lo(h.z) + bf_lo(m.z)) + bf_lo(l.z), (bf_hi(h.z) + bf_hi(m.z)) + bf_hi(l.z), (bf_lo(h.w) + bf_lo(m.w)) + bf_lo(l.w),
                       (bf_hi(h.w) + bf_hi(m.w)) + bf_hi(l.w));
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

// out = act(sum over the K slices in slice order + bias + residual): deterministic, one float4 per thread; residual / output as f32 and / or
// as activation planes, like the tile epilogue.  (Round 6 tried the one-launch form - every slice counts itself in on a per-stream counter, the last one
// to arrive sums the tile - and measured it SLOWER: training step 71.2 -> 83.3 ms, e2e 42.3 -> 39.9 frames/s, bit-identical results.  The partial tiles
// must be visible across XCDs, and the device-scope release / acquire fences that takes (__threadfence: an L2 write-back per workgroup) cost far more
// than the 10 us launch they save; profiles/r06_split_presplit.txt, last section.)
__global__ __launch_bounds__(256) void gemm_split_reduce_kernel(const float4* __restrict__ part, int splitk, long mn4, int n4, const float4* __restrict__ bias,
                                                                const float* __restrict__ residual, const unsigned char* __restrict__ resp, long ldc, int relu,
                                                                float* __restrict__ out, unsigned char* __restrict__ outp) {
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
    if (resp) {
        const unsigned char* q = resp + planes_offset((int)row, 4 * c4, 4 * n4);
        const uint2 h = *reinterpret_cast<const uint2*>(q), m = *reinterpret_cast<const uint2*>(q + 2048), l = *reinterpret_cast<const uint2*>(q + 4096);
        v.x += (bf_lo(h.x) + bf_lo(m.x)) + bf_lo(l.x); v.y += (bf_hi(h.x) + bf_hi(m.x)) + bf_hi(l.x);
        v.z += (bf_lo(h.y) + bf_lo(m.y)) + bf_lo(l.y); v.w += (bf_hi(h.y) + bf_hi(m.y)) + bf_hi(l.y);
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (out) *reinterpret_cast<float4*>(out + o) = v;
    if (outp) {
        unsigned h0, m0_, l0, h1, m1, l1;
        split_pair(v.x, v.y, h0, m0_, l0);
        split_pair(v.z, v.w, h1, m1, l1);
        unsigned char* q = outp + planes_offset((int)row, 4 * c4, 4 * n4);
        *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(q + 2048) = make_uint2(m0_, m1);
        *reinterpret_cast<uint2*>(q + 4096) = make_uint2(l0, l1);
    }
}

// Laboratory switches of this unit (tile height, K slices, tile order) are honoured only when WT_EXPERIMENT=1 is set as well: a product
// process ignores them - with a warning, because WD_SPLIT_MT=2|3 re-enables a known wrong-answer mode of OTHER kernels (see pick_plan).
bool experiment_mode() {
    static const bool on = []() { const char* e = getenv("WT_EXPERIMENT"); return e && e[0] == '1'; }();
    return on;
}
int experiment_knob(const char* name) {
    const char* e = getenv(name);
    if (!e) return 0;
    if (!experiment_mode()) {
        fprintf(stderr, "libwaymotrack: %s=%s ignored (laboratory switch; set WT_EXPERIMENT=1 to use it)\n", name, e);
        return 0;
    }
    return atoi(e);
}

// Tile height (32 MT rows) and K slices for a shape.  Cost model in microseconds from the measured ring kernel (MI355X, profiles/r05_split_*):
// a K step of 32 costs ~0.52 us per 32-row block of the tile, prologue + epilogue ~(2 + MT) us per workgroup, 256 workgroups run at once;
// a split adds the reduce launch (3 us + the partial tiles through HBM at ~4 TB/s).  Shapes with few tiles (FPN p5 / p6, the box-head FC)
// fill the chip through K slices; large ones pick the tile height with the fewest idle CU-rounds.  (Round 6, two frames in flight: pricing a launch by
// its work alone - "the other frame fills the CUs a partial round leaves idle" - picks smaller tiles / fewer slices and LOSES 7 %: 41.3 -> 38.5 frames/s,
// two alternating runs on one box; the rounds stay in the model.)
struct Plan { int mt, splitk; };

// Tile heights of 2 and 3 row blocks (64 / 96 rows) are NOT offered.  While waves that issue back-to-back v_mfma_f32_32x32x16_bf16 share a SIMD
// with a wave of deform_conv3x3_kernel<64, true> (230 VGPRs) or grouped_conv3x3_c8_kernel (256 VGPRs), those two kernels return hundreds of
// percent-level wrong outputs per launch (profiles/r05_costream_interference.txt, profiles/r06_costream_victim_side.txt: reproduced with a
// pure-register matrix-instruction burner on ZERO operands, so neither data nor power; the victims' LDS contents verified intact; no victim-side
// wait / barrier / M0 / occupancy variant cures it; every other product kernel is clean next to every tile height).  With >= 4 row blocks a
// workgroup's two waves per SIMD hold >= 2 x 184 registers: no wave of >= 230 registers fits beside them - launch<MT>() asserts that on the
// compiled kernels (occupancy_guard).
constexpr int MT_MIN = 4, MT_MAX = 6;

Plan pick_plan(long M, int N, int K, bool allow_split) {
    static const int forced_mt = experiment_knob("WD_SPLIT_MT");
    static const int forced_sk = experiment_knob("WD_SPLIT_SPLITK");
    const long tn = (N + BN - 1) / BN;
    const int nk_all = K / 32;
    Plan best{5, 1};
    double best_t = 1e30;
    // measured cost per row block and K step relative to MT = 4 / 5 (box-head conv 49000 x 256 x 2304, tools/conv_split_one.py): MT = 3 and MT = 6 +24 %
    const int mt_min = (forced_mt == 2 || forced_mt == 3) ? forced_mt : MT_MIN;
    for (int mt = MT_MAX; mt >= mt_min; --mt) {
        if (forced_mt >= 2 && forced_mt <= MT_MAX && mt != forced_mt) continue;
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

#ifdef WD_DEBUG
long long* g_stamps = nullptr;           // diagnostics (wd_gemm_split_debug_stamps)
#endif

// The co-residency rule of pick_plan, checked on what was actually compiled: a workgroup of this kernel puts two waves on every SIMD; a wave
// of the two kernels known to return wrong results beside bf16 matrix waves (their register counts are read from the compiled kernels too) must not fit
// into the 512-register file next to them.  VGPRs are allocated in blocks of 8.
int occupancy_guard(const void* fn, int mt) {
    hipFuncAttributes at{};
    WT_HIP(hipFuncGetAttributes(&at, fn));
    const int alloc = (at.numRegs + 7) / 8 * 8;
    // the smaller of the two vulnerable kernels AS COMPILED (today 232 and 256 allocated registers); 232 when the query fails
    int victim = 232;
    const int g = wt::victim_regs_grouped_conv(), d = wt::victim_regs_deform64();
    if (g > 0 && d > 0) victim = g < d ? g : d;
    if (2 * alloc + victim <= 512 && !experiment_mode()) {
        wt::set_error("split-operand kernel with %d row blocks was compiled to %d registers: a %d-register wave of deform_conv3x3_kernel<64> / "
                      "grouped_conv3x3_c8_kernel fits beside two of its waves on a SIMD, which is the co-residency that corrupts those kernels "
                      "(profiles/r06_costream_victim_side.txt); refused (WT_EXPERIMENT=1 overrides)", mt, at.numRegs, victim);
        return WT_ERR_INVALID;
    }
    return WT_OK;
}

// MODE 0 / 1: f32 activations (plain / convolution source); MODE 2: activation planes
template <int MT, int MODE>
int launch(const SplitArgs& a, hipStream_t stream) {
    constexpr size_t lds_epi = 32u * (MT < 3 ? MT : 3) * LDC * 4u;
    constexpr size_t lds_ring = (MODE == 2 ? 4u : 3u) * 3u * 32u * MT * 64u;
    constexpr size_t lds = lds_ring > lds_epi ? lds_ring : lds_epi;
    const void* fn = MODE == 2 ? reinterpret_cast<const void*>(gemm_split_planes_kernel<MT>)
                               : reinterpret_cast<const void*>(gemm_split_kernel<MT, MODE == 2 ? 0 : MODE>);      // MODE 3: position-major convolution tiles
    static wt::OncePerDevice attr;
    const int dev = wt::device_index();
    if (attr.needed(dev)) {
        WT_TRY(occupancy_guard(fn, MT));
        WT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.mark(dev);
    }
    const int total = a.tiles_m * a.tiles_n * a.splitk;
    const dim3 grid((unsigned)((total + 7) / 8 * 8));
    if constexpr (MODE == 2) hipLaunchKernelGGL((gemm_split_planes_kernel<MT>), grid, dim3(NTHREADS), lds, stream, a);
    else hipLaunchKernelGGL((gemm_split_kernel<MT, MODE>), grid, dim3(NTHREADS), lds, stream, a);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

template <int MODE>
int dispatch(SplitArgs& a, void* workspace, size_t workspace_bytes, hipStream_t stream) {
#ifdef WD_DEBUG
    a.stamps = g_stamps;
#endif
    const bool can_split = workspace != nullptr && (a.N % 4) == 0;
    Plan pl = pick_plan(a.M, a.N, a.K, can_split);
    if (pl.splitk > 1 && workspace_bytes < (size_t)pl.splitk * a.M * a.N * sizeof(float)) pl = pick_plan(a.M, a.N, a.K, false);
    const int mt = pl.mt;
    a.splitk = pl.splitk;
    a.part = pl.splitk > 1 ? (float*)workspace : nullptr;
    a.tiles_m = (a.M + 32 * mt - 1) / (32 * mt);
    a.tiles_n = (a.N + BN - 1) / BN;
    static const int xmap = experiment_knob("WD_SPLIT_XMAP");
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
                       (const float4*)a.bias, a.residual, a.resp, a.ldc, a.relu, a.out, a.outp);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

// Position-major tiles (MODE 3) for a 3x3 / stride 1 / pad 1 convolution over many small maps: tile height from the same cost model, summed over the three
// position classes (9 / 6 / 4 walked taps) and spread over the chip (the launch is several waves of unequal tiles, longest first).
int dispatch_position_major(SplitArgs& a, hipStream_t stream) {
#ifdef WD_DEBUG
    a.stamps = g_stamps;
#endif
    const int maps = a.M / (a.H * a.W), kc = a.C / 32;
    const long tn = (a.N + BN - 1) / BN;
    const long ni = (long)(a.H - 2) * (a.W - 2), ne = 2l * (a.W - 2) + 2l * (a.H - 2);
    int best = 4;
    double best_t = 1e30;
    static const int forced_mt = experiment_knob("WD_SPLIT_MT");
    for (int mt = MT_MAX; mt >= MT_MIN; --mt) {
        if (forced_mt >= MT_MIN && forced_mt <= MT_MAX && mt != forced_mt) continue;
        const long nch = (maps + 32 * mt - 1) / (32 * mt);
        const double eff = (mt == 4 || mt == 5) ? 1.0 : 1.24;
        auto tile = [&](int taps) { return mt * taps * kc * 0.52 * eff + 2.0 + mt; };
        const double work = (double)nch * tn * (ni * tile(9) + ne * tile(6) + 4 * tile(4)) / 256.0;
        const double t = work > tile(9) ? work : tile(9);
        if (t < best_t * 0.97) { best_t = t; best = mt; }
    }
    a.splitk = 1; a.part = nullptr; a.xmap = 0;
    a.tiles_m = a.H * a.W * ((maps + 32 * best - 1) / (32 * best));
    a.tiles_n = (int)tn;
    switch (best) {
        case 4: return launch<4, 3>(a, stream);
        case 6: return launch<6, 3>(a, stream);
        default: return launch<5, 3>(a, stream);
    }
}

bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace

extern "C" {

#ifdef WD_DEBUG
/* Diagnostics (debug library only, csrc/debug/waymodet_debug.h): following launches write eight int64 per workgroup (s_memtime at start / main loop /
 * epilogue / end, tile id, XCC id, s_memrealtime at start / end) to `buf` (device memory, 8 x grid size entries; nullptr switches it off). */
int wd_gemm_split_debug_stamps(long long* buf) {
    g_stamps = buf;
    return WT_OK;
}
#endif

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
    if (M <= 0 || N <= 0 || K <= 0 || (K % BK) || (N % 32)) return 0;
    const Plan pl = pick_plan(M, N, K, true);
    return pl.splitk > 1 ? (size_t)pl.splitk * M * N * sizeof(float) : 0;
}

/* 1 when the shape / pointers satisfy wd_gemm_split_f32's preconditions (host code asks before it routes a GEMM here) */
int wd_gemm_split_supported(const float* a, long lda, const float* bias, const float* residual, const float* out, long ldc, long M, int N, int K) {
    return M > 0 && N > 0 && K > 0 && (K % BK) == 0 && (N % 32) == 0 && (lda & 3) == 0 && (ldc & 3) == 0 && lda >= K && ldc >= N && !misaligned(a) &&
           !misaligned(bias) && !misaligned(residual) && !misaligned(out) && M < (1l << 31) && M * lda < (1l << 31);
}

int wd_gemm_split_f32(const float* a, long lda, const void* packed_w, const float* bias, const float* residual, float* out, long ldc,
                      int M, int N, int K, int relu, void* workspace, size_t workspace_bytes, void* stream_) {
    WT_TRY(wt::ensure_device());
    if (M <= 0 || N <= 0) return WT_OK;
    if (!a || !packed_w || !out || misaligned(packed_w) || !wd_gemm_split_supported(a, lda, bias, residual, out, ldc, M, N, K)) {
        wt::set_error("wd_gemm_split_f32: needs K %% %d == 0, N %% 32 == 0, 16-byte aligned rows and M * lda < 2^31 (M=%d N=%d K=%d lda=%ld)", BK, M, N,
                      K, lda);
        return WT_ERR_INVALID;
    }
    SplitArgs s{};
    s.a = a; s.w = (const uint4*)packed_w; s.bias = bias; s.residual = residual; s.out = out;
    s.lda = lda; s.ldc = ldc; s.M = M; s.N = N; s.K = K; s.relu = relu;
    return dispatch<0>(s, workspace, workspace_bytes, (hipStream_t)stream_);
}

size_t wd_split_planes_bytes(long M, int K) {
    if (M <= 0 || K <= 0 || (K % 32)) return 0;
    return (size_t)((M + 31) / 32) * (size_t)(K / 32) * CHUNK;
}

int wd_split_planes_pack_f32(const float* a, long lda, long M, int K, void* planes, void* stream_) {
    WT_TRY(wt::ensure_device());
    if (!a || !planes || M <= 0 || M >= (1l << 31) || K <= 0 || (K % 32) || (lda & 3) || lda < K || misaligned(a) || misaligned(planes)) {
        wt::set_error("wd_split_planes_pack_f32: needs K %% 32 == 0 and 16-byte aligned rows (M=%ld K=%d lda=%ld)", M, K, lda);
        return WT_ERR_INVALID;
    }
    const long total = ((M + 31) / 32) * 32 * (K / 8);
    hipLaunchKernelGGL(split_planes_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, a, lda, (int)M, K,
                       (unsigned char*)planes, total);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

int wd_split_planes_unpack_f32(const void* planes, long M, int K, float* out, long ldo, void* stream_) {
    WT_TRY(wt::ensure_device());
    if (!out || !planes || M <= 0 || M >= (1l << 31) || K <= 0 || (K % 32) || (ldo & 3) || ldo < K || misaligned(out) || misaligned(planes)) {
        wt::set_error("wd_split_planes_unpack_f32: needs K %% 32 == 0 and 16-byte aligned rows (M=%ld K=%d ldo=%ld)", M, K, ldo);
        return WT_ERR_INVALID;
    }
    const long total = M * (K / 8);
    hipLaunchKernelGGL(split_planes_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, (const unsigned char*)planes,
                       (int)M, K, out, ldo, total);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

int wd_gemm_split_io(const WdSplitIO* io, const void* packed_w, const float* bias, int M, int N, int K, int relu, void* workspace,
                     size_t workspace_bytes, void* stream_) {
    WT_TRY(wt::ensure_device());
    if (M <= 0 || N <= 0) return WT_OK;
    if (!io || !packed_w || misaligned(packed_w) || (!io->a && !io->a_planes) || (io->a && io->a_planes) || (!io->out && !io->out_planes) ||
        (io->residual && io->residual_planes) || K <= 0 || (K % BK) || (N % 32) || misaligned(io->a_planes) || misaligned(io->residual_planes) ||
        misaligned(io->out_planes) ||
        !wd_gemm_split_supported(io->a ? io->a : (const float*)packed_w, io->a ? io->lda : K, bias, io->residual, io->out, io->ldc ? io->ldc : N, M, N, K)) {
        wt::set_error("wd_gemm_split_io: exactly one of a / a_planes, at least one of out / out_planes, at most one residual form; K %% %d == 0, "
                      "N %% 32 == 0, 16-byte aligned pointers and rows (M=%d N=%d K=%d)", BK, M, N, K);
        return WT_ERR_INVALID;
    }
    SplitArgs s{};
    s.a = io->a; s.ap = (const unsigned char*)io->a_planes; s.w = (const uint4*)packed_w; s.bias = bias;
    s.residual = io->residual; s.resp = (const unsigned char*)io->residual_planes; s.out = io->out; s.outp = (unsigned char*)io->out_planes;
    s.lda = io->a ? io->lda : K; s.ldc = io->ldc ? io->ldc : N; s.M = M; s.N = N; s.K = K; s.relu = relu;
    return io->a ? dispatch<0>(s, workspace, workspace_bytes, (hipStream_t)stream_) : dispatch<2>(s, workspace, workspace_bytes, (hipStream_t)stream_);
}

int wd_conv_split_f32(const float* x, int batch, int H, int W, int C, const void* packed_w, int ksize, int stride, int pad, const float* bias,
                      const float* residual, float* out, int N, int relu, void* workspace, size_t workspace_bytes, void* stream_) {
    WT_TRY(wt::ensure_device());
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const long M = (long)batch * Ho * Wo;
    if (M <= 0 || N <= 0) return WT_OK;
    if (!x || !packed_w || !out || (ksize != 1 && ksize != 3) || stride < 1 || pad < 0 || C <= 0 || (C % BK) || (N % 32) || misaligned(x) ||
        misaligned(out) || misaligned(bias) || misaligned(residual) || (long)batch * H * W * C >= (1l << 31) || M >= (1l << 31)) {
        wt::set_error("wd_conv_split_f32: needs ksize 1 or 3, C %% %d == 0, N %% 32 == 0 and fewer than 2^31 input elements (C=%d N=%d k=%d)", BK, C, N, ksize);
        return WT_ERR_INVALID;
    }
    SplitArgs s{};
    s.a = x; s.w = (const uint4*)packed_w; s.bias = bias; s.residual = residual; s.out = out;
    s.lda = C; s.ldc = N; s.M = (int)M; s.N = N; s.K = ksize * ksize * C; s.relu = relu;
    s.H = H; s.W = W; s.C = C; s.Ho = Ho; s.Wo = Wo; s.stride = stride; s.pad = pad; s.ksize = ksize;
    // many small maps (the box heads: 1000 ROIs x 7 x 7): position-major tiles skip the taps that fall into the zero padding (MODE 3)
    static const int no_pm = experiment_knob("WD_SPLIT_NO_POSMAJOR");
    if (!no_pm && ksize == 3 && stride == 1 && pad == 1 && H >= 3 && W >= 3 && H * W <= 81 && batch >= 256)
        return dispatch_position_major(s, (hipStream_t)stream_);
    return dispatch<1>(s, workspace, workspace_bytes, (hipStream_t)stream_);
}

}  // extern "C"
