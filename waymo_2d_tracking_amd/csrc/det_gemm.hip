// f32 GEMM on the gfx950 matrix cores: C = act(A (M,K) * Bt (N,K)^T + bias [+ residual]).
// Used for the cascade box-head FC 12544 -> 1024 (logs/12442/job.log:1146-1218) and for the
// 1x1 convolutions of the NHWC backbone / FPN (a 1x1 conv on NHWC storage IS this GEMM).
//
// v_mfma_f32_16x16x4_f32: exact f32 (bitwise an fmaf chain over k), 157 TFLOP/s chip peak.  256 threads = 2x2
// waves; each wave owns (16*WM) x (16*WN) outputs.  K is streamed in 32-wide slabs through double-buffered LDS
// (row stride 34 floats: the 32 lanes of a ds_read_b32 group hit 32 distinct banks; 8-byte aligned rows keep the
// ds_write_b64 of the global->LDS copy aligned).  Next slab's global loads are issued before the MFMAs of the
// current slab; bias / residual / ReLU are fused in the epilogue.
#include "common.h"
#include "../../include/waymodet.h"
#include <cstdlib>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int BK = 32;
constexpr int LD = BK + 2;

template <int WM, int WN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ Bt,
                                                      const float* __restrict__ bias, const float* __restrict__ residual,
                                                      int relu, int M, int N, int K, float* __restrict__ C, int splitk) {
    constexpr int BM = 32 * WM, BN = 32 * WN;            // 2 waves along each dimension
    constexpr int A_F4 = BM * BK / 4 / 256;              // float4 loads per thread per slab
    constexpr int B_F4 = BN * BK / 4 / 256;
    __shared__ __attribute__((aligned(16))) float sA[2][BM * LD];
    __shared__ __attribute__((aligned(16))) float sB[2][BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive workgroups (round-robin over the 8 XCDs) walk N first inside an M stripe
    const int tiles_n = (N + BN - 1) / BN;
    const int tiles = tiles_n * ((M + BM - 1) / BM);
    const int tile_id = blockIdx.x % tiles, kz = blockIdx.x / tiles;     // split-K: slice kz of the K range
    const int bm = tile_id / tiles_n, bn = tile_id % tiles_n;
    const int m0 = bm * BM, n0 = bn * BN;

    float4 ra[A_F4], rb[B_F4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + i * 256;                  // float4 index inside the slab: row = f / 8, kq = f % 8
            const int r = f >> 3, kq = f & 7;
            const int gm = m0 + r, gk = k0 + kq * 4;
            ra[i] = (gm < M && gk < K) ? *reinterpret_cast<const float4*>(A + (size_t)gm * K + gk) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int f = tid + i * 256;
            const int r = f >> 3, kq = f & 7;
            const int gn = n0 + r, gk = k0 + kq * 4;
            rb[i] = (gn < N && gk < K) ? *reinterpret_cast<const float4*>(Bt + (size_t)gn * K + gk) : make_float4(0, 0, 0, 0);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + i * 256;
            const int r = f >> 3, kq = f & 7;
            float2* p = reinterpret_cast<float2*>(&sA[buf][r * LD + kq * 4]);
            p[0] = make_float2(ra[i].x, ra[i].y);
            p[1] = make_float2(ra[i].z, ra[i].w);
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int f = tid + i * 256;
            const int r = f >> 3, kq = f & 7;
            float2* p = reinterpret_cast<float2*>(&sB[buf][r * LD + kq * 4]);
            p[0] = make_float2(rb[i].x, rb[i].y);
            p[1] = make_float2(rb[i].z, rb[i].w);
        }
    };

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nslab_all = (K + BK - 1) / BK;
    const int per = (nslab_all + splitk - 1) / splitk;
    const int s_begin = kz * per, s_end = (s_begin + per < nslab_all) ? s_begin + per : nslab_all;
    const int nslab = s_end - s_begin;
    if (nslab <= 0) return;
    const int kbase = s_begin * BK;
    gload(kbase);
    sstore(0);
    __syncthreads();
    const int arow = wm * 16 * WM + (lane & 15);
    const int brow = wn * 16 * WN + (lane & 15);
    const int kl = lane >> 4;
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) gload(kbase + (s + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            float a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = sA[buf][(arow + 16 * i) * LD + kk * 4 + kl];
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = sB[buf][(brow + 16 * j) * LD + kk * 4 + kl];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nslab) sstore(buf ^ 1);
        __syncthreads();
    }
    // epilogue: C/D layout of the 16x16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = n0 + wn * 16 * WN + 16 * j + (lane & 15);
            if (col >= N) continue;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 16 * WM + 16 * i + (lane >> 4) * 4 + r;
                if (row >= M) continue;
                if (splitk > 1) {                      // partial sums: C was zeroed, epilogue runs in a second pass
                    atomicAdd(&C[(size_t)row * N + col], acc[i][j][r]);
                    continue;
                }
                float v = acc[i][j][r] + bv;
                if (residual) v += residual[(size_t)row * N + col];
                if (relu) v = fmaxf(v, 0.f);
                C[(size_t)row * N + col] = v;
            }
        }
}

// ---- v2 (round 2): 128 x 128 tiles, fragments by ds_read_b128, deterministic split-K -----------------------------------
// Each wave owns 64 x 64 outputs (4 x 4 MFMA tiles, 64 accumulator VGPRs).  A K slab of 32 sits in LDS as rows of 128 bytes whose
// eight 16-byte granules are XOR-swizzled with a table of the row pair (kSwz) so that the ds_read_b128 lane groups of gfx950
// ({0-3, 12-15, 20-27}, ...) hit 16 distinct granules: lane (r = lane & 15, kq = lane >> 4) reads k = 8 kq .. 8 kq + 7 of its row
// as two float4 - the SAME k permutation on both operands, so MFMA step t multiplies k = 8 kq + t of A with k = 8 kq + t of B.
// 16 ds_read_b128 feed 128 MFMAs per wave and slab (v1: 64 ds_read_b32 per 32 MFMAs).  Split-K partial tiles go to a workspace
// [slice][M][N] and are summed in slice order by the epilogue kernel (bias + ReLU fused): bit-reproducible, no atomics.
constexpr unsigned kSwz = 0x32765410u;          // f(h) for h = 0..7 = 0 1 4 5 6 7 2 3 (exhaustive search over permutations)
__device__ __forceinline__ int swz(int row) { return (int)((kSwz >> (((row >> 1) & 7) * 4)) & 7u); }

__global__ __launch_bounds__(256, 2) void gemm_nt_v2_kernel(const float* __restrict__ A, const float* __restrict__ Bt, int M, int N,
                                                            int K, int splitk, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float smem2[];
    float* sA = smem2;                              // [2][128 * 32]
    float* sB = smem2 + 2 * 128 * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + 127) / 128, tiles = tiles_n * ((M + 127) / 128);
    // workgroup b runs on XCD b % 8: with kz = b % splitk (8 slices for the FC) every XCD works on ONE K slice of all tiles, so the
    // 8 workgroups sharing an A (or B) row block of that slice meet in the same L2 (tile-major numbering spread them over all
    // eight L2s: 247 us -> see DESIGN.md)
    const int kz = blockIdx.x % splitk, tile_id = blockIdx.x / splitk;
    (void)tiles;
    const int m0 = (tile_id / tiles_n) * 128, n0 = (tile_id % tiles_n) * 128;
    const int nslab_all = (K + 31) / 32;
    const int per = (nslab_all + splitk - 1) / splitk;
    const int s_begin = kz * per, s_end = (s_begin + per < nslab_all) ? s_begin + per : nslab_all;
    const int nslab = s_end - s_begin;

    // global -> registers -> LDS: thread handles float4 f = tid + 256 i of the 128 x 8 float4 grid of each operand.  Rows past
    // M / N are clamped to the last row (their products land in accumulator rows that are never stored), K is a multiple of 32
    // (checked by the launcher): no bounds tests and no address arithmetic beyond one pointer bump per slab in the loop.
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;      // staging registers (named: arrays captured by the lambdas ended up in scratch)
    // per-thread element offsets (uniform base + 32-bit lane offset: the loads take the SGPR-base form, the slab advance is scalar)
    const int f0 = tid, f1 = tid + 256, f2 = tid + 512, f3 = tid + 768;
    auto row_off = [&](int f, int base, int limit) {
        const int r = f >> 3, q = f & 7;
        const int rr = (base + r < limit) ? base + r : limit - 1;
        return rr * K + q * 4;
    };
    const int oa0 = row_off(f0, m0, M), oa1 = row_off(f1, m0, M), oa2 = row_off(f2, m0, M), oa3 = row_off(f3, m0, M);
    const int ob0 = row_off(f0, n0, N), ob1 = row_off(f1, n0, N), ob2 = row_off(f2, n0, N), ob3 = row_off(f3, n0, N);
    auto lds_off = [&](int f) { const int r = f >> 3, q = f & 7; return r * 32 + ((q ^ swz(r)) << 2); };
    const int so0 = lds_off(f0), so1 = lds_off(f1), so2 = lds_off(f2), so3 = lds_off(f3);
    auto gload = [&](int slab) {
        const float* Ab = A + (size_t)(s_begin + slab) * 32;
        const float* Bb = Bt + (size_t)(s_begin + slab) * 32;
        ra0 = *reinterpret_cast<const float4*>(Ab + oa0); ra1 = *reinterpret_cast<const float4*>(Ab + oa1);
        ra2 = *reinterpret_cast<const float4*>(Ab + oa2); ra3 = *reinterpret_cast<const float4*>(Ab + oa3);
        rb0 = *reinterpret_cast<const float4*>(Bb + ob0); rb1 = *reinterpret_cast<const float4*>(Bb + ob1);
        rb2 = *reinterpret_cast<const float4*>(Bb + ob2); rb3 = *reinterpret_cast<const float4*>(Bb + ob3);
    };
    auto sstore = [&](int buf) {
        float* a_ = sA + buf * 128 * 32;
        float* b_ = sB + buf * 128 * 32;
        *reinterpret_cast<float4*>(a_ + so0) = ra0; *reinterpret_cast<float4*>(a_ + so1) = ra1;
        *reinterpret_cast<float4*>(a_ + so2) = ra2; *reinterpret_cast<float4*>(a_ + so3) = ra3;
        *reinterpret_cast<float4*>(b_ + so0) = rb0; *reinterpret_cast<float4*>(b_ + so1) = rb1;
        *reinterpret_cast<float4*>(b_ + so2) = rb2; *reinterpret_cast<float4*>(b_ + so3) = rb3;
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kq = lane >> 4;
    int offA[4], offB[4];                           // float offset of the lane's first granule in a buffer; second = ^ 4 floats
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra_ = wm * 64 + 16 * i + r16, rb_ = wn * 64 + 16 * i + r16;
        offA[i] = ra_ * 32 + (((2 * kq) ^ swz(ra_)) << 2);
        offB[i] = rb_ * 32 + (((2 * kq) ^ swz(rb_)) << 2);
    }
    // Software pipeline with ONE fragment register set and ONE barrier per slab.  A slab's 128 MFMAs run as two halves: H0 uses
    // the first float4 of every fragment (k-steps 0-3), H1 the second (k-steps 4-7).  While H0(s) runs the second halves of slab
    // s arrive from LDS; while H1(s) runs the first halves of slab s + 1 arrive (their registers are free by then); the global
    // loads of slab s + 2 are in flight over the whole iteration and are written to the buffer of slab s after H1(s).
    //   iteration s:  read f1(s) | H0(s) | barrier | read f0(s + 1) | H1(s) | store slab s + 2
    // The barrier orders both hazards: slab s + 1 was stored before it (end of iteration s - 1) and every wave has finished its
    // reads of slab s's buffer when the store of slab s + 2 overwrites it.
    float4 fa[4][2], fb[4][2];
    auto read_half = [&](int buf, int h) {
        const float* pa = sA + buf * 128 * 32;
        const float* pb = sB + buf * 128 * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (h == 0) {
                fa[i][0] = *reinterpret_cast<const float4*>(pa + offA[i]);
                fb[i][0] = *reinterpret_cast<const float4*>(pb + offB[i]);
            } else {
                fa[i][1] = *reinterpret_cast<const float4*>(pa + (offA[i] ^ 4));
                fb[i][1] = *reinterpret_cast<const float4*>(pb + (offB[i] ^ 4));
            }
        }
    };
#define WD_MFMA_HALF(h)                                                                                                  \
    {                                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][h].x, fb[j][h].x, acc[i][j], 0, 0, 0);                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][h].y, fb[j][h].y, acc[i][j], 0, 0, 0);                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][h].z, fb[j][h].z, acc[i][j], 0, 0, 0);                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][h].w, fb[j][h].w, acc[i][j], 0, 0, 0);                \
    }
    if (nslab > 0) {
        gload(0);
        sstore(0);
        if (nslab > 1) gload(1);
        __syncthreads();
        read_half(0, 0);
        if (nslab > 1) sstore(1);
        int s = 0;
#ifndef WD_GEMM_NO_INTERLEAVE
        // steady state (branch-free body): the loads / LDS reads / LDS writes are interleaved with the MFMAs of the same half by
        // sched_group_barrier (one memory instruction per 4 MFMAs): issued in the matrix pipe's shadow instead of in front of it
        for (; s + 2 < nslab; ++s) {
            const int buf = s & 1;
            gload(s + 2);
            read_half(buf, 1);
            WD_MFMA_HALF(0)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // 4 MFMA
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_half(buf ^ 1, 0);
            WD_MFMA_HALF(1)
            sstore(buf);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // 1 DS write
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        for (; s < nslab; ++s) {                                // tail (and the whole loop with WD_GEMM_NO_INTERLEAVE)
            const int buf = s & 1;
            if (s + 2 < nslab) gload(s + 2);
            read_half(buf, 1);
            __builtin_amdgcn_sched_barrier(0);          // hipcc otherwise sinks these reads behind H0 and waits for them at once
            WD_MFMA_HALF(0)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0) only: LDS reads of this wave done, global loads stay in flight
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < nslab) read_half(buf ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            WD_MFMA_HALF(1)
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < nslab) sstore(buf);
        }
    }
#undef WD_MFMA_HALF
    // partial tile -> part[kz][M][N]; C/D layout of a 16 x 16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg
    float* dst = part + (size_t)kz * M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + 16 * j + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + 16 * i + kq * 4 + r;
                if (row < M && col < N) dst[(size_t)row * N + col] = acc[i][j][r];
            }
        }
}

// C = act(sum over slices in slice order + bias [+ residual]); float4 per thread (N % 4 == 0)
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float4* __restrict__ part, int splitk, long mn4, int n4,
                                                          const float4* __restrict__ bias, const float4* __restrict__ residual,
                                                          int relu, float4* __restrict__ C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= mn4) return;
    float4 v = part[i];
    for (int z = 1; z < splitk; ++z) {
        const float4 p = part[(size_t)z * mn4 + i];
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (bias) { const float4 b = bias[i % n4]; v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    if (residual) { const float4 r = residual[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    C[i] = v;
}

}  // namespace

extern "C" int wd_gemm_nt_f32(const float* A, const float* Bt, const float* bias, const float* residual, int relu,
                              int M, int N, int K, float* C, void* stream_) {
    WT_TRY(wt::ensure_device());
    if (M <= 0 || N <= 0) return WT_OK;
    if (K <= 0 || (K & 3) || ((uintptr_t)A & 15) || ((uintptr_t)Bt & 15)) {
        wt::set_error("wd_gemm_nt_f32: K must be a positive multiple of 4 and A/Bt 16-byte aligned (K=%d)", K);
        return WT_ERR_INVALID;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const long big_tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (big_tiles >= 192) {
        hipLaunchKernelGGL((gemm_nt_kernel<4, 4>), dim3((unsigned)big_tiles), dim3(256), 0, stream, A, Bt, bias, residual,
                           relu, M, N, K, C, 1);
    } else {
        const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
        // few tiles and a long K (the box-head FC: 256 tiles, K = 12544): split K so that >= 2 workgroups share a CU and
        // hide each other's LDS latency; partial sums meet in C through f32 atomics, the epilogue runs as a second pass
        // Run-to-run determinism: with TWO slices the result 0 + a + b is the same in either arrival order (f32 addition is
        // commutative), with more slices the atomic order would matter - so the split is capped at 2 unless
        // WD_GEMM_SPLITK_MAX (experiments) raises it.
        static const int splitk_max = []() { const char* e = getenv("WD_GEMM_SPLITK_MAX"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
        int splitk = 1;
        if (!residual && (N % 4) == 0 && K >= 2048) {
            while (tiles * splitk < 512 && splitk * 2 <= splitk_max && K / (splitk * 2) >= 1024) splitk *= 2;
        }
        if (splitk > 1) WT_HIP(hipMemsetAsync(C, 0, sizeof(float) * (size_t)M * N, stream));
        hipLaunchKernelGGL((gemm_nt_kernel<2, 2>), dim3((unsigned)(tiles * splitk)), dim3(256), 0, stream, A, Bt, bias, residual,
                           relu, M, N, K, C, splitk);
        if (splitk > 1 && (bias || relu)) {
            WT_HIP(hipGetLastError());
            return wd_bias_relu_f32(C, bias, M, N, relu, stream_);
        }
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

/* Split-K slices the v2 path would use and the workspace it needs (0 bytes = the shape stays on the v1 kernel). */
extern "C" size_t wd_gemm_nt_workspace(int M, int N, int K) {
    if (M < 256 || N < 256 || K < 1024 || (N & 3) || (K & 31) || (long)M * K >= (1l << 31) || (long)N * K >= (1l << 31)) return 0;
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    static const int target = []() { const char* e = getenv("WD_GEMM_V2_WGS"); return e ? atoi(e) : 448; }();     // experiments
    int splitk = 1;
    while (tiles * splitk < target && K / (splitk * 2) >= 512) splitk *= 2;
    return (size_t)splitk * M * N * sizeof(float);
}

/* Same contract as wd_gemm_nt_f32 for large shapes (the box-head FC): 128 x 128 tiles, b128 fragment reads, deterministic
 * two-pass split-K through `workspace` (wd_gemm_nt_workspace bytes). */
extern "C" int wd_gemm_nt_ws_f32(const float* A, const float* Bt, const float* bias, const float* residual, int relu, int M, int N,
                                 int K, float* C, void* workspace, size_t workspace_bytes, void* stream_) {
    WT_TRY(wt::ensure_device());
    const size_t need = wd_gemm_nt_workspace(M, N, K);
    if (need == 0) return wd_gemm_nt_f32(A, Bt, bias, residual, relu, M, N, K, C, stream_);
    if (!workspace || workspace_bytes < need || ((uintptr_t)A & 15) || ((uintptr_t)Bt & 15) || ((uintptr_t)C & 15) ||
        ((uintptr_t)workspace & 15) || (bias && ((uintptr_t)bias & 15)) || (residual && ((uintptr_t)residual & 15))) {
        wt::set_error("wd_gemm_nt_ws_f32: workspace too small (%zu < %zu) or unaligned pointer", workspace_bytes, need);
        return WT_ERR_CAPACITY;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const int splitk = (int)(need / ((size_t)M * N * sizeof(float)));
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    static wt::OncePerDevice attr;
    if (const int dev = wt::device_index(); attr.needed(dev)) {
        WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        attr.mark(dev);
    }
    hipLaunchKernelGGL(gemm_nt_v2_kernel, dim3((unsigned)(tiles * splitk)), dim3(256), 65536, stream, A, Bt, M, N, K, splitk,
                       (float*)workspace);
    const long mn4 = (long)M * N / 4;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((mn4 + 255) / 256)), dim3(256), 0, stream, (const float4*)workspace, splitk,
                       mn4, N / 4, (const float4*)bias, (const float4*)residual, relu, (float4*)C);
    WT_HIP(hipGetLastError());
    return WT_OK;
}
