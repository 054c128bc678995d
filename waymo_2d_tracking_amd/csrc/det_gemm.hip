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
