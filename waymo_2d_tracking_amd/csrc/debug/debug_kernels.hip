// Laboratory kernels: NOT part of the product library.  Built into libwaymotrack.so only when WD_DEBUG_BUILD=1 is set for
// `python -m waymo_2d_tracking_amd.build` (tools/costream/*, tools/hold_experiment.py, tools/archive/diag_*.py use them); declared in
// csrc/debug/waymodet_debug.h, never in include/.
#include "../common.h"
#include "waymodet_debug.h"
#include <cstdlib>
#include <cstring>

// ---- diagnostics: a "canary" workgroup for co-residency experiments (tools/archive/diag_canary.py) ----------------------------------------------
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

// Experiments only (tools/hold_experiment.py): n_wg one-wave workgroups that each hold `lds_bytes` of LDS and spin for `cycles` shader
// cycles - a stand-in for the tracker's workgroups next to the detector (how much does HOLDING compute units cost the other stream?).
namespace {
__global__ __launch_bounds__(64) void debug_hold_kernel(long long cycles, int* sink) {
    extern __shared__ int hold_lds[];
    const long long t0 = clock64();
    int v = 0;
    while (clock64() - t0 < cycles) { hold_lds[threadIdx.x] = v; v += hold_lds[(threadIdx.x + 1) & 63]; }
    if (v == 0x7fffffff) sink[0] = v;
}
}  // namespace

extern "C" int wd_debug_hold(int n_wg, int lds_bytes, long long cycles, int* sink, void* stream) {
    WT_TRY(wt::ensure_device());
    if (n_wg < 1 || lds_bytes < 256) { wt::set_error("wd_debug_hold: bad argument"); return WT_ERR_INVALID; }
    if (lds_bytes > 48 * 1024)
        WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(debug_hold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(debug_hold_kernel, dim3((unsigned)n_wg), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, cycles, sink);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

// ---- round 6: a pure-register matrix-instruction "burner" (tools/costream/run_burn.sh) ---------------------------------------------------
// 512 threads = two waves per SIMD, ~100 registers, no LDS, no memory traffic inside the loop: nothing but back-to-back matrix instructions on
// register operands.  kind: 1 v_mfma_f32_32x32x16_bf16 on pseudo-random operands, 2 the same on zeros, 3 v_mfma_f32_32x32x2_f32 on random operands,
// 4 v_mfma_f32_16x16x32_bf16 random, 5 bf16 32x32x16 with operands of ONE repeated value (no toggling between instructions),
// 6 = 1 with ONE random A and ONE random B register set (the register footprint of kinds 2 / 5, random data), 7 = 2 (zeros) with 28 extra live registers
// (the footprint of kind 1, zero data), 8 = 6 with A == B (one random register set for both operands), 9 = 4 (the 16x16x32 form) with ONE random A and ONE
// random B register set (what a deformable conv on this instruction would issue: a tap's weight fragment against every pixel group)
namespace {
using bf16x8_ = __attribute__((ext_vector_type(8))) __bf16;
using f32x4_ = __attribute__((ext_vector_type(4))) float;
using f32x16_ = __attribute__((ext_vector_type(16))) float;
template <int kind>
__global__ __launch_bounds__(512, 2) void mfma_burn_kernel(int iters, unsigned* __restrict__ sink) {
    unsigned seed = 0x9E3779B9u * (threadIdx.x + 1) + 7919u * blockIdx.x;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed; };
    bf16x8_ x[4], y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned u[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u[j] = (kind == 2 || kind == 7) ? 0u : kind == 5 ? 0x3f803f80u : ((rnd() & 0x7fff7fffu) | 0x30003000u) & 0xbfffbfffu;
            v[j] = (kind == 2 || kind == 7) ? 0u : kind == 5 ? 0x3f803f80u : ((rnd() & 0x7fff7fffu) | 0x30003000u) & 0xbfffbfffu;
        }
        if (kind == 8) { for (int j = 0; j < 4; ++j) v[j] = u[j]; }
        x[i] = __builtin_bit_cast(bf16x8_, *reinterpret_cast<f32x4_*>(u));
        y[i] = __builtin_bit_cast(bf16x8_, *reinterpret_cast<f32x4_*>(v));
    }
    if (kind == 6 || kind == 8 || kind == 9) {
#pragma unroll
        for (int i = 1; i < 4; ++i) { x[i] = x[0]; y[i] = y[0]; }
    }
    unsigned pad[28];
    if (kind == 7) {
#pragma unroll
        for (int j = 0; j < 28; ++j) { pad[j] = rnd(); asm volatile("" : "+v"(pad[j])); }
    }
    f32x16_ acc[4];
    f32x4_ acc4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        acc4[i] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    }
    for (int it = 0; it < iters; ++it) {
        if (kind == 3) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(f32x4_, x[(i + r) & 3])[0], __builtin_bit_cast(f32x4_, y[i])[0], acc[i], 0, 0, 0);
        } else if (kind == 4 || kind == 9) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[(i + r) & 3], y[i], acc4[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[(i + r) & 3], y[i], acc[i], 0, 0, 0);
        }
        if ((it & 15) == 15) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[i] *= 1e-3f; acc4[i] *= 1e-3f; }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
        s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    }
    if (kind == 7) {
#pragma unroll
        for (int j = 0; j < 28; ++j) { asm volatile("" : "+v"(pad[j])); s += __uint_as_float(pad[j] & 0x3fffffffu); }
    }
    if (s == 1234.5678f) sink[0] = 1;
}
}  // namespace

extern "C" int wd_debug_mfma_burn(int workgroups, int kind, int iters, unsigned* sink, void* stream_) {
    WT_TRY(wt::ensure_device());
    const dim3 g((unsigned)workgroups), b(512);
    hipStream_t st = (hipStream_t)stream_;
    switch (kind) {
        case 1: hipLaunchKernelGGL(mfma_burn_kernel<1>, g, b, 0, st, iters, sink); break;
        case 2: hipLaunchKernelGGL(mfma_burn_kernel<2>, g, b, 0, st, iters, sink); break;
        case 3: hipLaunchKernelGGL(mfma_burn_kernel<3>, g, b, 0, st, iters, sink); break;
        case 4: hipLaunchKernelGGL(mfma_burn_kernel<4>, g, b, 0, st, iters, sink); break;
        case 6: hipLaunchKernelGGL(mfma_burn_kernel<6>, g, b, 0, st, iters, sink); break;
        case 7: hipLaunchKernelGGL(mfma_burn_kernel<7>, g, b, 0, st, iters, sink); break;
        case 8: hipLaunchKernelGGL(mfma_burn_kernel<8>, g, b, 0, st, iters, sink); break;
        case 9: hipLaunchKernelGGL(mfma_burn_kernel<9>, g, b, 0, st, iters, sink); break;
        default: hipLaunchKernelGGL(mfma_burn_kernel<5>, g, b, 0, st, iters, sink); break;
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}
