// Round 6: a self-contained reproducer attempt for the co-residency corruption (profiles/r06_costream_victim_side.txt).
// One workgroup = 8 waves: waves 0-3 (one per SIMD) are VICTIMS running a chain of one kind of f32 operation on full-mantissa pseudo-random operands,
// waves 4-7 (the second wave of every SIMD) are AGGRESSORS running back-to-back matrix instructions.  The victim's per-lane result is deterministic, so
// every launch must reproduce the result of the launch with idle aggressors bit for bit.
//   victim kinds:    0 v_pk_fma_f32   1 v_fma_f32   2 v_mfma_f32_16x16x4_f32   3 v_pk_mul_f32 + v_pk_add_f32   4 integer v_mad_u32_u24   5 ds_read_b128 of a pattern
//   aggressor kinds: 0 idle (s_sleep)  1 v_mfma_f32_32x32x16_bf16 on random data   2 the same on zeros   3 v_mfma_f32_32x32x2_f32 on random data
//                    4 bf16 MFMA on random data, results feeding back (acc chain only; same as 1 but 1 accumulator)   5 v_mfma_f32_16x16x32_bf16 random
// build: hipcc --offload-arch=gfx950 -O3 -o tools/costream/fma_canary tools/costream/fma_canary.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
// a float in [1, 2) or [0.5, 1) with a full random mantissa
__device__ __forceinline__ float rf(unsigned& s, float lo) { return __uint_as_float((rnd(s) >> 9) | 0x3f800000u) * lo; }

__global__ __launch_bounds__(512, 1) void canary(int vkind, int akind, int iters, int agg_iters, float* __restrict__ out, unsigned* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) float pat[4][64 * 4 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned seed = 0x9E3779B9u * (unsigned)(lane + 64 * (wave & 3) + 1) + 12345u * blockIdx.x;
    if (wave < 4) {
        // ---- victim ----
        float a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = rf(seed, 0.5f) + 0.2499f; b[i] = rf(seed, 1.0f) - 1.5f; }      // |a| < 1: the chain stays bounded
        if (vkind == 5) {
            for (int i = 0; i < 32; ++i) pat[wave][lane * 32 + i] = rf(seed, 1.0f);
        }
        __builtin_amdgcn_s_barrier();
        f32x2 acc = {rf(seed, 1.0f), rf(seed, 1.0f)};
        f32x4 macc = {0.f, 0.f, 0.f, 0.f};
        unsigned iacc = rnd(seed);
        for (int it = 0; it < iters; ++it) {
            if (vkind == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const f32x2 av = {a[i], a[(i + 3) & 7]}, bv = {b[i], b[(i + 5) & 7]}; acc = __builtin_elementwise_fma(acc, av, bv); }
            } else if (vkind == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { acc.x = __builtin_fmaf(acc.x, a[i], b[i]); acc.y = __builtin_fmaf(acc.y, a[(i + 3) & 7], b[(i + 5) & 7]); }
            } else if (vkind == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) macc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i] * 0.25f, b[i], macc, 0, 0, 0);
                macc *= 0.5f;
            } else if (vkind == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x2 av = {a[i], a[(i + 3) & 7]}, bv = {b[i], b[(i + 5) & 7]};
                    f32x2 t = acc * av;
                    asm volatile("" : "+v"(t));
                    acc = t + bv;
                }
            } else if (vkind == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) iacc = (iacc & 0xffffffu) * (__float_as_uint(a[i]) & 0xffffffu) + __float_as_uint(b[i]);
            } else {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&pat[wave][((lane * 7 + it) & 63) * 32 + 4 * (it & 7)]);
                acc.x = acc.x * 0.5f + v.x + v.z; acc.y = acc.y * 0.5f + v.y + v.w;
            }
        }
        float* o = out + ((size_t)blockIdx.x * 256 + tid) * 4;
        o[0] = acc.x; o[1] = acc.y; o[2] = macc[0] + macc[1] + macc[2] + macc[3]; o[3] = __uint_as_float(iacc & 0x3fffffffu);
    } else {
        // ---- aggressor ----
        __builtin_amdgcn_s_barrier();
        if (akind == 0) {
            for (int it = 0; it < agg_iters; ++it) __builtin_amdgcn_s_sleep(64);
            return;
        }
        bf16x8 x[4], y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned u[4], v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u[j] = (akind == 2) ? 0u : ((rnd(seed) & 0x7fff7fffu) | 0x30003000u) & 0xbfffbfffu;      // bf16 pairs of moderate magnitude, random mantissas
                v[j] = (akind == 2) ? 0u : ((rnd(seed) & 0x7fff7fffu) | 0x30003000u) & 0xbfffbfffu;
            }
            x[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<f32x4*>(u));
            y[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<f32x4*>(v));
        }
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        f32x4 acc4[4] = {};
        for (int it = 0; it < agg_iters; ++it) {
            if (akind == 1 || akind == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[(i + r) & 3], y[i], acc[i], 0, 0, 0);
            } else if (akind == 3) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(f32x4, x[(i + r) & 3])[0], __builtin_bit_cast(f32x4, y[i])[0], acc[i], 0, 0, 0);
            } else if (akind == 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[r & 3], y[(r >> 2) & 3], acc[0], 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[(i + r) & 3], y[i], acc4[i], 0, 0, 0);
            }
            // keep the accumulators bounded without touching the operand toggle rate
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
        if (s == 1234.5678f) sink[0] = 1;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    float* out; unsigned* sink;
    const size_t n = (size_t)wgs * 256 * 4;
    CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&sink, 16)); CK(hipMemset(sink, 0, 16));
    std::vector<float> ref(n), got(n);
    const char* vn[] = {"v_pk_fma_f32", "v_fma_f32", "v_mfma_f32_16x16x4_f32", "v_pk_mul+v_pk_add", "v_mad_u32_u24", "ds_read_b128"};
    const char* an[] = {"idle", "bf16 MFMA 32x32x16 random", "bf16 MFMA 32x32x16 zeros", "f32 MFMA 32x32x2 random", "bf16 MFMA one accumulator", "bf16 MFMA 16x16x32 random"};
    printf("%d workgroups x (4 victim + 4 aggressor waves), %d victim iterations x 8 ops, %d launches per cell; cells = launches whose victim output differs from the idle-aggressor launch\n", wgs, iters, reps);
    for (int vk = 0; vk < 6; ++vk) {
        // aggressor iterations sized so that the aggressor outlives the victim: measured per kind by trial (16 MFMAs of 32 cycles per iteration ~ 512 cycles)
        hipLaunchKernelGGL(canary, dim3(wgs), dim3(512), 0, 0, vk, 0, iters, 1, out, sink);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ref.data(), out, n * 4, hipMemcpyDeviceToHost));
        for (int ak = 0; ak < 6; ++ak) {
            int bad_launches = 0; long bad_vals = 0; long lane_hist[4] = {0, 0, 0, 0};
            const int agg_iters = ak == 0 ? iters / 8 : iters / 4;
            for (int r = 0; r < reps; ++r) {
                hipLaunchKernelGGL(canary, dim3(wgs), dim3(512), 0, 0, vk, ak, iters, agg_iters, out, sink);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(got.data(), out, n * 4, hipMemcpyDeviceToHost));
                long b = 0;
                for (size_t i = 0; i < n; ++i)
                    if (memcmp(&got[i], &ref[i], 4)) { ++b; ++lane_hist[((i / 4) & 63) >> 4]; }
                if (b) { ++bad_launches; bad_vals += b; }
            }
            printf("victim %-24s aggressor %-28s: %2d / %d launches differ, %ld values (lanes 0-15 / 16-31 / 32-47 / 48-63: %ld %ld %ld %ld)\n", vn[vk], an[ak], bad_launches, reps,
                   bad_vals, lane_hist[0], lane_hist[1], lane_hist[2], lane_hist[3]);
            fflush(stdout);
        }
    }
    return 0;
}
