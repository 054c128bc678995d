// Micro-probe (experiments only): how do v_mfma_f32_16x16x4_f32 bursts, VALU work and ds_read_b128 traffic of the waves on one SIMD
// add up on gfx950?  One workgroup per CU (LDS-limited), 1 or 2 waves per SIMD; per iteration a wave issues
//   G: nv VALU instructions (kind: 0 = independent v_fma_f32, 1 = v_pk_fma_f32, 2 = dependent v_fma chain) + nl ds_read_b128
//   M: nm MFMAs on 4 accumulator chains
// cycles per iteration per wave are reported (s_memtime around the loop, max over waves of workgroup 0) and the wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int NV, int KIND, int NL, int NM, int SPLIT>
__global__ __launch_bounds__(512) void probe(int iters, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float va[16];
    f32x2 vp[8];
    for (int j = 0; j < 16; ++j) va[j] = threadIdx.x * 0.001f + j;
    for (int j = 0; j < 8; ++j) vp[j] = f32x2{va[j], va[j + 8]};
    float a = threadIdx.x * 0.5f, b = 1.0001f;
    const unsigned ad = (threadIdx.x & 63) * 16;
    f32x4 ld[NL > 0 ? NL : 1];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        // ---- G ----
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(va[j & 15]) : "v"(b));
            else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(vp[j & 7]) : "v"(vp[(j + 1) & 7]));
            else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(va[0]) : "v"(b));
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[j]) : "v"(ad), "n"(j * 1024));
        // ---- M ---- (SPLIT: the VALU work is interleaved between the MFMAs instead of preceding them)
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(a), "v"(b));
            if (SPLIT) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(va[j & 15]) : "v"(b));
        }
        if (NL > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < NL; ++j) asm volatile("" :: "v"(ld[j]));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int j = 0; j < 16; ++j) s += va[j];
    for (int j = 0; j < 8; ++j) s += vp[j][0] + vp[j][1];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int KIND, int NL, int NM, int SPLIT>
void run(const char* name, int threads, unsigned long long* d_out, float* d_sink) {
    const int iters = 4000, grid = 256;
    auto k = probe<NV, KIND, NL, NM, SPLIT>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 100 * 1024, 0, 100, d_out, d_sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 100 * 1024, 0, iters, d_out, d_sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(8);
    hipMemcpy(h.data(), d_out, 64, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
    // s_memtime ticks at a constant 100 MHz on this chip family: report wall-derived cycles at the measured clock instead
    printf("%-44s waves/SIMD %d  nv %3d kind %d nl %2d nm %2d split %d : %8.1f us  = %7.1f ns per iteration per wave-pair slot, memtime ticks/iter %.2f\n",
           name, threads / 256, NV, KIND, NL, NM, SPLIT, ms * 1e3, ms * 1e6 / iters, (double)mx / iters);
}

int main() {
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, 8 * 8 * 256); hipMalloc(&d_sink, 4 * 512 * 256);
    for (int threads = 256; threads <= 512; threads += 256) {
        run<0, 0, 0, 16, 0>("mfma only", threads, d_out, d_sink);
        run<32, 0, 0, 0, 0>("32 v_fma only", threads, d_out, d_sink);
        run<64, 0, 0, 0, 0>("64 v_fma only", threads, d_out, d_sink);
        run<32, 1, 0, 0, 0>("32 v_pk_fma only", threads, d_out, d_sink);
        run<32, 2, 0, 0, 0>("32 dependent v_fma only", threads, d_out, d_sink);
        run<0, 0, 13, 0, 0>("13 ds_read_b128 only", threads, d_out, d_sink);
        run<32, 0, 0, 16, 0>("32 v_fma + 16 mfma", threads, d_out, d_sink);
        run<64, 0, 0, 16, 0>("64 v_fma + 16 mfma", threads, d_out, d_sink);
        run<32, 1, 0, 16, 0>("32 v_pk_fma + 16 mfma", threads, d_out, d_sink);
        run<16, 1, 0, 16, 0>("16 v_pk_fma + 16 mfma", threads, d_out, d_sink);
        run<0, 0, 13, 16, 0>("13 ds_read + 16 mfma", threads, d_out, d_sink);
        run<40, 0, 13, 16, 0>("40 v_fma + 13 ds_read + 16 mfma", threads, d_out, d_sink);
        run<24, 0, 13, 16, 1>("24+16 v_fma (16 between mfmas) + 13 ds + 16 mfma", threads, d_out, d_sink);
        run<0, 0, 0, 16, 1>("16 mfma with a v_fma after each", threads, d_out, d_sink);
    }
    return 0;
}
