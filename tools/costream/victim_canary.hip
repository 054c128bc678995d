// Round 6: how little of a victim does it take?  A self-contained program: stream 1 runs the pure-register matrix-instruction burner (zeros, ONE operand register set: the most
// aggressive aggressor of profiles/r06_costream_victim_side.txt), stream 0 a SYNTHETIC victim shaped like grouped_conv3x3_c8_kernel's inner loop - NW float2 weights resident in
// registers for the whole kernel, inputs read from an LDS patch with ds_read_b128, one v_pk_fma_f32 per weight - whose per-thread result is deterministic.  Every victim launch is
// compared bit for bit with the launch that ran alone.  NW sweeps the live-register count (the two real victims hold 230 / 256).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/costream/victim_canary tools/costream/victim_canary.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int NW, int OCC>
__global__ __launch_bounds__(256, OCC) void victim(const float* __restrict__ wsrc, const float* __restrict__ xsrc, int iters, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float patch[];
    const int tid = threadIdx.x;
    f32x2 w[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const float2 t = *reinterpret_cast<const float2*>(wsrc + ((size_t)(blockIdx.x % 64) * 256 + tid) * 2 * NW + 2 * k);
        w[k] = (f32x2){t.x, t.y};
    }
    for (int i = tid; i < 12800; i += 256) patch[i] = xsrc[i];
    __syncthreads();
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
    const int gl = (tid & 63) >> 2;
    for (int it = 0; it < iters; ++it) {
        const float* src = patch + ((it * 37 + (tid >> 6) * 11) % 90) * 128 + gl * 8;
#pragma unroll
        for (int k = 0; k < NW; k += 8) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(src + ((k >> 3) % 9) * 128), x1 = *reinterpret_cast<const f32x4*>(src + ((k >> 3) % 9) * 128 + 4);
            a0 += w[k] * x0.x; a1 += w[k + 1] * x0.y; a2 += w[k + 2] * x0.z; a3 += w[k + 3] * x0.w;
            a0 += w[k + 4] * x1.x; a1 += w[k + 5] * x1.y; a2 += w[k + 6] * x1.z; a3 += w[k + 7] * x1.w;
        }
        a0 *= 0.5f; a1 *= 0.5f; a2 *= 0.5f; a3 *= 0.5f;
    }
    const f32x2 r = (a0 + a1) + (a2 + a3);
    *reinterpret_cast<float2*>(out + ((size_t)blockIdx.x * 256 + tid) * 2) = make_float2(r.x, r.y);
}

__global__ __launch_bounds__(512, 2) void burner(int iters, unsigned* __restrict__ sink) {
    bf16x8 z = {};
    asm volatile("" : "+v"(z));
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z, z, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 1234.5f) sink[0] = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NW, int OCC>
int run(const float* wsrc, const float* xsrc, float* out, unsigned* sink, hipStream_t s0, hipStream_t s1, int wgs, int iters, int reps) {
    const size_t n = (size_t)wgs * 512;
    std::vector<float> ref(n), got(n);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(victim<NW, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 51200));
    hipFuncAttributes at{};
    CK(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(victim<NW, OCC>)));
    hipLaunchKernelGGL((victim<NW, OCC>), dim3(wgs), dim3(256), 51200, s0, wsrc, xsrc, iters, out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), out, n * 4, hipMemcpyDeviceToHost));
    int bad = 0, alone_bad = 0;
    long vals = 0, lanes[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {                 // control: alone again
        hipLaunchKernelGGL((victim<NW, OCC>), dim3(wgs), dim3(256), 51200, s0, wsrc, xsrc, iters, out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), out, n * 4, hipMemcpyDeviceToHost));
        if (memcmp(got.data(), ref.data(), n * 4)) ++alone_bad;
    }
    for (int r = 0; r < reps; ++r) {
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(burner, dim3(512), dim3(512), 0, s1, 400, sink);
        hipLaunchKernelGGL((victim<NW, OCC>), dim3(wgs), dim3(256), 51200, s0, wsrc, xsrc, iters, out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), out, n * 4, hipMemcpyDeviceToHost));
        long b = 0;
        for (size_t i = 0; i < n; ++i)
            if (memcmp(&got[i], &ref[i], 4)) { ++b; ++lanes[((i / 2) & 63) >> 4]; }
        if (b) { ++bad; vals += b; }
    }
    printf("victim: %3d weight registers x2, __launch_bounds__(256, %d), compiled to %3d VGPRs: alone %d / %d differ; next to the burner %2d / %d launches differ, %ld values "
           "(lanes 0-15 / 16-31 / 32-47 / 48-63: %ld %ld %ld %ld)\n", NW, OCC, at.numRegs, alone_bad, reps, bad, reps, vals, lanes[0], lanes[1], lanes[2], lanes[3]);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 600, reps = argc > 3 ? atoi(argv[3]) : 40;
    float *wsrc, *xsrc, *out;
    unsigned* sink;
    std::vector<float> hw((size_t)64 * 256 * 2 * 128), hx(12800);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hw) v = rnd();
    for (auto& v : hx) v = rnd() * 2.0f;
    CK(hipMalloc(&wsrc, hw.size() * 4)); CK(hipMalloc(&xsrc, hx.size() * 4)); CK(hipMalloc(&out, (size_t)wgs * 512 * 4)); CK(hipMalloc(&sink, 16));
    CK(hipMemcpy(wsrc, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(xsrc, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(sink, 0, 16));
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    printf("%d victim workgroups (256 threads, 50 KB LDS), %d iterations, %d launches per row; burner: 512 x 512 threads, v_mfma_f32_32x32x16_bf16 on zero operands, one operand register set\n",
           wgs, iters, reps);
    if (run<8, 2>(wsrc, xsrc, out, sink, s0, s1, wgs, iters, reps)) return 1;
    if (run<32, 2>(wsrc, xsrc, out, sink, s0, s1, wgs, iters, reps)) return 1;
    if (run<72, 2>(wsrc, xsrc, out, sink, s0, s1, wgs, iters, reps)) return 1;
    if (run<96, 2>(wsrc, xsrc, out, sink, s0, s1, wgs, iters, reps)) return 1;
    if (run<112, 2>(wsrc, xsrc, out, sink, s0, s1, wgs, iters, reps)) return 1;
    if (run<120, 2>(wsrc, xsrc, out, sink, s0, s1, wgs, iters, reps)) return 1;
    if (run<72, 1>(wsrc, xsrc, out, sink, s0, s1, wgs, iters, reps)) return 1;
    if (run<120, 1>(wsrc, xsrc, out, sink, s0, s1, wgs, iters, reps)) return 1;
    return 0;
}
