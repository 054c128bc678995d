// Microbenchmark (tools only): issue rate of ds_add_f32 on gfx950 for the address patterns a fused deformable backward would produce.
//   pattern 0: 64 consecutive floats (lane = channel)
//   pattern 1: 4 rows x 16 consecutive floats, row pitch 33 floats apart x random row (MFMA 16x16 accumulator layout: 4 pixels x 16 channels)
//   pattern 2: as 1 with all 4 rows equal (same-address collisions inside one instruction)
//   pattern 3: ds_read_b32 with pattern 1 (reference: plain LDS read rate)
// hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N = 8192, ITER = 256;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const int* rows, int pitch) {
    __shared__ float acc[N];
    for (int i = threadIdx.x; i < N; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    for (int it = 0; it < ITER; ++it) {
        int a;
        if (MODE == 0) a = ((rows[it * 4 + wave] * pitch) + lane) % N;
        else if (MODE == 2) a = (rows[(it * 4 + wave) * 4] * pitch + (lane & 15)) % N;
        else a = (rows[(it * 4 + wave) * 4 + (lane >> 4)] * pitch + (lane & 15)) % N;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = (a + u * 16 * pitch) % N;
            if (MODE == 3) s += acc[b]; else atomicAdd(&acc[b], 1.0f);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) out[blockIdx.x * N + i] = acc[i] + s;
}
int main() {
    float* out; int* rows;
    const int nb = 256 * 4;
    hipMalloc(&out, sizeof(float) * N * nb);
    std::vector<int> h(ITER * 16);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x >> 8) % 196; }
    hipMalloc(&rows, h.size() * 4);
    hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pitch : {32, 33, 36, 40}) {
        for (int mode = 0; mode < 4; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<nb, 256>>>(out, rows, pitch);
                if (mode == 1) k<1><<<nb, 256>>>(out, rows, pitch);
                if (mode == 2) k<2><<<nb, 256>>>(out, rows, pitch);
                if (mode == 3) k<3><<<nb, 256>>>(out, rows, pitch);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            // per CU: nb / 256 workgroups in sequence (if 1 WG per CU at a time: LDS 32 KB -> up to 5 resident; report raw)
            const double instr = (double)nb * 4 * ITER * 8;          // wave-level LDS instructions in total
            printf("pitch %2d mode %d: %8.3f ms  %.1f ns per wave instruction per CU (256 CUs)  = %.1f cycles at 2.4 GHz\n", pitch, mode, ms,
                   ms * 1e6 / (instr / 256), ms * 1e6 / (instr / 256) * 2.4);
        }
    }
    return 0;
}
