// Micro-probe: cycles per ds_read_b128 for a given lane -> LDS byte address map (experiments only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
__global__ void probe(const unsigned* addr, int iters, unsigned long long* out, float* sink, unsigned base, unsigned span) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 39936; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    const unsigned a = addr[threadIdx.x & 63];
    float4 acc = make_float4(0, 0, 0, 0);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        float4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned ad = base + ((a + 256u * ((i * 16 + j) & 127)) % span);      // + k * 256 B: same banks
            asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(ad));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
static int slot(int u, int q) { return (u << 3) + (q ^ ((u >> 1) & 7)); }
static int row_pixel(int r, int m) { int blk = r >> 2, i = r & 3; int y = m + 4 * (blk >> 1); int x = 2 * i + ((blk == 1 || blk == 2) ? 1 : 0); return y * 8 + x; }
int main() {
    std::vector<std::pair<const char*, std::vector<unsigned>>> pats;
    { std::vector<unsigned> a(64); for (int l = 0; l < 64; ++l) a[l] = l * 16; pats.push_back({"linear lane*16", a}); }
    { std::vector<unsigned> a(64); for (int l = 0; l < 64; ++l) a[l] = (l & 15) * 16; pats.push_back({"16 addresses broadcast x4", a}); }
    { std::vector<unsigned> a(64); for (int l = 0; l < 64; ++l) a[l] = l * 256; pats.push_back({"all same bank (64-way)", a}); }
    { std::vector<unsigned> a(64); for (int l = 0; l < 64; ++l) a[l] = (l & 31) * 16 + (l >> 5) * 512; pats.push_back({"2-way halves", a}); }
    for (int kh = 0; kh < 3; kh += 2) for (int kw = 0; kw < 3; kw += 1) for (int corner = 0; corner < 4; corner += 3) {
        std::vector<unsigned> a(64);
        for (int l = 0; l < 64; ++l) {
            int r16 = l & 15, kq = l >> 4, m = 1;
            int p = row_pixel(r16, m), y = p >> 3, x = p & 7;
            int u = (y + kh + 2) * 14 + (x + kw + 2) + (corner & 1) + (corner >> 1) * 14;
            a[l] = ((unsigned)(slot(u, 0) << 4) ^ ((unsigned)(2 * kq) << 4));
        }
        char* nm = (char*)malloc(64); snprintf(nm, 64, "corner%d tap(%d,%d) zero offset", corner, kh, kw);
        pats.push_back({nm, a});
    }
    { std::vector<unsigned> a(64); for (int l = 0; l < 64; ++l) { int p = row_pixel(l & 15, 1); a[l] = (p * 9 + 4) * 16; } pats.push_back({"table entry", a}); }
    { std::vector<unsigned> a(64); for (int l = 0; l < 64; ++l) { int p = row_pixel(l & 15, 1); a[l] = (p * 10 + 4) * 16; } pats.push_back({"table entry pitch 10", a}); }
    { std::vector<unsigned> a(64); for (int l = 0; l < 64; ++l) { int p = row_pixel(l & 15, 1); a[l] = (p * 9 + 4 + (p >> 5)) * 16; } pats.push_back({"table entry +1 for rows>=4", a}); }
    unsigned* d_addr; unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_addr, 256); hipMalloc(&d_out, 8 * 256); hipMalloc(&d_sink, 4 * 256 * 256);
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 159744);
    for (auto& p : pats) {
        hipMemcpy(d_addr, p.second.data(), 256, hipMemcpyHostToDevice);
        for (unsigned base = 0; base <= 98304; base += 32768) {
            for (int big = 0; big < 2; ++big) {
                if (!big && base) continue;
                hipLaunchKernelGGL(probe, dim3(1), dim3(1024), big ? 159744 : 65536, 0, d_addr, iters, d_out, d_sink, base, 32768u);
                hipDeviceSynchronize();
                unsigned long long c; hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
                printf("%-34s lds %3d KB base %6u: %.2f\n", p.first, big ? 156 : 64, base, (double)c / (iters * 16.0 * 16));
            }
        }
    }
    return 0;
}
