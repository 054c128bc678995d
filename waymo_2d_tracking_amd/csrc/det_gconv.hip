// Grouped 3x3 convolution with 8 channels per group (res2 of ResNeXt-152 32x8d: 256 channels, 32 groups; job.log:357-401),
// stride 1, pad 1, fused FrozenBN affine + ReLU, NHWC.
//
// With K = 8 x 9 = 72 per output channel and only 8 output channels per group an MFMA tile is half empty (M = 8 of 16 rows), and
// on this chip the f32 matrix rate equals the packed-f32 vector rate anyway - so this shape runs on the VECTOR units:
//   * a thread owns TWO adjacent output channels and keeps their 2 x 72 weights in registers for the whole kernel
//     (persistent workgroups; 144 VGPRs), one v_pk_fma_f32 per (tap, input channel) updates both;
//   * a workgroup = 128 channels (16 groups) x an 8 x 8 pixel tile; the 10 x 10 x 128-channel input patch (50 KB) is staged in
//     LDS once per tile; wave w computes pixels w, w + 4, ...: its 64 lanes = the 64 channel pairs, so the 8 input channels of a
//     group are read by 4 lanes from the same 32 bytes (LDS broadcast) and a wave's 18 reads per pixel are 512 contiguous bytes;
//   * two workgroups per CU: one loads its next patch while the other computes.
// Work per pixel and wave: 18 ds_read_b128 + 72 v_pk_fma_f32; the shape is 2.83 G fma = 36 us at the vector peak.
// Measured on MI355X (256 ch, 320 x 480): 137-150 us (the MFMA gather kernel it replaces: 187-208 us; MIOpen's grouped conv: 391 us).
// Ablations: FMAs 46 us, LDS reads + loop skeleton 56 us (5.5 GB of ds_read_b128 = 35 us at the LDS peak: the 4-lane broadcast of a
// group's channels costs full LDS cycles), output stores 18 us, patch loads 30 us (overlapped by the second workgroup of the CU).
// The next step would halve the LDS reads (4-pixel strips with the input channels split over lane pairs: 72 weight registers).
#include "common.h"
#include <cstdlib>
#include "../../include/waymodet.h"

namespace {

namespace gc {
constexpr int CH = 128;                 // channels per workgroup (16 groups of 8)
constexpr int TS = 8;                   // tile side
constexpr int PS = TS + 2;              // patch side
constexpr int PATCH_F = PS * PS * CH;   // floats
}  // namespace gc

typedef float float2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 2) void grouped_conv3x3_c8_kernel(const float* __restrict__ x, const float* __restrict__ packed,
                                                                    const float* __restrict__ scale, const float* __restrict__ bias,
                                                                    int relu, int batch, int H, int W, int C, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float patch[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int halves = C / gc::CH;
    const int half = blockIdx.x % halves;                       // fixed per workgroup: its weights stay in registers
    const int c0 = half * gc::CH;
    const int tiles_x = (W + gc::TS - 1) / gc::TS, tiles_y = (H + gc::TS - 1) / gc::TS;
    const int ntiles = batch * tiles_y * tiles_x;
    const int co = c0 + 2 * lane;                               // this thread's output channel pair
    const int g = co >> 3, col = co & 7;                        // group, first channel inside the group (0, 2, 4, 6)
    // weights: packed[((g * 9 + tap) * 8 + ci) * 8 + co_local] (wd_deform_pack_weight) -> w[tap][ci] = (co, co + 1)
    float2v wr[9][8];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const float2 t = *reinterpret_cast<const float2*>(packed + ((size_t)(g * 9 + k) * 8 + ci) * 8 + col);
            wr[k][ci] = (float2v){t.x, t.y};
        }
    const float2v sc = scale ? (float2v){scale[co], scale[co + 1]} : (float2v){1.f, 1.f};
    const float2v bi = bias ? (float2v){bias[co], bias[co + 1]} : (float2v){0.f, 0.f};
    const int gl = lane >> 2;                                   // group inside the workgroup's 16: LDS channel offset 8 * gl
    // Patch fill by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes land lane-linear at M0, no staging registers).  One
    // instruction = 1 KB = two patch pixels (2 j, 2 j + 1) x 128 channels; the patch row holds 10 pixels, so both lie in the same
    // row: source = scalar base of pixel 2 j + ONE per-lane constant (second pixel: + C floats).  50 instructions per patch, wave m
    // issues j = m, m + 4, ...
    constexpr int NI = gc::PS * gc::PS / 2;                     // 50
    constexpr int NW = (NI + 3) / 4;                            // 13 per wave
    const int lane_off = ((lane >> 5) * C + (lane & 31) * 4) * 4;        // bytes
    const char* xb = reinterpret_cast<const char*>(x);
    const int step = gridDim.x / halves;
    for (int t = blockIdx.x / halves; t < ntiles; t += step) {
        const int tn = t / (tiles_y * tiles_x), tr = t - tn * tiles_y * tiles_x;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int y0 = ty * gc::TS - 1, x0 = tx * gc::TS - 1;
        const char* origin = xb + ((((long)tn * H + y0) * W + x0) * C + c0) * 4;
        const bool inner = y0 >= 0 && x0 >= 0 && y0 + gc::PS <= H && x0 + gc::PS <= W;
        __syncthreads();                                        // every wave is done reading the previous patch
        {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int j = wave + 4 * i;                     // wave-uniform
                if (j >= NI) continue;
                const int py = (2 * j) / gc::PS, px = 2 * j - py * gc::PS;
                const char* sbase = origin + ((long)py * W + px) * C * 4;
                const unsigned dst = (unsigned)(j * 1024);
                const int iy = y0 + py, ix = x0 + px + (lane >> 5);
                if (inner || (iy >= 0 && iy < H && ix >= 0 && ix < W)) {
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(lane_off), "s"(sbase), "s"(dst) : "memory");
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<char*>(patch) + dst + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0070);                     // vmcnt(0) + lgkmcnt(0): this wave's part of the patch has landed
        __syncthreads();
        // compute: wave w takes pixels w, w + 4, ... of the tile.  Per pixel 9 taps x (2 ds_read_b128 + 8 v_pk_fma_f32); the reads of
        // kernel row kh + 1 (or of the next pixel's row 0) are issued BEFORE the 24 FMAs of row kh (sched_barrier pins that order),
        // so the vector unit only waits for LDS at the very first row of a tile.
        constexpr int npix = gc::TS * gc::TS;
        float4 cur[3][2], nxt[3][2];
        auto read_row = [&](int p, int kh, float4 (&dst)[3][2]) {
            const float* src = patch + (((p >> 3) + kh) * gc::PS + (p & 7)) * gc::CH + gl * 8;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                dst[kw][0] = *reinterpret_cast<const float4*>(src + kw * gc::CH);
                dst[kw][1] = *reinterpret_cast<const float4*>(src + kw * gc::CH + 4);
            }
        };
        read_row(wave, 0, cur);
        for (int p = wave; p < npix; p += 4) {
            float2v acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f}, acc3 = {0.f, 0.f};     // four independent FMA chains
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                if (kh < 2) read_row(p, kh + 1, nxt);
                else read_row((p + 4 < npix) ? p + 4 : p, 0, nxt);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int k = kh * 3 + kw;
                    acc0 += wr[k][0] * cur[kw][0].x; acc1 += wr[k][1] * cur[kw][0].y;
                    acc2 += wr[k][2] * cur[kw][0].z; acc3 += wr[k][3] * cur[kw][0].w;
                    acc0 += wr[k][4] * cur[kw][1].x; acc1 += wr[k][5] * cur[kw][1].y;
                    acc2 += wr[k][6] * cur[kw][1].z; acc3 += wr[k][7] * cur[kw][1].w;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) { cur[kw][0] = nxt[kw][0]; cur[kw][1] = nxt[kw][1]; }
            }
            const int oy = ty * gc::TS + (p >> 3), ox = tx * gc::TS + (p & 7);
            if (oy < H && ox < W) {
                float2v v = ((acc0 + acc1) + (acc2 + acc3)) * sc + bi;
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
                *reinterpret_cast<float2*>(y + (((size_t)tn * H + oy) * W + ox) * C + co) = make_float2(v.x, v.y);
            }
        }
    }
}

}  // namespace

int wt::victim_regs_grouped_conv() {
    hipFuncAttributes at{};
    if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(grouped_conv3x3_c8_kernel)) != hipSuccess) return 0;
    return (at.numRegs + 7) / 8 * 8;
}

// Plain grouped 3x3 conv, 8 channels per group, stride 1, pad 1 (called by wd_deform_conv3x3_f32 when offset == NULL).
int wd_grouped_conv3x3_c8_launch(const float* x, const float* packed_weight, const float* scale, const float* bias, int relu,
                                 int batch, int h, int w, int c, hipStream_t stream, float* y) {
    const int n_cu = wt::device_cus() > 0 ? wt::device_cus() : 256;
    static wt::OncePerDevice attr;
    const int dev = wt::device_index();
    const size_t smem = (size_t)gc::PATCH_F * sizeof(float);
    if (attr.needed(dev)) {
        WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(grouped_conv3x3_c8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)smem));
        attr.mark(dev);
    }
    const int halves = c / gc::CH;
    const long ntiles = (long)batch * ((h + gc::TS - 1) / gc::TS) * ((w + gc::TS - 1) / gc::TS);
    long per_half = (2L * n_cu + halves - 1) / halves;              // two workgroups per CU in total
    if (per_half > ntiles) per_half = ntiles;
    if (per_half < 1) per_half = 1;
    hipLaunchKernelGGL(grouped_conv3x3_c8_kernel, dim3((unsigned)(per_half * halves)), dim3(256), smem, stream, x, packed_weight,
                       scale, bias, relu, batch, h, w, c, y);
    WT_HIP(hipGetLastError());
    return WT_OK;
}
