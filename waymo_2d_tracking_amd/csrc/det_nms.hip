// Greedy hard NMS (torchvision.ops.nms / detectron2 batched_nms restated) on gfx950.
//   pass 1: 64x64 tiles, one wavefront per tile row: lane i tests its box against the 64 boxes of the column
//           tile staged in LDS and packs the result into one 64-bit suppression word (upper triangle only);
//   pass 2: one wavefront sweeps the boxes in score order, 64 at a time: the intra-tile dependency chain runs on
//           the diagonal words with v_readlane broadcasts, the kept rows are OR-ed into the running `removed`
//           bitmap with coalesced 64-bit loads.
// Boxes arrive sorted by descending score; groups (FPN level / class) never suppress each other.
#include "common.h"
#include "../../include/waymodet.h"

namespace {

__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float thr) {
    const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float inter = width * height;
    const float sa = (a.z - a.x) * (a.w - a.y);
    const float sb = (b.z - b.x) * (b.w - b.y);
    return (inter / (sa + sb - inter)) > thr;
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes, const int32_t* __restrict__ idxs,
                                                      int n, int nb, float thr, unsigned long long* __restrict__ mask) {
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    __shared__ float4 cb[64];
    __shared__ int cg[64];
    const int t = threadIdx.x;
    const int j0 = bj * 64;
    if (j0 + t < n) { cb[t] = boxes[j0 + t]; cg[t] = idxs ? idxs[j0 + t] : 0; }
    __syncthreads();
    const int i = bi * 64 + t;
    if (i >= n) return;
    const float4 a = boxes[i];
    const int ga = idxs ? idxs[i] : 0;
    unsigned long long bits = 0ull;
    const int cnt = (n - j0) < 64 ? (n - j0) : 64;
    const int start = (bi == bj) ? t + 1 : 0;
    for (int k = start; k < cnt; ++k)
        if (cg[k] == ga && iou_gt(a, cb[k], thr)) bits |= 1ull << k;
    mask[(size_t)i * nb + bj] = bits;
}

__global__ __launch_bounds__(64) void nms_sweep_kernel(const unsigned long long* __restrict__ mask, int n, int nb,
                                                       unsigned long long* __restrict__ removed, uint8_t* __restrict__ keep,
                                                       int32_t* __restrict__ n_keep) {
    const int lane = threadIdx.x;
    for (int w = lane; w < nb; w += 64) removed[w] = 0ull;
    __syncthreads();
    int total = 0;
    for (int b = 0; b < nb; ++b) {
        const int i = b * 64 + lane;
        const unsigned long long diag = (i < n) ? mask[(size_t)i * nb + b] : 0ull;
        unsigned long long rem = removed[b];
        const int cnt = (n - b * 64) < 64 ? (n - b * 64) : 64;
        unsigned long long kept = 0ull;
        const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
        for (int t = 0; t < cnt; ++t) {
            if (!((rem >> t) & 1ull)) {
                kept |= 1ull << t;
                const unsigned lo = __builtin_amdgcn_readlane(dlo, t), hi = __builtin_amdgcn_readlane(dhi, t);
                rem |= ((unsigned long long)hi << 32) | lo;
            }
        }
        if (i < n) keep[i] = (kept >> lane) & 1ull;
        total += __popcll(kept);
        // OR the kept rows into the words of the later tiles: every lane owns words w, w+64, ...; the 64 row loads
        // per word are independent (suppressed rows are masked out, not branched around) so they pipeline
        for (int w = b + 1 + lane; w < nb; w += 64) {
            unsigned long long acc = removed[w];
            const unsigned long long* col = mask + (size_t)(b * 64) * nb + w;
#pragma unroll 16
            for (int t = 0; t < 64; ++t) {
                const unsigned long long v = (t < cnt) ? col[(size_t)t * nb] : 0ull;
                acc |= ((kept >> t) & 1ull) ? v : 0ull;
            }
            removed[w] = acc;
        }
        __syncthreads();
    }
    if (lane == 0) *n_keep = total;
}

}  // namespace

extern "C" {

size_t wd_nms_workspace(int n) {
    const size_t nb = (size_t)(n + 63) / 64;
    return wt::align_up((size_t)(n > 0 ? n : 1) * nb * 8) + wt::align_up(nb * 8 + 8) + 256;
}

int wd_nms_sorted_f32(const float* boxes, const int32_t* idxs, int n, float iou_threshold, uint8_t* keep_mask,
                      int32_t* n_keep, void* workspace, size_t workspace_bytes, void* stream_) {
    WT_TRY(wt::ensure_device());
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) {
        WT_HIP(hipMemsetAsync(n_keep, 0, sizeof(int32_t), stream));
        return WT_OK;
    }
    if (!workspace || workspace_bytes < wd_nms_workspace(n)) {
        wt::set_error("wd_nms_sorted_f32: workspace too small (%zu < %zu)", workspace_bytes, wd_nms_workspace(n));
        return WT_ERR_CAPACITY;
    }
    if (((uintptr_t)boxes & 15) != 0) { wt::set_error("boxes must be 16-byte aligned"); return WT_ERR_INVALID; }
    const int nb = (n + 63) / 64;
    const uintptr_t mis = (uintptr_t)workspace & 255;
    wt::Carver cv((char*)workspace + (mis ? 256 - mis : 0));
    unsigned long long* mask = cv.take<unsigned long long>((size_t)n * nb);
    unsigned long long* removed = cv.take<unsigned long long>((size_t)nb + 1);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb), dim3(64), 0, stream, (const float4*)boxes, idxs, n, nb,
                       iou_threshold, mask);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), 0, stream, mask, n, nb, removed, keep_mask, n_keep);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // extern "C"
