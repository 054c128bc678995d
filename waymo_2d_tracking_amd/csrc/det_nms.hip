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

__global__ __launch_bounds__(256) void nms_sweep_kernel(const unsigned long long* __restrict__ mask, int n, int nb,
                                                        unsigned long long* __restrict__ removed, uint8_t* __restrict__ keep,
                                                        int32_t* __restrict__ n_keep) {
    // 256 threads: every thread owns words of the `removed` bitmap (in LDS); per 64-box tile, wave 0 resolves the
    // intra-tile dependency chain on the diagonal word (v_readlane broadcast per kept box), publishes the tile's
    // `kept` word, then all four waves OR the kept rows into the words of the later tiles (independent, pipelined loads)
    extern __shared__ unsigned long long rem_s[];          // nb words + 1 (kept word of the current tile)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int w = tid; w < nb; w += 256) rem_s[w] = 0ull;
    __syncthreads();
    int total = 0;
    unsigned long long diag_next = 0ull;                   // wave 0: diagonal word of the next tile, prefetched
    if (wave == 0 && lane < n) diag_next = mask[(size_t)lane * nb];
    for (int b = 0; b < nb; ++b) {
        const int cnt = (n - b * 64) < 64 ? (n - b * 64) : 64;
        if (wave == 0) {
            const int i = b * 64 + lane;
            const unsigned long long diag = diag_next;
            if (b + 1 < nb && i + 64 < n) diag_next = mask[(size_t)(i + 64) * nb + b + 1];
            else diag_next = 0ull;
            // scalar chain: `rem` lives in SGPRs (readfirstlane), the branch is a scalar branch
            const unsigned long long r0 = rem_s[b];
            unsigned rlo = __builtin_amdgcn_readfirstlane((unsigned)r0), rhi = __builtin_amdgcn_readfirstlane((unsigned)(r0 >> 32));
            unsigned klo = 0u, khi = 0u;
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
            for (int t = 0; t < cnt && t < 32; ++t) {
                if (!((rlo >> t) & 1u)) {
                    klo |= 1u << t;
                    rlo |= __builtin_amdgcn_readlane(dlo, t);
                    rhi |= __builtin_amdgcn_readlane(dhi, t);
                }
            }
            for (int t = 32; t < cnt; ++t) {
                if (!((rhi >> (t - 32)) & 1u)) {
                    khi |= 1u << (t - 32);
                    rhi |= __builtin_amdgcn_readlane(dhi, t);
                }
            }
            const unsigned long long kept = ((unsigned long long)khi << 32) | klo;
            if (i < n) keep[i] = (kept >> lane) & 1ull;
            if (lane == 0) rem_s[nb] = kept;
            total += __popcll(kept);
        }
        __syncthreads();
        const unsigned long long kept = rem_s[nb];
        // (word, quarter-of-the-rows) work items: 16 independent loads per item, combined with an LDS atomic OR
        const int nw = nb - b - 1;
        for (int e = tid; e < nw * 4; e += 256) {
            const int w = b + 1 + (e >> 2), part = e & 3;
            const unsigned long long* col = mask + (size_t)(b * 64 + part * 16) * nb + w;
            unsigned long long acc = 0ull;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int tt = part * 16 + t;
                const unsigned long long v = (tt < cnt) ? col[(size_t)t * nb] : 0ull;
                acc |= ((kept >> tt) & 1ull) ? v : 0ull;
            }
            if (acc) atomicOr(&rem_s[w], acc);
        }
        __syncthreads();
    }
    if (tid == 0) *n_keep = total;
    (void)removed;
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
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(256), (size_t)(nb + 1) * 8, stream, mask, n, nb, removed, keep_mask, n_keep);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // extern "C"
