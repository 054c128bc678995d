// Greedy hard NMS (torchvision.ops.nms / detectron2 batched_nms restated) on gfx950.
//   pass 1: 64x64 tiles, one wavefront per tile row: lane i tests its box against the 64 boxes of the column
//           tile staged in LDS and packs the result into one 64-bit suppression word (upper triangle only);
//   pass 2: one wavefront sweeps the boxes in score order, 64 at a time: the intra-tile dependency chain runs on
//           the diagonal words with v_readlane broadcasts, the kept rows are OR-ed into the running `removed`
//           bitmap with coalesced 64-bit loads.
// Boxes arrive sorted by descending score; groups (FPN level / class) never suppress each other.
#include "common.h"
#include <cstdlib>
#include <cstring>
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

// Independent problems in one launch (segments = FPN levels of the RPN candidates): problem z = rows [off[z], off[z + 1]),
// blockIdx.z (mask) / blockIdx.x (sweep) selects it; n_seg == 0: one problem over all n rows.
struct NmsSegs { int n_seg; int off[9]; };

__device__ __forceinline__ size_t seg_mask_words(const NmsSegs& sg, int z) {      // mask words in front of problem z
    size_t w = 0;
    for (int q = 0; q < z; ++q) { const size_t m = (size_t)(sg.off[q + 1] - sg.off[q]); w += m * ((m + 63) / 64); }
    return w;
}

template <bool COLMAJOR>     // COLMAJOR: mask[column tile][row] (the column sweep reads a tile's words of consecutive rows coalesced)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes, const int32_t* __restrict__ idxs,
                                                      int n, int nb, float thr, unsigned long long* __restrict__ mask,
                                                      unsigned long long* __restrict__ diag_pred, NmsSegs sg) {
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (sg.n_seg) {
        const int z = blockIdx.z, base = sg.off[z];
        n = sg.off[z + 1] - base;
        nb = (n + 63) / 64;
        if (bi >= nb || bj >= nb) return;
        boxes += base;
        if (idxs) idxs += base;
        diag_pred += base;
        mask += seg_mask_words(sg, z);
    }
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
    if (COLMAJOR) mask[(size_t)bj * n + i] = bits;
    else mask[(size_t)i * nb + bj] = bits;
    if (COLMAJOR && bi == bj) {          // predecessors of box i inside its own tile (the transposed diagonal word)
        unsigned long long pred = 0ull;
        for (int k = 0; k < t; ++k)
            if (cg[k] == ga && iou_gt(cb[k], a, thr)) pred |= 1ull << k;
        diag_pred[i] = pred;
    }
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

// Column sweep (n <= 256 * MAXR): the `removed` word of tile c is gathered just in time from column c of the mask - the
// words of ALL earlier rows are requested one iteration ahead (before the kept bits of the newest tile are known) and
// masked with the kept bitmap afterwards, so no global round trip sits on the tile-to-tile dependency chain (the row
// sweep above pays one per tile: 400 us for 4741 boxes; this one ~1 us per tile).
template <int MAXR>
__global__ __launch_bounds__(256) void nms_sweep_col_kernel(const unsigned long long* __restrict__ maskT,
                                                            const unsigned long long* __restrict__ diag_pred, int n, int nb,
                                                            uint8_t* __restrict__ keep, int32_t* __restrict__ n_keep, NmsSegs sg) {
    if (sg.n_seg) {
        const int z = blockIdx.x, base = sg.off[z];
        n = sg.off[z + 1] - base;
        nb = (n + 63) / 64;
        maskT += seg_mask_words(sg, z);
        diag_pred += base;
        keep += base;
        n_keep += z;
    }
    extern __shared__ unsigned long long sh[];             // keptw[nb] + red
    unsigned long long* keptw = sh;
    unsigned long long* red = sh + nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) red[0] = 0ull;
    unsigned long long pre[MAXR];
#pragma unroll
    for (int j = 0; j < MAXR; ++j) pre[j] = 0ull;
    unsigned long long pred_next = (wave == 0 && lane < n) ? diag_pred[lane] : 0ull;   // tile 0
    int total = 0;
    __syncthreads();
    for (int c = 0; c < nb; ++c) {
        // 1. OR of column c over the kept rows of the earlier tiles (words prefetched during the previous iteration)
        unsigned long long acc = 0ull;
#pragma unroll
        for (int j = 0; j < MAXR; ++j) {
            const int i = tid + 256 * j;
            if (i < c * 64 && ((keptw[i >> 6] >> (i & 63)) & 1ull)) acc |= pre[j];
        }
        if (acc) atomicOr(&red[0], acc);
        __syncthreads();
        // 2. request column c + 1 for every row up to and including tile c (its kept bits are decided below)
        const unsigned long long pred = pred_next;
        if (c + 1 < nb) {
            const unsigned long long* col = maskT + (size_t)(c + 1) * n;
#pragma unroll
            for (int j = 0; j < MAXR; ++j) {
                const int i = tid + 256 * j;
                pre[j] = (i < (c + 1) * 64 && i < n) ? col[i] : 0ull;
            }
            const int id = (c + 1) * 64 + lane;
            pred_next = (wave == 0 && id < n) ? diag_pred[id] : 0ull;
        }
        // 3. wave 0: kept bits of the tile = the unique fixed point of kept[k] = alive[k] & !(pred[k] & kept), reached by
        //    iterating from kept = alive (one ballot per round; rounds = depth of the suppression chains, typically 2 - 6)
        if (wave == 0) {
            const int cnt = (n - c * 64) < 64 ? (n - c * 64) : 64;
            const unsigned long long valid = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
            const unsigned long long alive = ~red[0] & valid;
            unsigned long long kept = alive;
            for (int round = 0; round < 64; ++round) {
                const unsigned long long blocked = __ballot((pred & kept) != 0ull);
                const unsigned long long nk = alive & ~blocked;
                if (nk == kept) break;
                kept = nk;
            }
            const int i = c * 64 + lane;
            if (i < n) keep[i] = (kept >> lane) & 1ull;
            if (lane == 0) { keptw[c] = kept; red[0] = 0ull; }
            total += __popcll(kept);
        }
        __syncthreads();
    }
    if (tid == 0) *n_keep = total;
}

// First `cap` entries of a keep mask (in its order) as fixed-size index list: out_idx[k] = order ? order[i] : i of the k-th i
// with keep[i] (and valid[i]); unused slots = -1; *count = min(total, cap).  One workgroup: ballot prefix sums, the running
// base carried across 256-entry chunks.  Lets the proposal / detection lists keep a static shape (no host round trip for the
// number of survivors).
__global__ __launch_bounds__(256) void select_kept_kernel(const uint8_t* __restrict__ keep, const uint8_t* __restrict__ valid,
                                                          const int64_t* __restrict__ order, int n, int cap,
                                                          int64_t* __restrict__ out_idx, int32_t* __restrict__ count) {
    __shared__ int wave_sum[4];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    for (int k = tid; k < cap; k += 256) out_idx[k] = -1;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + tid;
        const bool f = i < n && keep[i] && (!valid || valid[i]);
        const unsigned long long m = __ballot(f);
        if (lane == 0) wave_sum[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_sum[w];
        const int k = off + __popcll(m & ((1ull << lane) - 1ull));
        if (f && k < cap) out_idx[k] = order ? order[i] : (int64_t)i;
        __syncthreads();
        if (tid == 0) base_s += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        __syncthreads();
        if (base_s >= cap) break;
    }
    if (tid == 0) *count = base_s < cap ? base_s : cap;
}

}  // namespace

extern "C" {

int wd_select_kept(const uint8_t* keep_mask, const uint8_t* valid, const int64_t* order, int n, int cap, int64_t* out_idx,
                   int32_t* count, void* stream) {
    WT_TRY(wt::ensure_device());
    if (n < 0 || cap <= 0 || !out_idx || !count || (n > 0 && !keep_mask)) { wt::set_error("wd_select_kept: bad argument"); return WT_ERR_INVALID; }
    hipLaunchKernelGGL(select_kept_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, keep_mask, valid, order, n, cap, out_idx, count);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

size_t wd_nms_workspace(int n) {
    const size_t nb = (size_t)(n + 63) / 64;
    return wt::align_up((size_t)(n > 0 ? n : 1) * nb * 8) + wt::align_up(nb * 8 + 8) + wt::align_up((size_t)(n > 0 ? n : 1) * 8) + 256;
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
    unsigned long long* diag_pred = cv.take<unsigned long long>((size_t)n);
    constexpr int kMaxR = 24;                               // column sweep: up to 6144 boxes
    const char* mode = getenv("WD_NMS_SWEEP");              // experiments: "row" forces the row sweep
    if (n <= 256 * kMaxR && !(mode && strcmp(mode, "row") == 0)) {
        hipLaunchKernelGGL(nms_mask_kernel<true>, dim3(nb, nb), dim3(64), 0, stream, (const float4*)boxes, idxs, n, nb,
                           iou_threshold, mask, diag_pred, NmsSegs{});
        hipLaunchKernelGGL(nms_sweep_col_kernel<kMaxR>, dim3(1), dim3(256), (size_t)(nb + 1) * 8, stream, mask, diag_pred, n, nb,
                           keep_mask, n_keep, NmsSegs{});
    } else {
        hipLaunchKernelGGL(nms_mask_kernel<false>, dim3(nb, nb), dim3(64), 0, stream, (const float4*)boxes, idxs, n, nb,
                           iou_threshold, mask, diag_pred, NmsSegs{});
        hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(256), (size_t)(nb + 1) * 8, stream, mask, n, nb, removed, keep_mask,
                           n_keep);
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

/* The same NMS on n_seg independent row ranges [seg_offsets[z], seg_offsets[z + 1]) in ONE pair of launches (RPN: one range per
 * FPN level, each sorted by descending score): the suppression chains of the ranges run in parallel workgroups instead of one
 * 75-tile chain.  seg_offsets: host array of n_seg + 1 ints; n_keep: n_seg device ints; workspace as for n rows. */
int wd_nms_segmented_f32(const float* boxes, const int32_t* idxs, const int32_t* seg_offsets, int n_seg, float iou_threshold,
                         uint8_t* keep_mask, int32_t* n_keep, void* workspace, size_t workspace_bytes, void* stream_) {
    WT_TRY(wt::ensure_device());
    hipStream_t stream = (hipStream_t)stream_;
    if (n_seg < 1 || n_seg > 8 || !seg_offsets || !boxes || !keep_mask || !n_keep || ((uintptr_t)boxes & 15)) {
        wt::set_error("wd_nms_segmented_f32: bad argument (1..8 segments)");
        return WT_ERR_INVALID;
    }
    constexpr int kMaxR = 24;
    NmsSegs sg{};
    sg.n_seg = n_seg;
    int max_n = 0;
    for (int z = 0; z <= n_seg; ++z) sg.off[z] = seg_offsets[z];
    for (int z = 0; z < n_seg; ++z) {
        const int m = sg.off[z + 1] - sg.off[z];
        if (m < 0 || m > 256 * kMaxR) { wt::set_error("wd_nms_segmented_f32: segment %d has %d rows (0..%d)", z, m, 256 * kMaxR); return WT_ERR_INVALID; }
        max_n = m > max_n ? m : max_n;
    }
    const int n = sg.off[n_seg] - sg.off[0];
    if (n <= 0) { WT_HIP(hipMemsetAsync(n_keep, 0, sizeof(int32_t) * n_seg, stream)); return WT_OK; }
    if (sg.off[0] != 0 || !workspace || workspace_bytes < wd_nms_workspace(n)) {
        wt::set_error("wd_nms_segmented_f32: offsets must start at 0 / workspace too small");
        return WT_ERR_CAPACITY;
    }
    const int nbm = (max_n + 63) / 64;
    const uintptr_t mis = (uintptr_t)workspace & 255;
    wt::Carver cv((char*)workspace + (mis ? 256 - mis : 0));
    unsigned long long* mask = cv.take<unsigned long long>((size_t)n * ((n + 63) / 64));
    (void)cv.take<unsigned long long>((size_t)((n + 63) / 64) + 1);
    unsigned long long* diag_pred = cv.take<unsigned long long>((size_t)n);
    if (nbm > 0) {
        hipLaunchKernelGGL(nms_mask_kernel<true>, dim3(nbm, nbm, n_seg), dim3(64), 0, stream, (const float4*)boxes, idxs, n, nbm,
                           iou_threshold, mask, diag_pred, sg);
        hipLaunchKernelGGL(nms_sweep_col_kernel<kMaxR>, dim3(n_seg), dim3(256), (size_t)(nbm + 1) * 8, stream, mask, diag_pred, n, nbm,
                           keep_mask, n_keep, sg);
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // extern "C"
