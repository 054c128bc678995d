// The detector's data-dependent tail with static shapes: one kernel per step instead of chains of tiny sort / index /
// elementwise launches, no host round trip anywhere (the whole frame is captured as one hipGraph).
//
//   RPN (detectron2 RPN.predict_proposals / find_top_rpn_proposals):
//     rpn_topk_stage_kernel  x2-3  per-level top-k of the objectness logits by block-wise bitonic selection
//                                  (8192 keys per workgroup in LDS, the best k survive to the next stage); the last
//                                  stage decodes the k anchors (apply_deltas + clip) and flags empty boxes
//     sort_candidates_kernel x1    stable descending sort of all levels' candidates (one workgroup) + gather
//     [det_nms.hip: suppression bitmask + column sweep]
//     gather_kept_kernel     x1    first `post` kept candidates -> zero-padded proposal list + device-side count
//   box inference (fast_rcnn_inference_single_image):
//     sort_candidates_kernel x1    (box, class) candidates: 3-stage score average, finite / threshold tests, clip, sort
//     [det_nms.hip]
//     gather_kept_kernel     x1    first `topk` kept -> boxes, scores, classes + count
//   wire format (Detectron2Det.predict + COCODetection.load_prediction): wire_kernel x1
//
// Keys are 64-bit: (order-preserving bits of the float score) << 32 | ~position, so that a descending sort yields
// score-descending, position-ascending order = a stable descending sort.  Built with -ffp-contract=off: the float
// arithmetic is the torch sequence operation by operation.
#include "common.h"
#include <cstdlib>
#include <cstring>
#include "../../include/waymodet.h"
#include "det_boxmath.h"

namespace {

constexpr int kChunk = 8192;          // keys one workgroup sorts in LDS (64 KB)
constexpr int kThreads = 1024;

__device__ __forceinline__ unsigned int orderable(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ unsigned long long make_key(float score, unsigned int pos) {
    if (score == 0.f) score = 0.f;                         // -0.0 and +0.0 compare equal in torch's sort: one key for both
    return ((unsigned long long)orderable(score) << 32) | (unsigned long long)(0xffffffffu - pos);
}

// Descending bitonic sort of s[0..p) (p a power of two, >= 2), all kThreads threads of the workgroup.
__device__ __forceinline__ void bitonic_desc(unsigned long long* s, int p) {
    for (int size = 2; size <= p; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (p >> 1); t += kThreads) {
                const int i = 2 * t - (t & (stride - 1));
                const int j = i + stride;
                const unsigned long long a = s[i], b = s[j];
                const bool desc = (i & size) == 0;
                if ((a < b) == desc) { s[i] = b; s[j] = a; }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int pow2_at_least(int n) {
    int p = 2;
    while (p < n) p <<= 1;
    return p;
}

constexpr int kMaxLevels = 8;

struct TopkStage {
    int n_levels, k;
    int chunk_begin[kMaxLevels + 1];          // workgroup index range of every level
    int in_n[kMaxLevels];                     // keys (stage 0: logits) of the level entering this stage
    long long in_off[kMaxLevels];             // offset of the level in the key buffer (stages > 0)
    long long out_off[kMaxLevels];            // offset in the output key buffer / (final) the candidate row of the level
    const float* logits[kMaxLevels];          // stage 0 input
    const float* deltas[kMaxLevels];          // final stage: (n, 4) per level
    const float* anchors[kMaxLevels];
};

// One selection stage.  first: keys are built from the logits; final: exactly one chunk per level, the sorted best k
// are decoded into candidate rows.
__global__ __launch_bounds__(kThreads) void rpn_topk_stage_kernel(TopkStage st, int first, int final,
                                                                 const unsigned long long* __restrict__ keys_in,
                                                                 unsigned long long* __restrict__ keys_out, float clip_h,
                                                                 float clip_w, float scale_clamp, float4* __restrict__ out_boxes,
                                                                 float* __restrict__ out_scores, int* __restrict__ out_group,
                                                                 unsigned char* __restrict__ out_valid) {
    extern __shared__ unsigned long long skeys[];
    int level = 0;
    while (level + 1 < st.n_levels && (int)blockIdx.x >= st.chunk_begin[level + 1]) ++level;
    const int chunk = blockIdx.x - st.chunk_begin[level];
    const int n = st.in_n[level];
    const int lo = chunk * kChunk;
    const int len = (n - lo) < kChunk ? (n - lo) : kChunk;
    const int p = pow2_at_least(len);
    for (int t = threadIdx.x; t < p; t += kThreads) {
        unsigned long long key = 0ull;
        if (t < len)
            key = first ? make_key(st.logits[level][lo + t], (unsigned int)(lo + t)) : keys_in[st.in_off[level] + lo + t];
        skeys[t] = key;
    }
    bitonic_desc(skeys, p);
    const int kout = len < st.k ? len : st.k;
    if (!final) {
        for (int t = threadIdx.x; t < kout; t += kThreads) keys_out[st.out_off[level] + (long long)chunk * st.k + t] = skeys[t];
        return;
    }
    const float4* deltas = reinterpret_cast<const float4*>(st.deltas[level]);
    const float4* anchors = reinterpret_cast<const float4*>(st.anchors[level]);
    for (int t = threadIdx.x; t < kout; t += kThreads) {
        const unsigned long long key = skeys[t];
        const unsigned int idx = 0xffffffffu - (unsigned int)(key & 0xffffffffull);
        const float4 b = wd::decode_box(deltas[idx], anchors[idx], 1.0f, 1.0f, 1.0f, 1.0f, scale_clamp, clip_w, clip_h);
        const long long row = st.out_off[level] + t;
        out_boxes[row] = b;
        out_scores[row] = from_orderable((unsigned int)(key >> 32));
        const bool ok = (b.z - b.x) > 0.f && (b.w - b.y) > 0.f;     // find_top_rpn_proposals: empty boxes are dropped
        out_group[row] = ok ? level : -1;
        out_valid[row] = ok ? 1 : 0;
    }
}

// Candidate sources of sort_candidates_kernel.
struct CandRpn {                       // rows already hold boxes / scores / groups / valid flags
    const float4* boxes; const float* scores; const int* group; const unsigned char* valid;
    const unsigned char* valid2;       // optional second flag (the per-level NMS keep mask), ANDed into valid
};
struct CandBox {                       // (row, class) pairs of the last cascade stage
    const float4* boxes;               // (r, 4) unclipped
    const float* s0; const float* s1; const float* s2;   // (r, nc + 1) softmax of the three stages
    const int* n_valid;                // device count of real rows (nullable)
    int nc; float score_thresh, clip_h, clip_w;
};

__device__ __forceinline__ bool finite4(const float4 b) { return isfinite(b.x) && isfinite(b.y) && isfinite(b.z) && isfinite(b.w); }

// One workgroup: stable descending sort of n <= kChunk candidates by score, gathered into sorted order.
// MODE 0: RPN candidates.  MODE 1: box-head candidates i = row * nc + class (built on the fly).
template <int MODE>
__global__ __launch_bounds__(kThreads) void sort_candidates_kernel(CandRpn a, CandBox c, int n, float4* __restrict__ out_boxes,
                                                                  float* __restrict__ out_scores, int* __restrict__ out_group,
                                                                  unsigned char* __restrict__ out_valid,
                                                                  long long* __restrict__ out_order) {
    extern __shared__ unsigned long long skeys[];
    const int p = pow2_at_least(n);
    for (int t = threadIdx.x; t < p; t += kThreads) {
        unsigned long long key = 0ull;
        if (t < n) {
            float s;
            if (MODE == 0) {
                s = a.scores[t];
            } else {
                const int row = t / c.nc, cls = t - row * c.nc;
                const int w = c.nc + 1;
                bool row_ok = finite4(c.boxes[row]) && (!c.n_valid || row < *c.n_valid);
                float mine = 0.f;
                for (int q = 0; q < w; ++q) {               // (s0 + s1 + s2) * (1 / 3): the torch sequence
                    const float v = ((c.s0[row * w + q] + c.s1[row * w + q]) + c.s2[row * w + q]) * (float)(1.0 / 3);
                    row_ok = row_ok && isfinite(v);
                    if (q == cls) mine = v;
                }
                s = (row_ok && mine > c.score_thresh) ? mine : -1.0f;
            }
            key = make_key(s, (unsigned int)t);
        }
        skeys[t] = key;
    }
    bitonic_desc(skeys, p);
    for (int t = threadIdx.x; t < n; t += kThreads) {
        const unsigned long long key = skeys[t];
        const unsigned int i = 0xffffffffu - (unsigned int)(key & 0xffffffffull);
        const float s = from_orderable((unsigned int)(key >> 32));
        out_order[t] = (long long)i;
        out_scores[t] = s;
        if (MODE == 0) {
            out_boxes[t] = a.boxes[i];
            out_group[t] = a.group[i];
            out_valid[t] = a.valid[i] & (a.valid2 ? a.valid2[i] : (unsigned char)1);
        } else {
            const int row = i / c.nc, cls = i - row * c.nc;
            float4 b = c.boxes[row];
            b = wd::clip_box(b, c.clip_w, c.clip_h);
            const bool real = s >= 0.f;                      // real candidates carry a softmax average > thresh >= 0
            out_boxes[t] = b;
            out_group[t] = real ? cls : -1;
            out_valid[t] = real ? 1 : 0;
        }
    }
}

// First `cap` kept & valid candidates (in sorted order) -> fixed-size outputs, unused rows zero, *count = how many.
// box_stride 4: (cap, 4) boxes; 5: ROI rows (0, x1, y1, x2, y2).
__global__ __launch_bounds__(256) void gather_kept_kernel(const unsigned char* __restrict__ keep, const unsigned char* __restrict__ valid,
                                                          const float4* __restrict__ boxes, const float* __restrict__ scores,
                                                          const long long* __restrict__ order, int n, int cap, int nc,
                                                          float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                          long long* __restrict__ out_class, int* __restrict__ count) {
    __shared__ int wave_sum[4];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    for (int k = tid; k < cap; k += 256) {
        reinterpret_cast<float4*>(out_boxes)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (out_scores) out_scores[k] = 0.f;
        if (out_class) out_class[k] = 0;
    }
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + tid;
        const bool f = i < n && keep[i] && valid[i];
        const unsigned long long m = __ballot(f);
        if (lane == 0) wave_sum[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_sum[w];
        const int k = off + __popcll(m & ((1ull << lane) - 1ull));
        if (f && k < cap) {
            reinterpret_cast<float4*>(out_boxes)[k] = boxes[i];
            if (out_scores) out_scores[k] = scores[i];
            if (out_class) out_class[k] = order[i] % nc;
        }
        __syncthreads();
        if (tid == 0) base_s += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        __syncthreads();
        if (base_s >= cap) break;
    }
    if (tid == 0) *count = base_s < cap ? base_s : cap;
}

// Detections of one frame -> the detection-JSON wire values, all slots of the frame in one launch.
__global__ __launch_bounds__(128) void wire_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores,
                                                   const long long* __restrict__ classes, const int* __restrict__ count, int n,
                                                   float in_w, float in_h, int hflip, double out_w, double out_h,
                                                   double* __restrict__ xywhs, int* __restrict__ category) {
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    float4 b = boxes[i];
    if (hflip) b = make_float4(in_w - b.z, b.y, in_w - b.x, b.w);           // HFlipTTA.post_process on the pixel boxes
    const float iw = (float)(1.0 / (double)in_w), ih = (float)(1.0 / (double)in_h);
    const float x1 = b.x * iw, y1 = b.y * ih, x2 = b.z * iw, y2 = b.w * ih;
    const float cx = (x1 + x2) * 0.5f, cy = (y1 + y2) * 0.5f;
    const float w = x2 - x1, h = y2 - y1;
    const double cx64 = (double)cx * out_w, cy64 = (double)cy * out_h;
    const double w64 = (double)w * out_w, h64 = (double)h * out_h;
    xywhs[0 * n + i] = trunc(cx64 - w64 * 0.5);
    xywhs[1 * n + i] = trunc(cy64 - h64 * 0.5);
    xywhs[2 * n + i] = trunc(w64);
    xywhs[3 * n + i] = trunc(h64);
    xywhs[4 * n + i] = rint((double)scores[i] * 1e5) / 1e5;
    category[i] = (!count || i < *count) ? (int)classes[i] + 1 : 0;
}

}  // namespace

extern "C" {

size_t wd_rpn_topk_workspace(const int* n_per_level, int n_levels, int k) {
    size_t a = 0;
    for (int l = 0; l < n_levels; ++l) {
        const size_t chunks = ((size_t)n_per_level[l] + kChunk - 1) / kChunk;
        a += chunks * (size_t)k;
    }
    return 2 * wt::align_up(a * 8) + 256;
}

// Per-level top-k of the objectness logits + decode of the selected anchors.  Candidate rows of level l start at
// sum_{m<l} min(k, n_m); out_* hold sum_l min(k, n_l) rows: boxes (clipped), scores (sorted descending inside a level),
// group = level or -1 for an empty box, valid = 0 for an empty box.
int wd_rpn_topk_decode_f32(const float* const* logits, const float* const* deltas, const float* const* anchors,
                           const int* n_per_level, int n_levels, int k, float img_h, float img_w, float scale_clamp,
                           float* out_boxes, float* out_scores, int32_t* out_group, uint8_t* out_valid, void* workspace,
                           size_t workspace_bytes, void* stream_) {
    WT_TRY(wt::ensure_device());
    hipStream_t stream = (hipStream_t)stream_;
    if (n_levels < 1 || n_levels > kMaxLevels || k < 1 || k > kChunk || !logits || !deltas || !anchors || !n_per_level) {
        wt::set_error("wd_rpn_topk_decode_f32: bad argument (levels=%d k=%d)", n_levels, k);
        return WT_ERR_INVALID;
    }
    if (!workspace || workspace_bytes < wd_rpn_topk_workspace(n_per_level, n_levels, k)) {
        wt::set_error("wd_rpn_topk_decode_f32: workspace too small");
        return WT_ERR_CAPACITY;
    }
    size_t half = 0;
    for (int l = 0; l < n_levels; ++l) half += (((size_t)n_per_level[l] + kChunk - 1) / kChunk) * (size_t)k;
    half = wt::align_up(half * 8);
    const uintptr_t mis = (uintptr_t)workspace & 255;
    char* base = (char*)workspace + (mis ? 256 - mis : 0);
    unsigned long long* buf[2] = {(unsigned long long*)base, (unsigned long long*)(base + half)};
    int cur_n[kMaxLevels];
    long long cur_off[kMaxLevels];
    for (int l = 0; l < n_levels; ++l) {
        if (n_per_level[l] < 1) { wt::set_error("wd_rpn_topk_decode_f32: empty level %d", l); return WT_ERR_INVALID; }
        cur_n[l] = n_per_level[l];
        cur_off[l] = 0;
    }
    bool first = true;
    int which = 0;
    for (int stage = 0; stage < 8; ++stage) {
        TopkStage st;
        memset(&st, 0, sizeof(st));
        st.n_levels = n_levels;
        st.k = k;
        bool final = true;
        int chunks_total = 0;
        long long out_off = 0, row = 0;
        for (int l = 0; l < n_levels; ++l) {
            const int chunks = (cur_n[l] + kChunk - 1) / kChunk;
            if (chunks > 1) final = false;
            st.chunk_begin[l] = chunks_total;
            chunks_total += chunks;
            st.in_n[l] = cur_n[l];
            st.in_off[l] = cur_off[l];
            st.logits[l] = logits[l];
            st.deltas[l] = deltas[l];
            st.anchors[l] = anchors[l];
        }
        st.chunk_begin[n_levels] = chunks_total;
        int next_n[kMaxLevels];
        for (int l = 0; l < n_levels; ++l) {
            const int chunks = (cur_n[l] + kChunk - 1) / kChunk;
            const int last = cur_n[l] - (chunks - 1) * kChunk;
            next_n[l] = (chunks - 1) * k + (last < k ? last : k);
            st.out_off[l] = final ? row : out_off;
            out_off += (long long)chunks * k;
            row += next_n[l];
        }
        hipLaunchKernelGGL(rpn_topk_stage_kernel, dim3((unsigned)chunks_total), dim3(kThreads), (size_t)kChunk * 8, stream, st,
                           first ? 1 : 0, final ? 1 : 0, (const unsigned long long*)buf[which], buf[which ^ 1], img_h, img_w,
                           scale_clamp, (float4*)out_boxes, out_scores, out_group, out_valid);
        WT_HIP(hipGetLastError());
        if (final) return WT_OK;
        for (int l = 0; l < n_levels; ++l) { cur_n[l] = next_n[l]; cur_off[l] = st.out_off[l]; }
        which ^= 1;
        first = false;
    }
    wt::set_error("wd_rpn_topk_decode_f32: selection did not converge");
    return WT_ERR_INVALID;
}

// Stable descending sort of RPN candidates (n <= 8192) and gather into sorted order.
int wd_sort_candidates_f32(const float* boxes, const float* scores, const int32_t* group, const uint8_t* valid,
                           const uint8_t* valid2, int n, float* out_boxes, float* out_scores, int32_t* out_group, uint8_t* out_valid, int64_t* out_order,
                           void* stream) {
    WT_TRY(wt::ensure_device());
    if (n < 1 || n > kChunk || !boxes || !scores || !group || !valid) {
        wt::set_error("wd_sort_candidates_f32: bad argument (n=%d, at most %d)", n, kChunk);
        return WT_ERR_INVALID;
    }
    CandRpn a{(const float4*)boxes, scores, group, valid, valid2};
    CandBox c{};
    hipLaunchKernelGGL(sort_candidates_kernel<0>, dim3(1), dim3(kThreads), (size_t)kChunk * 8, (hipStream_t)stream, a, c, n,
                       (float4*)out_boxes, out_scores, out_group, out_valid, (long long*)out_order);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

// Box-head candidates of fast_rcnn_inference_single_image: rows x classes, score = mean of the three stages' softmax.
int wd_box_candidates_f32(const float* boxes, const float* s0, const float* s1, const float* s2, const int32_t* n_valid, int rows,
                          int num_classes, float score_thresh, float img_h, float img_w, float* out_boxes, float* out_scores,
                          int32_t* out_group, uint8_t* out_valid, int64_t* out_order, void* stream) {
    WT_TRY(wt::ensure_device());
    const long long n = (long long)rows * num_classes;
    if (rows < 1 || num_classes < 1 || n > kChunk || !boxes || !s0 || !s1 || !s2 || !(score_thresh >= 0.f)) {
        wt::set_error("wd_box_candidates_f32: bad argument (rows=%d classes=%d, at most %d candidates, threshold >= 0)", rows,
                      num_classes, kChunk);
        return WT_ERR_INVALID;
    }
    CandRpn a{};
    CandBox c{(const float4*)boxes, s0, s1, s2, n_valid, num_classes, score_thresh, img_h, img_w};
    hipLaunchKernelGGL(sort_candidates_kernel<1>, dim3(1), dim3(kThreads), (size_t)kChunk * 8, (hipStream_t)stream, a, c, (int)n,
                       (float4*)out_boxes, out_scores, out_group, out_valid, (long long*)out_order);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

int wd_gather_kept_f32(const uint8_t* keep, const uint8_t* valid, const float* boxes, const float* scores, const int64_t* order,
                       int n, int cap, int num_classes, float* out_boxes, float* out_scores, int64_t* out_class, int32_t* count,
                       void* stream) {
    WT_TRY(wt::ensure_device());
    if (n < 0 || cap < 1 || !keep || !valid || !boxes || !out_boxes || !count || (out_class && (!order || num_classes < 1)) ||
        (out_scores && !scores)) {
        wt::set_error("wd_gather_kept_f32: bad argument");
        return WT_ERR_INVALID;
    }
    hipLaunchKernelGGL(gather_kept_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, keep, valid, (const float4*)boxes, scores,
                       (const long long*)order, n, cap, num_classes, out_boxes, out_scores, (long long*)out_class, count);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

int wd_detections_to_wire(const float* boxes, const float* scores, const int64_t* classes, const int32_t* count, int n, int in_w,
                          int in_h, int hflip, int out_w, int out_h, double* xywhs, int32_t* category, void* stream) {
    WT_TRY(wt::ensure_device());
    if (n < 0 || in_w < 1 || in_h < 1 || out_w < 1 || out_h < 1 || !xywhs || !category || (n > 0 && (!boxes || !scores || !classes))) {
        wt::set_error("wd_detections_to_wire: bad argument");
        return WT_ERR_INVALID;
    }
    if (n == 0) return WT_OK;
    hipLaunchKernelGGL(wire_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream, (const float4*)boxes,
                       scores, (const long long*)classes, count, n, (float)in_w, (float)in_h, hflip, (double)out_w, (double)out_h,
                       xywhs, category);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // extern "C"
