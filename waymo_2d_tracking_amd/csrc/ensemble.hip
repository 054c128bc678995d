// Batched ensemble merge on gfx950: linear soft-NMS, hard NMS and weighted box fusion, one 64-lane wavefront
// per (image, category) group.  Replaces, per group,
//   detnet/ensemble.py:50-64 ensemble() -> detnet/nn/tta.py:8-19 nms_detections / :22-66 merge_detections
//   -> detnet/utils/box_utils.py:307-395 nms(soft=True)          (paths relative to the reference repo)
// float64 like the reference's CPU tensors; compiled with -ffp-contract=off so every product/sum is rounded
// exactly as in the reference's elementwise torch ops (bit-exact results, see tests/test_gpu_ensemble.py).
//
// Data movement per group (n rows): one coalesced read of 40*n B (AoS rows -> SoA in LDS), one coalesced
// write of 40*n_out B.  Boxes, scores, rank order and alive flags live in LDS (57 B/box); groups too large for
// 64 KiB of LDS use the same code on a global-memory scratch (template<bool kLds>).
#include "common.h"

namespace {

constexpr int kWave = 64;
constexpr size_t kLdsBudget = 64 * 1024;

// "a is processed before b": descending score, NaN first (torch.sort ascending puts NaN last), ties by the
// higher original index (= stable ascending sort read from the back, box_utils.py:324,345).
__device__ __forceinline__ bool before(double a, int ia, double b, int ib) {
    const bool an = a != a, bn = b != b;
    if (an != bn) return an;
    if (!an && a != b) return a > b;
    return ia > ib;
}

struct GroupMem {
    double* s;      // scores (decayed in place)
    double* x1;
    double* y1;
    double* x2;
    double* y2;
    double* area;
    int* pos;       // pos[rank] = box index
    int* keep;      // kept box indices in output order
    unsigned char* alive;
};

__host__ __device__ inline size_t group_mem_bytes(size_t cap) {
    return cap * (6 * sizeof(double) + 2 * sizeof(int)) + ((cap + 15) / 16) * 16;
}

__device__ __forceinline__ GroupMem carve(char* base, size_t cap) {
    GroupMem m;
    double* d = reinterpret_cast<double*>(base);
    m.s = d; m.x1 = d + cap; m.y1 = d + 2 * cap; m.x2 = d + 3 * cap; m.y2 = d + 4 * cap; m.area = d + 5 * cap;
    int* ip = reinterpret_cast<int*>(d + 6 * cap);
    m.pos = ip; m.keep = ip + cap;
    m.alive = reinterpret_cast<unsigned char*>(ip + 2 * cap);
    return m;
}

// Rank boxes (descending score) and run the greedy loop of box_utils.py:344-385 on corner-form boxes held in
// `m`.  soft: linear decay clamp((cut-IoU)/(cut-thr),0,1) with no re-sorting; hard: torchvision.ops.nms rule
// (IoU > thr suppresses).  Only ranks < limit take part (top_k, box_utils.py:325-327).  Returns #kept.
__device__ int nms_core(const GroupMem& m, int n, int limit, bool soft, double thr, double cut, double conf) {
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += kWave) {
        const double si = m.s[i];
        int cnt = 0;
        for (int j = 0; j < n; ++j) cnt += before(m.s[j], j, si, i) ? 1 : 0;
        m.pos[cnt] = i;
        m.alive[i] = 1;
    }
    __syncthreads();
    const int L = (limit > 0 && limit < n) ? limit : n;
    int nk = 0;
    for (int r = 0; r < L; ++r) {
        const int i = m.pos[r];
        if (r > 0 && !m.alive[i]) continue;            // wave-uniform
        if (lane == 0) m.keep[nk] = i;
        ++nk;
        const double bx1 = m.x1[i], by1 = m.y1[i], bx2 = m.x2[i], by2 = m.y2[i], barea = m.area[i];
        for (int p = r + 1 + lane; p < L; p += kWave) {
            const int j = m.pos[p];
            if (!m.alive[j]) continue;
            double xx1 = m.x1[j]; if (xx1 < bx1) xx1 = bx1;
            double yy1 = m.y1[j]; if (yy1 < by1) yy1 = by1;
            double xx2 = m.x2[j]; if (xx2 > bx2) xx2 = bx2;
            double yy2 = m.y2[j]; if (yy2 > by2) yy2 = by2;
            double w = xx2 - xx1; if (w < 0.) w = 0.;
            double h = yy2 - yy1; if (h < 0.) h = 0.;
            const double inter = w * h;
            if (soft) {
                const double uni = (m.area[j] - inter) + barea;          // box_utils.py:366
                const double iou = inter / uni;
                double wgt = (cut - iou) / (cut - thr);                  // box_utils.py:373
                wgt = wgt < 0. ? 0. : (wgt > 1. ? 1. : wgt);
                const double sj = m.s[j] * wgt;
                m.s[j] = sj;
                m.alive[j] = (sj >= conf) ? 1 : 0;                       // box_utils.py:379-381
            } else {
                const double iou = inter / ((barea + m.area[j]) - inter);
                if (iou > thr) m.alive[j] = 0;
            }
        }
        __syncthreads();
    }
    return nk;
}

// ensemble.py:50-64 with merge_func = nms_detections (methods 1, 2)
template <bool kLds>
__device__ void ensemble_nms_group(const double* __restrict__ in, int n, char* mem, size_t cap, bool soft, bool centre,
                                   double thr, double cut, double* __restrict__ out, int64_t* out_count) {
    const int lane = threadIdx.x;
    GroupMem m = carve(mem, cap);
    double* col[5] = {m.s, m.x1, m.y1, m.x2, m.y2};       // raw [score, x, y, w, h] first
    for (int j = lane; j < 5 * n; j += kWave) {
        const int r = j / 5, c = j - 5 * r;
        col[c][r] = in[j];
    }
    __syncthreads();
    for (int i = lane; i < n; i += kWave) {
        const double x = m.x1[i], y = m.y1[i], w = m.x2[i], h = m.y2[i];
        const double cx = centre ? x : x + w / 2, cy = centre ? y : y + h / 2;   // ensemble.py:19-22
        const double hx = w * 0.5, hy = h * 0.5;                           // box_utils.py:32-35
        const double a = cx - hx, b = cy - hy, c2 = cx + hx, d = cy + hy;
        m.x1[i] = a; m.y1[i] = b; m.x2[i] = c2; m.y2[i] = d;
        m.area[i] = (c2 - a) * (d - b);                                    // box_utils.py:342
    }
    __syncthreads();
    const int nk = nms_core(m, n, 0, soft, thr, cut, 0.0);
    for (int j = lane; j < 5 * nk; j += kWave) {
        const int k = j / 5, c = j - 5 * k;
        const int i = m.keep[k];
        double v;
        if (c == 0) {
            v = m.s[i];
        } else if (c == 1 || c == 3) {
            const double wd = m.x2[i] - m.x1[i];                            // box_utils.py:65-69
            const double cxo = (m.x1[i] + m.x2[i]) * 0.5;
            v = (c == 3) ? wd : (centre ? cxo : cxo - wd / 2);              // ensemble.py:25-28
        } else {
            const double hd = m.y2[i] - m.y1[i];
            const double cyo = (m.y1[i] + m.y2[i]) * 0.5;
            v = (c == 4) ? hd : (centre ? cyo : cyo - hd / 2);
        }
        out[j] = v;
    }
    if (lane == 0) *out_count = nk;
}

// jaccard_bbox on centre-form boxes (box_utils.py:126-140, :72-92, :114-123); NaN propagates like torch.min/max
__device__ __forceinline__ double tmin(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
__device__ __forceinline__ double tmax(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }

__device__ __forceinline__ double jaccard_center(const double a[4], const double b[4]) {
    const double ax1 = a[0] - a[2] * 0.5, ay1 = a[1] - a[3] * 0.5, ax2 = a[0] + a[2] * 0.5, ay2 = a[1] + a[3] * 0.5;
    const double bx1 = b[0] - b[2] * 0.5, by1 = b[1] - b[3] * 0.5, bx2 = b[0] + b[2] * 0.5, by2 = b[1] + b[3] * 0.5;
    double iw = tmin(ax2, bx2) - tmax(ax1, bx1); if (iw < 0.) iw = 0.;
    double ih = tmin(ay2, by2) - tmax(ay1, by1); if (ih < 0.) ih = 0.;
    const double inter = iw * ih;
    const double uni = (a[2] * a[3] + b[2] * b[3]) - inter;
    return inter / uni;
}

// merge_detections (nn/tta.py:22-66): greedy fold of the K inputs of one group.
//   res  : running fused rows, score-weighted ([s, s*cx, s*cy, s*w, s*h]), at most n rows
//   mbox : res[:,1:]/res[:,:1] refreshed once per input (tta.py:44)
//   am/mt: per `other` row: argmax result index and matched flag (tta.py:46-49)
//   win  : per result: the LAST matched other (numpy fancy `+=` keeps one contribution)
struct FuseMem {
    double* res;   // 5 x cap (SoA)
    double* mbox;  // 4 x cap
    double* oth;   // 5 x cap  scaled rows of the current input
    int* am;
    int* win;
    unsigned char* mt;   // 0 unmatched, 1 matched, 2 neither (NaN IoU)
};

__host__ __device__ inline size_t fuse_mem_bytes(size_t cap) {
    return cap * (14 * sizeof(double) + 2 * sizeof(int)) + ((cap + 15) / 16) * 16;
}

__device__ void ensemble_fusion_group(const double* __restrict__ in, int n, const int32_t* __restrict__ sizes,
                                      int k_inputs, char* mem, size_t cap, bool centre, double thr,
                                      double* __restrict__ out, int64_t* out_count) {
    const int lane = threadIdx.x;
    FuseMem f;
    double* d = reinterpret_cast<double*>(mem);
    f.res = d; f.mbox = d + 5 * cap; f.oth = d + 9 * cap;
    int* ip = reinterpret_cast<int*>(d + 14 * cap);
    f.am = ip; f.win = ip + cap;
    f.mt = reinterpret_cast<unsigned char*>(ip + 2 * cap);
    const double K = (double)k_inputs;
    int m = sizes[0];
    // tta.py:34-36 (after ensemble.py:19-22 lxly2cxcy)
    for (int i = lane; i < m; i += kWave) {
        const double* r = in + 5 * (size_t)i;
        const double s = r[0] / K;
        const double w = r[3], h = r[4];
        f.res[i] = s;
        f.res[cap + i] = (centre ? r[1] : r[1] + w / 2) * s;
        f.res[2 * cap + i] = (centre ? r[2] : r[2] + h / 2) * s;
        f.res[3 * cap + i] = w * s;
        f.res[4 * cap + i] = h * s;
    }
    __syncthreads();
    int off = sizes[0];
    for (int k = 1; k < k_inputs; ++k) {
        const int no = sizes[k];
        if (no > 0) {
            for (int i = lane; i < no; i += kWave) {                       // tta.py:40-41
                const double* r = in + 5 * (size_t)(off + i);
                const double s = r[0] / K;
                const double w = r[3], h = r[4];
                f.oth[i] = s;
                f.oth[cap + i] = (centre ? r[1] : r[1] + w / 2) * s;
                f.oth[2 * cap + i] = (centre ? r[2] : r[2] + h / 2) * s;
                f.oth[3 * cap + i] = w * s;
                f.oth[4 * cap + i] = h * s;
            }
            if (m > 0) {
                for (int r = lane; r < m; r += kWave) {                     // tta.py:44
                    const double s = f.res[r];
                    for (int c = 0; c < 4; ++c) f.mbox[c * cap + r] = f.res[(c + 1) * cap + r] / s;
                    f.win[r] = -1;
                }
                __syncthreads();
                for (int i = lane; i < no; i += kWave) {                    // tta.py:45-49
                    const double s = f.oth[i];
                    double ob[4];
                    for (int c = 0; c < 4; ++c) ob[c] = f.oth[(c + 1) * cap + i] / s;
                    double best = 0.;
                    int bi = 0;
                    for (int r = 0; r < m; ++r) {
                        double mb[4];
                        for (int c = 0; c < 4; ++c) mb[c] = f.mbox[c * cap + r];
                        const double v = jaccard_center(mb, ob);
                        if (r == 0 || (best == best && (v > best || v != v))) { best = v; bi = r; }
                    }
                    f.am[i] = bi;
                    f.mt[i] = (best >= thr) ? 1 : ((best < thr) ? 0 : 2);
                    if (best >= thr) atomicMax(&f.win[bi], i);              // last matched other wins (tta.py:55)
                }
                __syncthreads();
                for (int r = lane; r < m; r += kWave) {
                    const int o = f.win[r];
                    if (o >= 0)
                        for (int c = 0; c < 5; ++c) f.res[c * cap + r] = f.res[c * cap + r] + f.oth[c * cap + o];
                }
                // tta.py:57-59 append the unmatched rows in order
                const int m0 = m;
                for (int base = 0; base < no; base += kWave) {
                    const int i = base + lane;
                    const bool un = (i < no) && f.mt[i] == 0;
                    const unsigned long long mask = __ballot(un);
                    if (un) {
                        const int dst = m + __popcll(mask & ((1ull << lane) - 1ull));
                        for (int c = 0; c < 5; ++c) f.res[c * cap + dst] = f.oth[c * cap + i];
                    }
                    m += __popcll(mask);
                }
                (void)m0;
                __syncthreads();
            } else {                                                         // tta.py:60-62
                __syncthreads();
                for (int i = lane; i < no; i += kWave)
                    for (int c = 0; c < 5; ++c) f.res[c * cap + i] = f.oth[c * cap + i];
                m = no;
                __syncthreads();
            }
        }
        off += no;
    }
    // tta.py:65 then ensemble.py:25-28
    for (int j = lane; j < 5 * m; j += kWave) {
        const int r = j / 5, c = j - 5 * r;
        const double s = f.res[r];
        double v;
        if (c == 0) v = s;
        else if (c >= 3) v = f.res[c * cap + r] / s;
        else v = centre ? f.res[c * cap + r] / s : f.res[c * cap + r] / s - (f.res[(c + 2) * cap + r] / s) / 2;
        out[j] = v;
    }
    if (lane == 0) *out_count = m;
}


// ---- dependency-free soft-NMS (ensemble path: conf_thresh == 0, 0 <= thr < cut) ---------------------------------
// With conf_thresh = 0 a box only leaves the list when its score turns NaN / negative, so for finite non-negative
// scores every box survives and the decay of box j is just the product, in rank order, of the weights of the boxes
// ranked above it - the weights depend on geometry only.  That removes the per-rank barrier of the serial loop:
//   * 256 threads per group rank the boxes by counting and rewrite them in rank order in LDS (killers, read-only);
//   * victims are walked in x-sorted order, 64 per wavefront, so that a killer overlaps either several lanes of a
//     wave or none: the cheap interval test + one ballot skips the two float64 divisions for most (killer, chunk)
//     pairs (a non-overlapping pair has weight clamp(cut/(cut-thr)) == 1 exactly, so skipping is bit-exact);
//   * no synchronisation inside the main loop.
// Groups with NaN / negative scores, thr < 0 or conf_thresh > 0 are flagged and handled by the serial kernel.
constexpr int kFastThreads = 256;
constexpr int kFastMaxPerThread = 8;             // n <= 2048

__host__ __device__ inline size_t fast_mem_bytes(size_t cap) { return cap * (6 * sizeof(double) + sizeof(int)) + 32; }

__global__ __launch_bounds__(kFastThreads) void softnms_fast_kernel(
    const double* __restrict__ dets5, const int64_t* __restrict__ group_offsets, int64_t n_groups, int centre,
    double thr, double cut, double* __restrict__ out5, int64_t* __restrict__ out_counts, int* __restrict__ fallback,
    size_t cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ks = reinterpret_cast<double*>(smem);        // by rank: score, x1, y1, x2, y2, area
    double* kx1 = ks + cap; double* ky1 = ks + 2 * cap; double* kx2 = ks + 3 * cap; double* ky2 = ks + 4 * cap;
    double* kar = ks + 5 * cap;
    int* vord = reinterpret_cast<int*>(ks + 6 * cap);    // victims in x order -> rank index
    int& bad = vord[cap];                                // (kept inside the dynamic block: its base stays 16-B aligned)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t o0 = group_offsets[g];
        const int n = (int)(group_offsets[g + 1] - o0);
        if (tid == 0) bad = 0;
        __syncthreads();
        if (n == 0) { if (tid == 0) { out_counts[g] = 0; fallback[g] = 0; } continue; }
        if (n > kFastThreads * kFastMaxPerThread || (size_t)n > cap) { if (tid == 0) fallback[g] = 1; continue; }
        const double* in = dets5 + 5 * o0;
        // 1. coalesced load, AoS -> SoA (raw order) into the LDS arrays
        double* col[5] = {ks, kx1, ky1, kx2, ky2};
        for (int j = tid; j < 5 * n; j += kFastThreads) { const int r = j / 5, c = j - 5 * r; col[c][r] = in[j]; }
        __syncthreads();
        // 2. corners per owned box (registers), validity check, rank by counting
        double bs[kFastMaxPerThread], b1[kFastMaxPerThread], b2[kFastMaxPerThread], b3[kFastMaxPerThread], b4[kFastMaxPerThread];
        int rk[kFastMaxPerThread];
        bool mybad = false;
#pragma unroll
        for (int k = 0; k < kFastMaxPerThread; ++k) {
            const int i = tid + k * kFastThreads;
            rk[k] = -1;
            if (i < n) {
                const double sc = ks[i], x = kx1[i], y = ky1[i], w = kx2[i], h = ky2[i];
                const double cx = centre ? x : x + w / 2, cy = centre ? y : y + h / 2;
                const double hx = w * 0.5, hy = h * 0.5;
                bs[k] = sc; b1[k] = cx - hx; b2[k] = cy - hy; b3[k] = cx + hx; b4[k] = cy + hy;
                mybad = mybad || !(sc >= 0.) || !(sc - sc == 0.);
                int cnt = 0;
                for (int j = 0; j < n; ++j) cnt += before(ks[j], j, sc, i) ? 1 : 0;
                rk[k] = cnt;
            }
        }
        if (mybad) bad = 1;
        __syncthreads();
        if (bad) { if (tid == 0) fallback[g] = 1; __syncthreads(); continue; }
        // 3. rewrite in rank order (all reads of the raw arrays are done)
#pragma unroll
        for (int k = 0; k < kFastMaxPerThread; ++k)
            if (rk[k] >= 0) {
                const int r = rk[k];
                ks[r] = bs[k]; kx1[r] = b1[k]; ky1[r] = b2[k]; kx2[r] = b3[k]; ky2[r] = b4[k];
                kar[r] = (b3[k] - b1[k]) * (b4[k] - b2[k]);
            }
        __syncthreads();
        // 4. x order of the victims (count-rank on (x1, rank))
        for (int r = tid; r < n; r += kFastThreads) {
            const double xr = kx1[r];
            int cnt = 0;
            for (int j = 0; j < n; ++j) { const double xj = kx1[j]; cnt += (xj < xr || (xj == xr && j < r)) ? 1 : 0; }
            vord[cnt] = r;
        }
        __syncthreads();
        // 5. decay: each wave owns x-sorted chunks of 64 victims; killers walked in rank order
        for (int base = wave * 64; base < n; base += (kFastThreads / 64) * 64) {
            const int v = base + lane;
            const bool act = v < n;
            const int myr = act ? vord[v] : 0;
            const double mx1 = kx1[myr], my1 = ky1[myr], mx2 = kx2[myr], my2 = ky2[myr], mar = kar[myr];
            double ms = ks[myr];
            // x extent of the chunk: killers that cannot reach it are skipped with one scalar compare
            double cmin = act ? mx1 : 1e300, cmax = act ? mx2 : -1e300;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double a = __shfl_xor(cmin, o, 64), b = __shfl_xor(cmax, o, 64);
                cmin = a < cmin ? a : cmin; cmax = b > cmax ? b : cmax;
            }
            int rmax = act ? myr : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int a = __shfl_xor(rmax, o, 64); rmax = a > rmax ? a : rmax; }
            for (int r = 0; r < rmax; ++r) {
                const double bx1 = kx1[r], bx2 = kx2[r];
                if (!(bx2 > cmin && bx1 < cmax)) continue;                 // wave-uniform: no overlap in x
                const double by1 = ky1[r], by2 = ky2[r];
                double xx1 = mx1; if (xx1 < bx1) xx1 = bx1;
                double yy1 = my1; if (yy1 < by1) yy1 = by1;
                double xx2 = mx2; if (xx2 > bx2) xx2 = bx2;
                double yy2 = my2; if (yy2 > by2) yy2 = by2;
                double w = xx2 - xx1; if (w < 0.) w = 0.;
                double h = yy2 - yy1; if (h < 0.) h = 0.;
                const double inter = w * h;
                const bool hit = act && r < myr && inter > 0.;
                if (__ballot(hit) == 0ull) continue;
                if (hit) {
                    const double uni = (mar - inter) + kar[r];
                    const double iou = inter / uni;
                    double wgt = (cut - iou) / (cut - thr);
                    wgt = wgt < 0. ? 0. : (wgt > 1. ? 1. : wgt);
                    ms = ms * wgt;
                }
            }
            if (act) ks[myr] = ms;                                           // each rank slot is owned by one lane
        }
        __syncthreads();
        // 6. output in rank order (every box survives)
        double* out = out5 + 5 * o0;
        for (int j = tid; j < 5 * n; j += kFastThreads) {
            const int k = j / 5, c = j - 5 * k;
            double v;
            if (c == 0) v = ks[k];
            else if (c == 1 || c == 3) {
                const double wd = kx2[k] - kx1[k], cxo = (kx1[k] + kx2[k]) * 0.5;
                v = (c == 3) ? wd : (centre ? cxo : cxo - wd / 2);
            } else {
                const double hd = ky2[k] - ky1[k], cyo = (ky1[k] + ky2[k]) * 0.5;
                v = (c == 4) ? hd : (centre ? cyo : cyo - hd / 2);
            }
            out[j] = v;
        }
        if (tid == 0) { out_counts[g] = n; fallback[g] = 0; }
        __syncthreads();
    }
}

template <bool kLds>
__global__ __launch_bounds__(kWave) void ensemble_groups_kernel(
    const double* __restrict__ dets5, const int64_t* __restrict__ group_offsets,
    const int32_t* __restrict__ input_sizes, int64_t n_groups, int k_inputs, int method, double thr, double cut,
    double* __restrict__ out5, int64_t* __restrict__ out_counts, char* scratch, size_t lds_cap, size_t bytes_per_row,
    const int* __restrict__ only_flagged) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        if (only_flagged && !only_flagged[g]) continue;        // done by softnms_fast_kernel
        const int64_t o0 = group_offsets[g];
        const int n = (int)(group_offsets[g + 1] - o0);
        if (n == 0) {
            if (threadIdx.x == 0) out_counts[g] = 0;
            continue;
        }
        // LDS: arrays sized lds_cap.  Global scratch: this group's slice holds exactly n rows of every array.
        char* mem = kLds ? smem : scratch + (size_t)o0 * bytes_per_row + (size_t)g * 64;
        const size_t cap = kLds ? lds_cap : (size_t)n;
        const bool centre = (method & 16) != 0;
        const int mth = method & 15;
        if (mth == 0)
            ensemble_fusion_group(dets5 + 5 * o0, n, input_sizes + g * k_inputs, k_inputs, mem, cap, centre, thr,
                                  out5 + 5 * o0, out_counts + g);
        else
            ensemble_nms_group<kLds>(dets5 + 5 * o0, n, mem, cap, mth == 2, centre, thr, cut, out5 + 5 * o0,
                                     out_counts + g);
        __syncthreads();
    }
}

// nms(boxes, scores, ...) raw API (box_utils.py:307): one group, corner boxes, returns keep + scores
template <bool kLds>
__global__ __launch_bounds__(kWave) void nms_raw_kernel(const double* __restrict__ boxes4,
                                                        const double* __restrict__ scores, int n, bool soft,
                                                        double thr, double cut, double conf, int top_k,
                                                        int64_t* __restrict__ keep, double* __restrict__ out_scores,
                                                        int* __restrict__ n_keep, char* scratch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    GroupMem m = carve(kLds ? smem : scratch, (size_t)n);
    double* col[4] = {m.x1, m.y1, m.x2, m.y2};
    for (int j = lane; j < 4 * n; j += kWave) {
        const int r = j >> 2, c = j & 3;
        col[c][r] = boxes4[j];
    }
    for (int i = lane; i < n; i += kWave) m.s[i] = scores[i];
    __syncthreads();
    for (int i = lane; i < n; i += kWave) m.area[i] = (m.x2[i] - m.x1[i]) * (m.y2[i] - m.y1[i]);
    __syncthreads();
    const int nk = nms_core(m, n, top_k, soft, thr, cut, conf);
    for (int k = lane; k < nk; k += kWave) {
        const int i = m.keep[k];
        keep[k] = i;
        out_scores[k] = soft ? m.s[i] : scores[i];
    }
    if (lane == 0) *n_keep = nk;
}

size_t row_bytes(int method) { return method == 0 ? fuse_mem_bytes(1) + 16 : group_mem_bytes(1) + 16; }

int launch_groups(const double* dets5, const int64_t* group_offsets, const int32_t* input_sizes, int64_t n_rows,
                  int64_t n_groups, int64_t max_group_rows, int k_inputs, int method, double thr, double cut,
                  double* out5, int64_t* out_counts, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (n_groups <= 0) return WT_OK;
    const int mth = method & 15;
    if (method < 0 || (method & ~31) || mth > 2) { wt::set_error("method must be 0 (fusion), 1 (nms) or 2 (soft_nms), optionally | 16"); return WT_ERR_INVALID; }
    if (mth == 0 && (!input_sizes || k_inputs < 1)) { wt::set_error("weighted fusion needs input_sizes"); return WT_ERR_INVALID; }
    const size_t cap = (size_t)(max_group_rows > 0 ? max_group_rows : 1);
    const size_t lds = mth == 0 ? fuse_mem_bytes(cap) : group_mem_bytes(cap);
    const int64_t grid = n_groups < (1 << 20) ? n_groups : (1 << 20);
    // soft-NMS fast path (dependency-free form); it flags the groups it cannot take for the serial kernel below
    const int* flags = nullptr;
    if (mth == 2 && thr >= 0. && cut > thr && cap <= (size_t)kFastThreads * kFastMaxPerThread &&
        fast_mem_bytes(cap) <= 150 * 1024) {
        static thread_local int* d_flags = nullptr;
        static thread_local int64_t d_flags_cap = 0;
        if (d_flags_cap < n_groups) {
            if (d_flags) (void)hipFree(d_flags);
            WT_HIP(hipMalloc(&d_flags, sizeof(int) * (size_t)n_groups * 2));
            d_flags_cap = n_groups * 2;
        }
        const size_t fl = fast_mem_bytes(cap);
        if (fl > 48 * 1024)
            WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(softnms_fast_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl));
        hipLaunchKernelGGL(softnms_fast_kernel, dim3((unsigned)grid), dim3(kFastThreads), fl, stream, dets5, group_offsets,
                           n_groups, (method & 16) ? 1 : 0, thr, cut, out5, out_counts, d_flags, cap);
        flags = d_flags;
    }
    if (lds <= kLdsBudget) {
        hipLaunchKernelGGL(ensemble_groups_kernel<true>, dim3((unsigned)grid), dim3(kWave), lds, stream, dets5,
                           group_offsets, input_sizes, n_groups, k_inputs, method, thr, cut, out5, out_counts,
                           (char*)nullptr, cap, (size_t)0, flags);
    } else {
        const size_t need = wt_ensemble_groups_workspace(n_rows, n_groups, max_group_rows);
        if (!workspace || workspace_bytes < need) {
            wt::set_error("ensemble workspace too small: need %zu bytes, have %zu", need, workspace_bytes);
            return WT_ERR_CAPACITY;
        }
        // per-row bytes are padded so that each group's slice stays 16-byte aligned for any n
        hipLaunchKernelGGL(ensemble_groups_kernel<false>, dim3((unsigned)grid), dim3(kWave), 0, stream, dets5,
                           group_offsets, input_sizes, n_groups, k_inputs, method, thr, cut, out5, out_counts,
                           (char*)workspace, (size_t)0, (size_t)128, flags);
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // namespace

extern "C" {

size_t wt_ensemble_groups_workspace(int64_t n_rows, int64_t n_groups, int64_t max_group_rows) {
    const size_t cap = (size_t)(max_group_rows > 0 ? max_group_rows : 1);
    if (fuse_mem_bytes(cap) <= kLdsBudget) return 0;          // every method fits in LDS
    return (size_t)n_rows * 128 + (size_t)n_groups * 64 + 256;
}

int wt_ensemble_groups_dev(const double* dets5, const int64_t* group_offsets, const int32_t* input_sizes,
                           int64_t n_rows, int64_t n_groups, int64_t max_group_rows, int k_inputs, int method,
                           double iou_thresh, double soft_nms_cut, double* out5, int64_t* out_counts,
                           void* workspace, size_t workspace_bytes, void* stream) {
    WT_TRY(wt::ensure_device());
    return launch_groups(dets5, group_offsets, input_sizes, n_rows, n_groups, max_group_rows, k_inputs, method,
                         iou_thresh, soft_nms_cut, out5, out_counts, workspace, workspace_bytes, (hipStream_t)stream);
}

int wt_ensemble_groups_host(const double* dets5, const int64_t* group_offsets, const int32_t* input_sizes,
                            int64_t n_groups, int k_inputs, int method, double iou_thresh, double soft_nms_cut,
                            double* out5, int64_t* out_counts) {
    WT_TRY(wt::ensure_device());
    if (n_groups <= 0) return WT_OK;
    const int64_t n_rows = group_offsets[n_groups];
    int64_t max_rows = 0;
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t c = group_offsets[g + 1] - group_offsets[g];
        if (c < 0) { wt::set_error("group_offsets must be non-decreasing"); return WT_ERR_INVALID; }
        if (c > max_rows) max_rows = c;
    }
    wt::DevBuf d_in, d_off, d_sz, d_out, d_cnt, d_ws;
    WT_TRY(d_in.alloc(sizeof(double) * 5 * (size_t)n_rows));
    WT_TRY(d_off.alloc(sizeof(int64_t) * (size_t)(n_groups + 1)));
    WT_TRY(d_out.alloc(sizeof(double) * 5 * (size_t)n_rows));
    WT_TRY(d_cnt.alloc(sizeof(int64_t) * (size_t)n_groups));
    WT_HIP(hipMemcpy(d_in.p, dets5, sizeof(double) * 5 * (size_t)n_rows, hipMemcpyHostToDevice));
    WT_HIP(hipMemcpy(d_off.p, group_offsets, sizeof(int64_t) * (size_t)(n_groups + 1), hipMemcpyHostToDevice));
    if (input_sizes) {
        WT_TRY(d_sz.alloc(sizeof(int32_t) * (size_t)n_groups * (size_t)k_inputs));
        WT_HIP(hipMemcpy(d_sz.p, input_sizes, sizeof(int32_t) * (size_t)n_groups * (size_t)k_inputs, hipMemcpyHostToDevice));
    }
    const size_t ws = wt_ensemble_groups_workspace(n_rows, n_groups, max_rows);
    if (ws) WT_TRY(d_ws.alloc(ws));
    WT_TRY(launch_groups(d_in.as<double>(), d_off.as<int64_t>(), d_sz.as<int32_t>(), n_rows, n_groups, max_rows,
                         k_inputs, method, iou_thresh, soft_nms_cut, d_out.as<double>(), d_cnt.as<int64_t>(),
                         d_ws.p, ws, nullptr));
    WT_HIP(hipDeviceSynchronize());
    WT_HIP(hipMemcpy(out5, d_out.p, sizeof(double) * 5 * (size_t)n_rows, hipMemcpyDeviceToHost));
    WT_HIP(hipMemcpy(out_counts, d_cnt.p, sizeof(int64_t) * (size_t)n_groups, hipMemcpyDeviceToHost));
    return WT_OK;
}

static int nms_raw_host(const double* boxes4, const double* scores, int n, bool soft, double overlap, double cut,
                        double conf, int top_k, int64_t* keep, double* out_scores, int* n_keep) {
    WT_TRY(wt::ensure_device());
    *n_keep = 0;
    if (n <= 0) return WT_OK;
    wt::DevBuf d_b, d_s, d_k, d_o, d_n, d_ws;
    WT_TRY(d_b.alloc(sizeof(double) * 4 * (size_t)n));
    WT_TRY(d_s.alloc(sizeof(double) * (size_t)n));
    WT_TRY(d_k.alloc(sizeof(int64_t) * (size_t)n));
    WT_TRY(d_o.alloc(sizeof(double) * (size_t)n));
    WT_TRY(d_n.alloc(sizeof(int)));
    WT_HIP(hipMemcpy(d_b.p, boxes4, sizeof(double) * 4 * (size_t)n, hipMemcpyHostToDevice));
    WT_HIP(hipMemcpy(d_s.p, scores, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    const size_t lds = group_mem_bytes((size_t)n);
    if (lds <= kLdsBudget) {
        hipLaunchKernelGGL(nms_raw_kernel<true>, dim3(1), dim3(kWave), lds, nullptr, d_b.as<double>(), d_s.as<double>(),
                           n, soft, overlap, cut, conf, top_k, d_k.as<int64_t>(), d_o.as<double>(), d_n.as<int>(),
                           (char*)nullptr);
    } else {
        WT_TRY(d_ws.alloc(lds));
        hipLaunchKernelGGL(nms_raw_kernel<false>, dim3(1), dim3(kWave), 0, nullptr, d_b.as<double>(), d_s.as<double>(),
                           n, soft, overlap, cut, conf, top_k, d_k.as<int64_t>(), d_o.as<double>(), d_n.as<int>(),
                           d_ws.as<char>());
    }
    WT_HIP(hipGetLastError());
    WT_HIP(hipDeviceSynchronize());
    WT_HIP(hipMemcpy(n_keep, d_n.p, sizeof(int), hipMemcpyDeviceToHost));
    WT_HIP(hipMemcpy(keep, d_k.p, sizeof(int64_t) * (size_t)*n_keep, hipMemcpyDeviceToHost));
    WT_HIP(hipMemcpy(out_scores, d_o.p, sizeof(double) * (size_t)*n_keep, hipMemcpyDeviceToHost));
    return WT_OK;
}

int wt_softnms_f64_host(const double* boxes4, const double* scores, int n, double overlap, double cut,
                        double conf_thresh, int top_k, int64_t* keep, double* out_scores, int* n_keep) {
    return nms_raw_host(boxes4, scores, n, true, overlap, cut, conf_thresh, top_k, keep, out_scores, n_keep);
}

int wt_hardnms_f64_host(const double* boxes4, const double* scores, int n, double overlap, int top_k,
                        int64_t* keep, double* out_scores, int* n_keep) {
    return nms_raw_host(boxes4, scores, n, false, overlap, 1.0, 0.0, top_k, keep, out_scores, n_keep);
}

}  // extern "C"
