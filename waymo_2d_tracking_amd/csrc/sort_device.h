// Device-side SORT building blocks for gfx950: one 64-lane wavefront owns one tracker (one class of one
// (segment, camera) stream).  Mirrors, bit for bit, the arithmetic order of the reference path
//   tracking/sort/sort.py (KalmanBoxTracker :78-190, iou :33-47, associate :193-230, Sort.update :244-296)
// with filterpy's KalmanFilter and scikit-learn 0.22.2's Munkres restated (oracle/sort_oracle.c is the CPU twin).
// Compile with -ffp-contract=off: products and sums are rounded one by one, k ascending, like the oracle.
//
// Layout: Kalman state is SoA in global memory (element e of slot s at e*cap + s) so a wave's 64 tracks read
// 64 consecutive doubles per element; the float32 cost matrix and the Munkres cover/star arrays live in LDS
// (global scratch when N*T exceeds the LDS budget); lists are compacted with ballot + popcount prefix sums.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include <type_traits>
namespace wtdev {

constexpr int kWave = 64;
constexpr int kErrCapacity = 4;   // WT_ERR_CAPACITY
constexpr int kErrNumeric = 5;    // WT_ERR_NUMERIC

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// The tracker is ONE wave: what its phases need between each other is "my earlier LDS / global writes are visible to my other lanes", not a
// workgroup barrier.  (With helper waves in the workgroup - HelpJob below - a real barrier here would wake them.)
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }
__device__ __forceinline__ bool finite_d(double v) { return (v == v) && (v - v == 0.0); }

// Minimum over the 64 lanes, the same value in every lane.  Round 4: DPP lane permutations (VALU speed) instead of six ds_bpermute round trips
// (~100 cycles each; Munkres step 6 calls this ~26 times per frame).  The inputs are never NaN (the callers' per-lane minima start from +inf
// and skip NaN entries with `v < mn`), so the order of the comparisons does not matter.
__device__ __forceinline__ float wave_min_f(float v) {
#ifdef WT_WAVE_MIN_BPERMUTE
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float u = __shfl_xor(v, o, kWave);
        v = (u < v) ? u : v;
    }
    return v;
#else
    auto step = [](float x, auto CTRL, auto ROWMASK) {
        const int moved = __builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(CTRL)::value, decltype(ROWMASK)::value, 0xF, false);
        const float u = __int_as_float(moved);
        return (u < x) ? u : x;
    };
    using I = std::integral_constant<int, 0>;
    (void)sizeof(I);
    v = step(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});      // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});      // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});     // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});     // row_mirror: every lane of a row holds the row's minimum
    v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});     // row_bcast:15 into rows 1 and 3
    v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});     // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
#endif
}

// ---- sort.py:33-47 iou(float32 detection, float64 track) under NumPy-1.x scalar promotion -----------------
__device__ __forceinline__ double iou_det_trk(const float d[4], const double t[4]) {
    const double d0 = (double)d[0], d1 = (double)d[1], d2 = (double)d[2], d3 = (double)d[3];
    const double xx1 = (d0 > t[0]) ? d0 : t[0];
    const double yy1 = (d1 > t[1]) ? d1 : t[1];
    const double xx2 = (d2 < t[2]) ? d2 : t[2];
    const double yy2 = (d3 < t[3]) ? d3 : t[3];
    double w = xx2 - xx1; if (!(w > 0.)) w = (w != w) ? w : 0.;
    double h = yy2 - yy1; if (!(h > 0.)) h = (h != h) ? h : 0.;
    const double wh = w * h;
    const float dw = d[2] - d[0];
    const float dh = d[3] - d[1];
    const float darea = dw * dh;                                   // float32 product (sort.py:44)
    const double tarea = (t[2] - t[0]) * (t[3] - t[1]);
    return wh / (((double)darea + tarea) - wh);
}

// ---- sort.py:50-62 / :65-75 -------------------------------------------------------------------------------
__device__ __forceinline__ void bbox_to_z(const float b[4], double z[4]) {
    const float w = b[2] - b[0];
    const float h = b[3] - b[1];
    z[0] = (double)b[0] + (double)w / 2.;
    z[1] = (double)b[1] + (double)h / 2.;
    const float s = w * h;                                          // float32 area
    z[2] = (double)s;
    z[3] = (double)w / (double)h;
}

__device__ __forceinline__ void x_to_bbox(double x0, double x1, double x2, double x3, double b[4]) {
    const double w = sqrt(x2 * x3);
    const double h = x2 / w;
    b[0] = x0 - w / 2.;
    b[1] = x1 - h / 2.;
    b[2] = x0 + w / 2.;
    b[3] = x1 + h / 2.;
}

// Q, R, P0 of sort.py:111-115,127,133-134
__device__ __forceinline__ double kq(int i) { return i < 2 ? 2. : (i == 2 ? 1. : (i == 3 ? 25. : (i == 6 ? 5. : 4.))); }
__device__ __forceinline__ double kr(int i) { return i < 2 ? 1. : 10.; }
__device__ __forceinline__ double kp0(int i) { return i < 4 ? 10. : 10000.; }

struct TrackerMem {
    double* kx;      // [7][cap]   state, by slot
    double* kP;      // [49][cap]  covariance, by slot
    double* pbox;    // [4][cap]   predicted boxes, by list position
    long long* gid;  // [cap] id assigned at birth (single-tracker API: the global id)
    int* tsu;        // time_since_update, by slot
    int* streak;     // hit_streak, by slot
    int* bframe;     // birth frame (global frame index), by slot
    int* bk;         // birth rank inside (frame, class), by slot
    int* order;      // list position -> slot   (the python list self.trackers)
    int* freel;      // free-slot stack
    int* trk_match;  // by list position: matched detection or -1
    int* det_match;  // by detection: list position, -1 never assigned, -2 assigned but rejected by the threshold
    int* new_list;   // unmatched detections in the reference's order
    float* cost_g;   // global cost scratch (capN x cap), used when the matrix does not fit in LDS
    int cap;         // max tracks
    int capN;        // max detections per frame
};

struct TrackerState {   // wave-uniform
    int n_tracks;
    int n_free;
    int frame_count;
    int next_local;
};

// ---- predict (sort.py:166-178 + filterpy predict): x = Fx, P = F P F^T + Q --------------------------------
__device__ __forceinline__ void kalman_predict(const TrackerMem& M, int slot, double box[4]) {
    const int cap = M.cap;
    double x[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = M.kx[i * cap + slot];
    if ((x[6] + x[2]) <= 0) x[6] *= 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = x[i] + x[i + 4];
#pragma unroll
    for (int i = 0; i < 7; ++i) M.kx[i * cap + slot] = x[i];
    // P: rows 0..2 get rows 4..6 added (F P), then columns 0..2 get columns 4..6 added ((FP) F^T), then + Q
    double P[49];
#pragma unroll
    for (int e = 0; e < 49; ++e) P[e] = M.kP[e * cap + slot];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) P[i * 7 + j] = P[i * 7 + j] + P[(i + 4) * 7 + j];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) P[i * 7 + j] = P[i * 7 + j] + P[i * 7 + j + 4];
#pragma unroll
    for (int i = 0; i < 7; ++i) P[i * 7 + i] = P[i * 7 + i] + kq(i);
#pragma unroll
    for (int e = 0; e < 49; ++e) M.kP[e * cap + slot] = P[e];
    x_to_bbox(x[0], x[1], x[2], x[3], box);
}

// numpy.linalg.inv on 4x4: LU with partial pivoting + per-column solves (same order as oracle inv4)
__device__ __forceinline__ void inv4(const double S[16], double SI[16]) {
    double A[16];
    int piv[4];
#pragma unroll
    for (int e = 0; e < 16; ++e) A[e] = S[e];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int p = k;
        double best = fabs(A[k * 4 + k]);
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            const double v = fabs(A[i * 4 + k]);
            if (v > best) { best = v; p = i; }
        }
        piv[k] = p;
        if (p != k) {
#pragma unroll
            for (int i = k + 1; i < 4; ++i) {       // static indexing only: swap row k with row i when i == p
                if (i == p) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const double tmp = A[k * 4 + j]; A[k * 4 + j] = A[i * 4 + j]; A[i * 4 + j] = tmp; }
                }
            }
        }
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            A[i * 4 + k] = A[i * 4 + k] / A[k * 4 + k];
#pragma unroll
            for (int j = k + 1; j < 4; ++j) A[i * 4 + j] = A[i * 4 + j] - A[i * 4 + k] * A[k * 4 + j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double b[4] = {0., 0., 0., 0.};
        b[j] = 1.;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = k + 1; i < 4; ++i)
                if (piv[k] == i) { const double tmp = b[k]; b[k] = b[i]; b[i] = tmp; }
        }
#pragma unroll
        for (int i = 1; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < i; ++k) b[i] = b[i] - A[i * 4 + k] * b[k];
#pragma unroll
        for (int i = 3; i >= 0; --i) {
#pragma unroll
            for (int k = i + 1; k < 4; ++k) b[i] = b[i] - A[i * 4 + k] * b[k];
            b[i] = b[i] / A[i * 4 + i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) SI[i * 4 + j] = b[i];
    }
}

// ---- update (sort.py:153-164 + filterpy update, Joseph form) ----------------------------------------------
__device__ __forceinline__ void kalman_update(const TrackerMem& M, int slot, const float det[4]) {
    const int cap = M.cap;
    double z[4], y[4], S[16], SI[16], K[28], P[49];
    bbox_to_z(det, z);
#pragma unroll
    for (int e = 0; e < 49; ++e) P[e] = M.kP[e * cap + slot];
    double x[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = M.kx[i * cap + slot];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = z[i] - x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) S[i * 4 + j] = (i == j) ? (P[i * 7 + j] + kr(i)) : P[i * 7 + j];
    inv4(S, SI);
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double acc = P[i * 7 + 0] * SI[0 * 4 + j];
#pragma unroll
            for (int k = 1; k < 4; ++k) acc = acc + P[i * 7 + k] * SI[k * 4 + j];
            K[i * 4 + j] = acc;
        }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        double acc = K[i * 4 + 0] * y[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) acc = acc + K[i * 4 + k] * y[k];
        M.kx[i * cap + slot] = x[i] + acc;
    }
    // A = I - K H  (only its first four columns differ from I);  B = A P;  P' = B A^T + (K R) K^T
    double B[49];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            double acc = (((i == 0) ? 1. : 0.) - K[i * 4 + 0]) * P[0 * 7 + j];
#pragma unroll
            for (int k = 1; k < 4; ++k) acc = acc + (((i == k) ? 1. : 0.) - K[i * 4 + k]) * P[k * 7 + j];
            if (i >= 4) acc = acc + P[i * 7 + j];
            B[i * 7 + j] = acc;
        }
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            double acc = B[i * 7 + 0] * (((j == 0) ? 1. : 0.) - K[j * 4 + 0]);
#pragma unroll
            for (int k = 1; k < 4; ++k) acc = acc + B[i * 7 + k] * (((j == k) ? 1. : 0.) - K[j * 4 + k]);
            if (j >= 4) acc = acc + B[i * 7 + j];
            double krk = (K[i * 4 + 0] * kr(0)) * K[j * 4 + 0];
#pragma unroll
            for (int k = 1; k < 4; ++k) krk = krk + (K[i * 4 + k] * kr(k)) * K[j * 4 + k];
            M.kP[(i * 7 + j) * cap + slot] = acc + krk;
        }
}

// ---- new track (sort.py:88-151) ---------------------------------------------------------------------------
__device__ __forceinline__ void track_init(const TrackerMem& M, int slot, const float det[4]) {
    const int cap = M.cap;
    double z[4];
    bbox_to_z(det, z);
#pragma unroll
    for (int i = 0; i < 7; ++i) M.kx[i * cap + slot] = (i < 4) ? z[i] : 0.;
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) M.kP[(i * 7 + j) * cap + slot] = (i == j) ? kp0(i) : 0.;
}

// ---- scikit-learn 0.22.2 Munkres (call site sort.py:206) on one wavefront ----------------------------------
// State (LDS): the float32 cost matrix C (n x m, n <= m, odd leading dimension ld so that a wave reading one column of
// 32 consecutive rows hits 32 distinct banks), one 64-bit zero bitmap word per (row, 64 columns) and the star /
// prime index arrays.  Row and column covers live in registers: lane w owns cover word w (64 rows / columns each).
// All full-matrix passes are row-parallel (lane = row, sequential over the row's columns); the search for "the first
// uncovered zero in row-major order" is a ballot over rows of (zero word & ~column cover) followed by a ctz.
// Round 4: HELPER WAVES for one tracker (config 1 at its stated size: 20 trackers on a 256-CU chip).  The tracker stays one wave (wave 0
// of the workgroup: every serial decision, every compaction by ballot is its own); three more waves of the workgroup sleep at the
// workgroup barrier and are woken for the phases that are plain data parallelism over the n x m cost matrix - today Munkres step 6
// (37 % of the tracker's cycles at one segment): wave w takes the columns c = w (mod 4) of every row, the four partial minima and the
// partial zero bitmaps meet in LDS.  Elementwise the same float operations in the same order: identical matrices, identical bitmaps.
// Protocol: wave 0 fills the job, sets cmd and reaches a workgroup barrier (__syncthreads - the helpers sleep at theirs, so this is what
// wakes them; wsync() is a wave-level fence + wave barrier and wakes nobody), everybody runs the share (help_step6_share: two more
// workgroup barriers inside), wave 0 resets cmd.  The single-wave code between jobs uses wsync() only, so the helpers sleep through it.
struct HelpJob {
    int cmd;                       // HELP_NOP / HELP_STEP6 / HELP_EXIT
    int n, m, ld, changed;
    float* C;                      // cost matrix (generic pointer) ...
    int c_lds;                     // ... 1: it lives in LDS (low 32 bits = LDS address), 0: in global memory - the shares use typed pointers
    unsigned long long rcov[2], cc[6];
    int W;                         // column-parallel step 6 (more than 128 columns): words per row ...
    unsigned long long* zmask;     // ... and where the new zero bitmaps of the rows go ([n][W], MunkresMem::zmask)
    float mn_part[4];
    unsigned long long zpart[4][128][2];
    float rowmin_part[4][128];     // step 1: per wave, the minimum of its columns of every row
    // IoU matrix (tracker_step): detections through the caller's accessor (copied bytewise), predicted boxes from the tracker's SoA arrays
    int T, N, transposed, cap;
    const double* pbox;
    alignas(8) char dets[64];
};
constexpr int HELP_NOP = 0, HELP_STEP6 = 1, HELP_EXIT = 2, HELP_STEP1 = 3, HELP_IOU = 4, HELP_STEP6C = 5;
constexpr int kHelpWaves = 4;
__host__ __device__ inline size_t help_lds_bytes() { return (sizeof(HelpJob) + 15) / 16 * 16; }

struct MunkresMem {       // LDS
    int* row_star;        // [n]  column of the star in the row, -1 none
    int* col_star;        // [m]
    int* row_prime;       // [n]
    unsigned long long* zmask;   // [n][W]  bit c of word w: C[r][64 w + c] == 0
    HelpJob* help;        // helper waves present (workgroup of kHelpWaves waves), nullptr = the tracker is alone
    int cost_in_lds;      // (helper jobs) the cost matrix handed to munkres_wave lives in LDS
};

__host__ __device__ inline int munkres_ld(int m) { return m | 1; }
__host__ __device__ inline size_t munkres_lds_bytes(int n_small, int n_big) {
    const size_t w = (size_t)(n_big + 63) / 64;
    return (((size_t)(2 * n_small + n_big) * sizeof(int) + 7) / 8) * 8 + (size_t)n_small * w * 8 + 16;
}
__device__ __forceinline__ MunkresMem munkres_mem(char* lds, int n_small, int n_big) {
    MunkresMem L;
    int* ip = reinterpret_cast<int*>(lds);
    L.row_star = ip;
    L.row_prime = ip + n_small;
    L.col_star = ip + 2 * n_small;
    const size_t off = (((size_t)(2 * n_small + n_big) * sizeof(int) + 7) / 8) * 8;
    L.zmask = reinterpret_cast<unsigned long long*>(lds + off);
    L.help = nullptr;
    L.cost_in_lds = 0;
    return L;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// Step 6 on the columns c = w (mod kHelpWaves): run by wave 0 and by the helpers, same barriers in the same order.  n, m <= 128.
using help_lds_f = __attribute__((address_space(3))) float*;
using help_glb_f = __attribute__((address_space(1))) float*;
#define HELP_DISPATCH(core, ...) do { if (J->c_lds) core(__VA_ARGS__, (help_lds_f)(uintptr_t)(unsigned)(uintptr_t)J->C); \
                                      else core(__VA_ARGS__, (help_glb_f)(uintptr_t)J->C); } while (0)

template <class CP>
__device__ __forceinline__ void help_step6_core(HelpJob* J, int w, CP C) {
    const int lane = threadIdx.x & 63;
    const int n = J->n, m = J->m, ld = J->ld;
    const unsigned long long rc0 = J->rcov[0], rc1 = J->rcov[1], cc0 = J->cc[0], cc1 = J->cc[1];
    // smallest uncovered value of this wave's columns
    float mn = __builtin_inff();
    for (int j = 0; j < 2; ++j) {
        const int r = j * kWave + lane;
        const bool unc = (r < n) && !(((j ? rc1 : rc0) >> lane) & 1ull);
        if (unc) {
            for (int c = w; c < m; c += kHelpWaves) {
                const bool ccov = (((c >> 6) ? cc1 : cc0) >> (c & 63)) & 1ull;         // wave-uniform
                if (ccov) continue;
                const float v = C[r * ld + c];
                mn = (v < mn) ? v : mn;
            }
        }
    }
    mn = wave_min_f(mn);
    if (lane == 0) J->mn_part[w] = mn;
    __syncthreads();
    mn = J->mn_part[0];
#pragma unroll
    for (int q = 1; q < kHelpWaves; ++q) { const float o = J->mn_part[q]; mn = (o < mn) ? o : mn; }
    if (J->changed) {
        for (int j = 0; j < 2; ++j) {
            const int r = j * kWave + lane;
            if (r < n) {
                const bool rcv = ((j ? rc1 : rc0) >> lane) & 1ull;
                unsigned long long z0 = 0ull, z1 = 0ull;
                for (int c = w; c < m; c += kHelpWaves) {
                    const bool ccov = (((c >> 6) ? cc1 : cc0) >> (c & 63)) & 1ull;
                    float v = C[r * ld + c];
                    if (rcv || !ccov) {
                        if (rcv) v = v + mn;
                        if (!ccov) v = v - mn;
                        C[r * ld + c] = v;
                    }
                    const unsigned long long bit = (v == 0.f) ? (1ull << (c & 63)) : 0ull;
                    if (c >> 6) z1 |= bit; else z0 |= bit;
                }
                J->zpart[w][r][0] = z0;
                J->zpart[w][r][1] = z1;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void help_step6_share(HelpJob* J, int w) { HELP_DISPATCH(help_step6_core, J, w); }

// Step 6 in its COLUMN-parallel form (more than 128 columns: the online pipeline's 100 x 300 problems; lane = column, rows one after the other):
// wave q takes the rows r = q (mod kHelpWaves); the new zero bitmap words of a row are ballots, written to zmask for wave 0 to pick up.
template <class CP>
__device__ __forceinline__ void help_step6c_core(HelpJob* J, int q, CP C) {
    const int lane = threadIdx.x & 63;
    const int n = J->n, m = J->m, ld = J->ld, W = J->W;
    const unsigned long long rc0 = J->rcov[0], rc1 = J->rcov[1];
    unsigned long long* zmask = J->zmask;
    float mn = __builtin_inff();
    for (int r = q; r < n; r += kHelpWaves) {
        if ((((r >> 6) ? rc1 : rc0) >> (r & 63)) & 1ull) continue;             // covered row (wave-uniform)
        for (int w = 0; w < W; ++w) {
            const int c = 64 * w + lane;
            if (c < m && !((J->cc[w] >> lane) & 1ull)) {
                const float v = C[r * ld + c];
                mn = (v < mn) ? v : mn;
            }
        }
    }
    mn = wave_min_f(mn);
    if (lane == 0) J->mn_part[q] = mn;
    __syncthreads();
    mn = J->mn_part[0];
#pragma unroll
    for (int k = 1; k < kHelpWaves; ++k) { const float o = J->mn_part[k]; mn = (o < mn) ? o : mn; }
    if (J->changed) {
        for (int r = q; r < n; r += kHelpWaves) {
            const bool rcv = (((r >> 6) ? rc1 : rc0) >> (r & 63)) & 1ull;        // wave-uniform
            for (int w = 0; w < W; ++w) {
                const int c = 64 * w + lane;
                const bool valid = c < m;
                const bool ccov = (J->cc[w] >> lane) & 1ull;
                float v = 1.f;
                if (valid) {
                    v = C[r * ld + c];
                    if (rcv || !ccov) {
                        if (rcv) v = v + mn;
                        if (!ccov) v = v - mn;
                        C[r * ld + c] = v;
                    }
                }
                const unsigned long long zz = __ballot(valid && v == 0.f);
                if (lane == 0) zmask[(size_t)r * W + w] = zz;
            }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void help_step6c_share(HelpJob* J, int w) { HELP_DISPATCH(help_step6c_core, J, w); }

// Step 1 on the columns c = w (mod kHelpWaves): partial row minima, then subtract the row minimum and report the zero bits of the own columns.
template <class CP>
__device__ __forceinline__ void help_step1_core(HelpJob* J, int w, CP C) {
    const int lane = threadIdx.x & 63;
    const int n = J->n, m = J->m, ld = J->ld;
    for (int j = 0; j < 2; ++j) {
        const int r = j * kWave + lane;
        if (r < n) {
            // (wave 0 starts from the value of column 0, like the sequential loop: a NaN there propagates, a NaN elsewhere is skipped)
            float mn = (w == 0) ? C[r * ld] : __builtin_inff();
            for (int c = w; c < m; c += kHelpWaves) { const float v = C[r * ld + c]; mn = (v < mn) ? v : mn; }
            J->rowmin_part[w][r] = mn;
        }
    }
    __syncthreads();
    for (int j = 0; j < 2; ++j) {
        const int r = j * kWave + lane;
        if (r < n) {
            float mn = J->rowmin_part[0][r];
#pragma unroll
            for (int q = 1; q < kHelpWaves; ++q) { const float o = J->rowmin_part[q][r]; mn = (o < mn) ? o : mn; }
            unsigned long long z0 = 0ull, z1 = 0ull;
            for (int c = w; c < m; c += kHelpWaves) {
                const float v = C[r * ld + c] - mn;
                C[r * ld + c] = v;
                const unsigned long long bit = (v == 0.f) ? (1ull << (c & 63)) : 0ull;
                if (c >> 6) z1 |= bit; else z0 |= bit;
            }
            J->zpart[w][r][0] = z0;
            J->zpart[w][r][1] = z1;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void help_step1_share(HelpJob* J, int w) { HELP_DISPATCH(help_step1_core, J, w); }

// IoU matrix: wave w takes the tracks t = w (mod kHelpWaves); lane = detection, the wave's tracks are held lane-wise and broadcast by readlane
// exactly as in the single-wave loop of tracker_step (same iou_det_trk per pair: identical float32 entries)
template <class Dets, class CP>
__device__ __forceinline__ void help_iou_core(HelpJob* J, int w, CP C) {
    const int lane = threadIdx.x & 63;
    const int T = J->T, N = J->N, ld = J->ld, cap = J->cap;
    const bool transposed = J->transposed != 0;
    const double* pbox = J->pbox;
    Dets dets;
    __builtin_memcpy(&dets, J->dets, sizeof(Dets));
    for (int dbase = 0; dbase < N; dbase += kWave) {
        const int d = dbase + lane;
        float db[4] = {0.f, 0.f, 0.f, 0.f};
        if (d < N) dets.get(d, db);
        for (int tbase = 0; tbase < T; tbase += kWave) {
            const int tl = tbase + lane;
            double tbx[4] = {0., 0., 0., 0.};
            if (tl < T) {
#pragma unroll
                for (int q = 0; q < 4; ++q) tbx[q] = pbox[q * cap + tl];
            }
            const int tcnt = (T - tbase) < kWave ? (T - tbase) : kWave;
            for (int tt = w; tt < tcnt; tt += kHelpWaves) {
                double tb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned long long bits = readlane64((unsigned long long)__double_as_longlong(tbx[q]), tt);
                    tb[q] = __longlong_as_double((long long)bits);
                }
                const float v = (float)iou_det_trk(db, tb);
                if (d < N) {
                    const int t = tbase + tt;
                    const int r = transposed ? t : d, c = transposed ? d : t;
                    C[r * ld + c] = -v;
                }
            }
        }
    }
    __syncthreads();
}

template <class Dets>
__device__ __forceinline__ void help_iou_share(HelpJob* J, int w) {
    if (J->c_lds) help_iou_core<Dets>(J, w, (help_lds_f)(uintptr_t)(unsigned)(uintptr_t)J->C);
    else help_iou_core<Dets>(J, w, (help_glb_f)(uintptr_t)J->C);
}

// the life of a helper wave: sleep at the barrier, look at the command, take its share, sleep again
template <class Dets>
__device__ __forceinline__ void helper_loop(HelpJob* J, int w) {
    for (;;) {
        __syncthreads();
        const int cmd = J->cmd;
        if (cmd == HELP_EXIT) return;
        if (cmd == HELP_STEP6) help_step6_share(J, w);
        else if (cmd == HELP_STEP1) help_step1_share(J, w);
        else if (cmd == HELP_IOU) help_iou_share<Dets>(J, w);
        else if (cmd == HELP_STEP6C) help_step6c_share(J, w);
    }
}

#ifdef WT_PHASE_TIMING
__device__ unsigned long long wt_phase[12];
#define WT_T0 long long _t = clock64();
#define WT_TICK(i) { const long long _n = clock64(); if ((threadIdx.x & 63) == 0) atomicAdd(&wt_phase[i], (unsigned long long)(_n - _t)); _t = _n; }
#else
#define WT_T0
#define WT_TICK(i)
#endif

// Register variant for n, m <= 64 * RM / 64 * WM (the common case: <= 128 boxes per class and frame).  Same algorithm, same
// visiting order and tie-breaks as munkres_wave below - only the storage differs: lane l keeps the zero bitmaps of ITS rows
// (l, l + 64, ...) in VGPRs and the row / column covers are wave-uniform scalars, so the hot loop of step 4 ("first uncovered
// zero in row-major order", executed O(n^2) times per frame) runs on ballots / readlanes / scalar bit operations with a single
// LDS access (row_star) instead of ~6 dependent LDS round trips.
template <int RM, int WM, bool HELP = false, class CostPtr>
__device__ int munkres_wave_reg(CostPtr C, int n, int m, int ld, const MunkresMem& L) {
    const int lane = threadIdx.x & 63;
    const int W = (m + 63) >> 6, R = (n + 63) >> 6;
    unsigned long long z[RM][WM];
    unsigned long long cc[WM], rcov[RM];                      // wave-uniform cover bitmaps
#pragma unroll
    for (int j = 0; j < RM; ++j)
#pragma unroll
        for (int w = 0; w < WM; ++w) z[j][w] = 0ull;
    WT_T0
    for (int c = lane; c < m; c += kWave) L.col_star[c] = -1;
    bool step1_done = false;
    if constexpr (HELP && RM <= 2 && WM <= 2) {
        if (L.help) {                                // helper waves: a quarter of the columns per wave (HelpJob)
            HelpJob* J = L.help;
            wsync();                                 // the cost matrix written by this wave is visible to the others behind the barrier
            if (lane == 0) { J->n = n; J->m = m; J->ld = ld; J->C = (float*)C; J->c_lds = L.cost_in_lds; J->cmd = HELP_STEP1; }
            __syncthreads();
            help_step1_share(J, 0);
#pragma unroll
            for (int j = 0; j < RM; ++j) {
                const int r = j * kWave + lane;
                if (j < R && r < n) {
#pragma unroll
                    for (int w = 0; w < WM; ++w) {
                        if (w >= W) continue;
                        unsigned long long zz = 0ull;
#pragma unroll
                        for (int q = 0; q < kHelpWaves; ++q) zz |= J->zpart[q][r][w];
                        z[j][w] = zz;
                    }
                    L.row_star[r] = -1;
                    L.row_prime[r] = -1;
                }
            }
            step1_done = true;
        }
    }
    // step 1: subtract row minima and build the zero bitmaps (row-parallel)
#pragma unroll
    for (int j = 0; j < RM; ++j) {
        const int r = j * kWave + lane;
        if (!step1_done && j < R && r < n) {
            float mn = C[r * ld];
            for (int c = 1; c < m; ++c) { const float v = C[r * ld + c]; mn = (v < mn) ? v : mn; }
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                if (w >= W) continue;
                unsigned long long zz = 0ull;
                const int c1 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                for (int c = 0; c < c1; ++c) {
                    const float v = C[r * ld + 64 * w + c] - mn;
                    C[r * ld + 64 * w + c] = v;
                    zz |= (v == 0.f) ? (1ull << c) : 0ull;
                }
                z[j][w] = zz;
            }
            L.row_star[r] = -1;
            L.row_prime[r] = -1;
        }
    }
    wsync();
    WT_TICK(9)
    // zero word w of row r (r wave-uniform)
    auto zrow = [&](int r, int w) -> unsigned long long {
        unsigned long long v = 0ull;
#pragma unroll
        for (int j = 0; j < RM; ++j)
#pragma unroll
            for (int ww = 0; ww < WM; ++ww)
                if (j == (r >> 6) && ww == w) v = readlane64(z[j][ww], r & 63);
        return v;
    };
    auto get = [&](const unsigned long long (&a)[WM], int w) -> unsigned long long {
        unsigned long long v = 0ull;
#pragma unroll
        for (int ww = 0; ww < WM; ++ww) if (ww == w) v = a[ww];
        return v;
    };
#pragma unroll
    for (int w = 0; w < WM; ++w) cc[w] = 0ull;
    // greedy stars in row-major order (serial over rows)
    for (int r = 0; r < n; ++r) {
        int first = -1;
        for (int w = 0; w < W && first < 0; ++w) {
            const unsigned long long a = zrow(r, w) & ~get(cc, w);
            if (a) first = 64 * w + __builtin_ctzll(a);
        }
        if (first >= 0) {
            if (lane == 0) { L.row_star[r] = first; L.col_star[first] = r; }
#pragma unroll
            for (int w = 0; w < WM; ++w) if (w == (first >> 6)) cc[w] |= 1ull << (first & 63);
        }
    }
    wsync();
    WT_TICK(6)
    long guard = 0;
    const long guard_max = 64L + 8L * (long)(n + m) * (long)(n + m) * (long)(n + 1);
    for (;;) {
        // step 3: cover the starred columns; done when every row has a star
        int stars = 0;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
            cc[w] = 0ull;
            if (w < W) {
                const int c = 64 * w + lane;
                cc[w] = __ballot((c < m) && L.col_star[c] >= 0);
                stars += __popcll(cc[w]);
            }
        }
#pragma unroll
        for (int j = 0; j < RM; ++j) rcov[j] = 0ull;
        if (stars >= n) { WT_TICK(8) return 0; }
        // step 4 (+ step 6)
        int z0r = -1, z0c = -1;
        for (;;) {
            if (++guard > guard_max) return kErrNumeric;
            int fr = -1, fc = -1;
#pragma unroll
            for (int j = 0; j < RM; ++j) {
                if (j < R && fr < 0) {
                    const int r = j * kWave + lane;
                    bool has = false;
                    if (r < n && !((rcov[j] >> lane) & 1ull)) {
#pragma unroll
                        for (int w = 0; w < WM; ++w) has = has || ((z[j][w] & ~cc[w]) != 0ull);    // z is 0 beyond W
                    }
                    const unsigned long long mask = __ballot(has);
                    if (mask) fr = j * kWave + __builtin_ctzll(mask);
                }
            }
            if (fr >= 0) {
                for (int w = 0; w < W && fc < 0; ++w) {
                    const unsigned long long a = zrow(fr, w) & ~get(cc, w);
                    if (a) fc = 64 * w + __builtin_ctzll(a);
                }
            }
            if (fr < 0) {
                WT_TICK(8)
                if constexpr (WM <= 2 && RM <= 2) {
                if (HELP && L.help) {
                    // helper waves present: the same step on a quarter of the columns per wave (see HelpJob)
                    HelpJob* J = L.help;
                    bool any_r = false, any_c = false;
#pragma unroll
                    for (int w = 0; w < WM; ++w) {
                        if (w >= W) continue;
                        const int c1 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                        const unsigned long long valid = (c1 == 64) ? ~0ull : ((1ull << c1) - 1ull);
                        any_c = any_c || ((~cc[w] & valid) != 0ull);
                    }
#pragma unroll
                    for (int j = 0; j < RM; ++j) {
                        if (j >= R) continue;
                        const int r = j * kWave + lane;
                        any_r = any_r || (__ballot((r < n) && !((rcov[j] >> lane) & 1ull)) != 0ull);
                    }
                    if (lane == 0) {
                        J->n = n; J->m = m; J->ld = ld; J->C = (float*)C; J->c_lds = L.cost_in_lds; J->changed = (any_r && any_c) ? 1 : 0;
                        J->rcov[0] = rcov[0]; J->rcov[1] = RM > 1 ? rcov[RM - 1] : 0ull;
                        J->cc[0] = cc[0]; J->cc[1] = WM > 1 ? cc[WM - 1] : 0ull;
                        J->cmd = HELP_STEP6;
                    }
                    __syncthreads();                    // wakes the helpers
                    help_step6_share(J, 0);
                    if (lane == 0) J->cmd = HELP_NOP;
                    if (any_r && any_c) {
#pragma unroll
                        for (int j = 0; j < RM; ++j) {
                            if (j >= R) continue;
                            const int r = j * kWave + lane;
                            if (r < n) {
#pragma unroll
                                for (int w = 0; w < WM; ++w) {
                                    if (w >= W) continue;
                                    unsigned long long zz = 0ull;
#pragma unroll
                                    for (int q = 0; q < kHelpWaves; ++q) zz |= J->zpart[q][r][w];
                                    z[j][w] = zz;
                                }
                            }
                        }
                    }
                    wsync();
                    WT_TICK(7)
                    continue;
                }
                // (<= 128 columns: ROW-parallel - lane = row, serial over the row's columns; faster for the small LDS-resident matrices of
                //  the batch stages: 31.7 k vs 19.5 k frames/s on the 1-segment track stage)
                    // step 6: smallest uncovered value; add it to covered rows, subtract it from uncovered columns
                    float mn = __builtin_inff();
                    bool any_r = false, any_c = false;
    #pragma unroll
                    for (int w = 0; w < WM; ++w) {
                        if (w >= W) continue;
                        const int c1 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                        const unsigned long long valid = (c1 == 64) ? ~0ull : ((1ull << c1) - 1ull);
                        any_c = any_c || ((~cc[w] & valid) != 0ull);
                    }
    #pragma unroll
                    for (int j = 0; j < RM; ++j) {
                        if (j >= R) continue;
                        const int r = j * kWave + lane;
                        const bool unc = (r < n) && !((rcov[j] >> lane) & 1ull);
                        any_r = any_r || (__ballot(unc) != 0ull);
                        if (unc) {
    #pragma unroll
                            for (int w = 0; w < WM; ++w) {
                                if (w >= W) continue;
                                unsigned long long todo = ~cc[w];
                                const int c1 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                                if (c1 < 64) todo &= (1ull << c1) - 1ull;
                                while (todo) {
                                    const int c = __builtin_ctzll(todo);
                                    todo &= todo - 1ull;
                                    const float v = C[r * ld + 64 * w + c];
                                    mn = (v < mn) ? v : mn;
                                }
                            }
                        }
                    }
                    mn = wave_min_f(mn);
                    if (any_r && any_c) {
    #pragma unroll
                        for (int j = 0; j < RM; ++j) {
                            if (j >= R) continue;
                            const int r = j * kWave + lane;
                            if (r < n) {
                                const bool rcv = (rcov[j] >> lane) & 1ull;
    #pragma unroll
                                for (int w = 0; w < WM; ++w) {
                                    if (w >= W) continue;
                                    const unsigned long long cw = cc[w];
                                    const int c1 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                                    unsigned long long zz = 0ull;
                                    for (int c = 0; c < c1; ++c) {
                                        float v = C[r * ld + 64 * w + c];
                                        const bool ccov = (cw >> c) & 1ull;
                                        if (rcv || !ccov) {
                                            if (rcv) v = v + mn;
                                            if (!ccov) v = v - mn;
                                            C[r * ld + 64 * w + c] = v;
                                        }
                                        zz |= (v == 0.f) ? (1ull << c) : 0ull;
                                    }
                                    z[j][w] = zz;
                                }
                            }
                        }
                    }
                } else {
                    if (HELP && L.help && WM <= 6) {
                        // helper waves: a quarter of the rows per wave (help_step6c_core)
                        HelpJob* J = L.help;
                        bool any_r = false, any_c = false;
#pragma unroll
                        for (int w = 0; w < WM; ++w) {
                            if (w >= W) continue;
                            const int c1 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                            const unsigned long long valid = (c1 == 64) ? ~0ull : ((1ull << c1) - 1ull);
                            any_c = any_c || ((~cc[w] & valid) != 0ull);
                        }
#pragma unroll
                        for (int j = 0; j < RM; ++j) {
                            if (j >= R) continue;
                            const int r = j * kWave + lane;
                            any_r = any_r || (__ballot((r < n) && !((rcov[j] >> lane) & 1ull)) != 0ull);
                        }
                        if (lane == 0) {
                            J->n = n; J->m = m; J->ld = ld; J->W = W; J->C = (float*)C; J->c_lds = L.cost_in_lds; J->changed = (any_r && any_c) ? 1 : 0;
                            J->zmask = L.zmask;
                            J->rcov[0] = rcov[0]; J->rcov[1] = RM > 1 ? rcov[RM - 1] : 0ull;
#pragma unroll
                            for (int w = 0; w < WM; ++w) J->cc[w] = cc[w];
                            J->cmd = HELP_STEP6C;
                        }
                        __syncthreads();                    // wakes the helpers
                        help_step6c_share(J, 0);
                        if (lane == 0) J->cmd = HELP_NOP;
                        if (any_r && any_c) {
#pragma unroll
                            for (int j = 0; j < RM; ++j) {
                                if (j >= R) continue;
                                const int r = j * kWave + lane;
                                if (r < n) {
#pragma unroll
                                    for (int w = 0; w < WM; ++w) if (w < W) z[j][w] = L.zmask[(size_t)r * W + w];
                                }
                            }
                        }
                        wsync();
                        WT_TICK(7)
                        continue;
                    }
                    // step 6: smallest uncovered value; add it to covered rows, subtract it from uncovered columns.
                    // Round 3: COLUMN-parallel (lane = column, rows visited one after the other): the cost matrix of the online pipeline
                    // (100 x 300, 120 KB) lives in global memory, where the old row-per-lane walk touched 64 cache lines per load
                    // instruction - 90 % of the tracker's cycles on the end-to-end workload.  Elementwise the same float operations
                    // in the same order (v + mn first, then - mn), min is exact: identical matrices, identical zero bitmaps.
                    float mn = __builtin_inff();
                    bool any_r = false, any_c = false;
    #pragma unroll
                    for (int w = 0; w < WM; ++w) {
                        if (w >= W) continue;
                        const int c1 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                        const unsigned long long valid = (c1 == 64) ? ~0ull : ((1ull << c1) - 1ull);
                        any_c = any_c || ((~cc[w] & valid) != 0ull);
                    }
                    for (int r = 0; r < n; ++r) {
                        unsigned long long rc = 0ull;
    #pragma unroll
                        for (int j = 0; j < RM; ++j) if (j == (r >> 6)) rc = rcov[j];
                        if ((rc >> (r & 63)) & 1ull) continue;                      // covered row (wave-uniform)
                        any_r = true;
    #pragma unroll
                        for (int w = 0; w < WM; ++w) {
                            if (w >= W) continue;
                            const int c = 64 * w + lane;
                            if (c < m && !((cc[w] >> lane) & 1ull)) {
                                const float v = C[r * ld + c];
                                mn = (v < mn) ? v : mn;
                            }
                        }
                    }
                    mn = wave_min_f(mn);
                    if (any_r && any_c) {
                        for (int r = 0; r < n; ++r) {
                            unsigned long long rc = 0ull;
    #pragma unroll
                            for (int j = 0; j < RM; ++j) if (j == (r >> 6)) rc = rcov[j];
                            const bool rcv = (rc >> (r & 63)) & 1ull;                 // wave-uniform
    #pragma unroll
                            for (int w = 0; w < WM; ++w) {
                                if (w >= W) continue;
                                const int c = 64 * w + lane;
                                const bool valid = c < m;
                                const bool ccov = (cc[w] >> lane) & 1ull;
                                float v = 1.f;
                                if (valid) {
                                    v = C[r * ld + c];
                                    if (rcv || !ccov) {
                                        if (rcv) v = v + mn;
                                        if (!ccov) v = v - mn;
                                        C[r * ld + c] = v;
                                    }
                                }
                                const unsigned long long zz = __ballot(valid && v == 0.f);
                                if (lane == (r & 63)) {
    #pragma unroll
                                    for (int j = 0; j < RM; ++j) if (j == (r >> 6)) z[j][w] = zz;
                                }
                            }
                        }
                    }
                }
                wsync();
                WT_TICK(7)
                continue;
            }
            const int sc = uni(L.row_star[fr]);
            if (lane == 0) L.row_prime[fr] = fc;
            if (sc < 0) { z0r = fr; z0c = fc; break; }
#pragma unroll
            for (int j = 0; j < RM; ++j) if (j == (fr >> 6)) rcov[j] |= 1ull << (fr & 63);
#pragma unroll
            for (int w = 0; w < WM; ++w) if (w == (sc >> 6)) cc[w] &= ~(1ull << (sc & 63));
        }
        wsync();
        // step 5: augmenting path (serial, lane 0); then erase primes
        if (lane == 0) {
            int pr = z0r, pc = z0c;
            for (long it = 0; it <= (long)n + m; ++it) {
                const int r2 = L.col_star[pc];
                L.row_star[pr] = pc;
                L.col_star[pc] = pr;
                if (r2 < 0) break;
                pr = r2;
                pc = L.row_prime[r2];
            }
        }
        wsync();
        for (int r = lane; r < n; r += kWave) L.row_prime[r] = -1;
        wsync();
    }
}

// C: n x m float32 with leading dimension ld (n <= m <= 4096), modified in place.  Returns 0 or kErrNumeric.
template <bool HELP = false, class CostPtr>
__device__ int munkres_wave(CostPtr C, int n, int m, int ld, const MunkresMem& L) {
    if (n <= 128 && m <= 128) return munkres_wave_reg<2, 2, HELP>(C, n, m, ld, L);
    // round 3: the online detect -> track pipeline keeps (max_age + 2) x 100 track slots per class: up to 100 x 400 problems when the
    // detector's boxes do not persist.  Same code with wider column bitmaps (zero bitmaps 12 VGPR pairs per lane, covers scalar).
    // (<2, 7> and <2, 4> + <2, 7> trip a hipcc 7.2 backend error "V_CMP_NE_U32_e32 0, $src_shared_base"; <2, 6> compiles)
    if (n <= 128 && m <= 384) return munkres_wave_reg<2, 6, HELP>(C, n, m, ld, L);     // (HELP: one instantiation per kernel - a shared one is not inlined)
    const int lane = threadIdx.x & 63;
    const int W = (m + 63) >> 6;
    for (int c = lane; c < m; c += kWave) L.col_star[c] = -1;
    // step 1: subtract row minima and build the zero bitmaps (row-parallel)
    for (int r0 = 0; r0 < n; r0 += kWave) {
        const int r = r0 + lane;
        if (r < n) {
            float mn = C[r * ld];
            for (int c = 1; c < m; ++c) { const float v = C[r * ld + c]; mn = (v < mn) ? v : mn; }
            for (int w = 0; w < W; ++w) {
                unsigned long long z = 0ull;
                const int c1 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                for (int c = 0; c < c1; ++c) {
                    const float v = C[r * ld + 64 * w + c] - mn;
                    C[r * ld + 64 * w + c] = v;
                    z |= (v == 0.f) ? (1ull << c) : 0ull;
                }
                L.zmask[(size_t)r * W + w] = z;
            }
            L.row_star[r] = -1;
            L.row_prime[r] = -1;
        }
    }
    wsync();
    // greedy stars in row-major order (serial over rows; lane w keeps column-cover word w)
    unsigned long long colcov = 0ull, rowcov = 0ull;
    for (int r = 0; r < n; ++r) {
        int first = -1;
        for (int w = 0; w < W && first < 0; ++w) {
            const unsigned long long a = L.zmask[(size_t)r * W + w] & ~readlane64(colcov, w);
            if (a) first = 64 * w + __builtin_ctzll(a);
        }
        if (first >= 0) {
            if (lane == 0) { L.row_star[r] = first; L.col_star[first] = r; }
            if (lane == (first >> 6)) colcov |= 1ull << (first & 63);
        }
    }
    wsync();
    long guard = 0;
    const long guard_max = 64L + 8L * (long)(n + m) * (long)(n + m) * (long)(n + 1);
    for (;;) {
        // step 3: cover the starred columns; done when every row has a star
        int stars = 0;
        colcov = 0ull;
        for (int w = 0; w < W; ++w) {
            const int c = 64 * w + lane;
            const unsigned long long word = __ballot((c < m) && L.col_star[c] >= 0);
            if (lane == w) colcov = word;
            stars += __popcll(word);
        }
        rowcov = 0ull;
        if (stars >= n) return 0;
        // step 4 (+ step 6)
        int z0r = -1, z0c = -1;
        for (;;) {
            if (++guard > guard_max) return kErrNumeric;
            int fr = -1, fc = -1;
            for (int r0 = 0; r0 < n && fr < 0; r0 += kWave) {
                const int r = r0 + lane;
                const unsigned long long rc = readlane64(rowcov, r0 >> 6);
                bool has = false;
                if (r < n && !((rc >> lane) & 1ull))
                    for (int w = 0; w < W; ++w) has = has || ((L.zmask[(size_t)r * W + w] & ~readlane64(colcov, w)) != 0ull);
                const unsigned long long mask = __ballot(has);
                if (mask) fr = r0 + __builtin_ctzll(mask);
            }
            if (fr >= 0) {
                for (int w = 0; w < W && fc < 0; ++w) {
                    const unsigned long long a = L.zmask[(size_t)fr * W + w] & ~readlane64(colcov, w);
                    if (a) fc = 64 * w + __builtin_ctzll(a);
                }
            }
            if (fr < 0) {
                // step 6: smallest uncovered value; add it to covered rows, subtract it from uncovered columns
                float mn = __builtin_inff();
                bool any_r = false, any_c = false;
                for (int w = 0; w < W; ++w) {
                    const unsigned long long cw = readlane64(colcov, w);
                    const int c1 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                    const unsigned long long valid = (c1 == 64) ? ~0ull : ((1ull << c1) - 1ull);
                    any_c = any_c || ((~cw & valid) != 0ull);
                }
                for (int r0 = 0; r0 < n; r0 += kWave) {
                    const int r = r0 + lane;
                    const unsigned long long rc = readlane64(rowcov, r0 >> 6);
                    const bool unc = (r < n) && !((rc >> lane) & 1ull);
                    any_r = any_r || (__ballot(unc) != 0ull);
                    if (unc)
                        for (int w = 0; w < W; ++w) {
                            unsigned long long todo = ~readlane64(colcov, w);
                            const int c1 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                            if (c1 < 64) todo &= (1ull << c1) - 1ull;
                            while (todo) {
                                const int c = __builtin_ctzll(todo);
                                todo &= todo - 1ull;
                                const float v = C[r * ld + 64 * w + c];
                                mn = (v < mn) ? v : mn;
                            }
                        }
                }
                mn = wave_min_f(mn);
                if (any_r && any_c) {
                    for (int r0 = 0; r0 < n; r0 += kWave) {
                        const int r = r0 + lane;
                        const unsigned long long rc = readlane64(rowcov, r0 >> 6);
                        if (r < n) {
                            const bool rcov = (rc >> lane) & 1ull;
                            for (int w = 0; w < W; ++w) {
                                const unsigned long long cw = readlane64(colcov, w);
                                const int c1 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                                unsigned long long z = 0ull;
                                for (int c = 0; c < c1; ++c) {
                                    float v = C[r * ld + 64 * w + c];
                                    const bool ccov = (cw >> c) & 1ull;
                                    if (rcov || !ccov) {
                                        if (rcov) v = v + mn;
                                        if (!ccov) v = v - mn;
                                        C[r * ld + 64 * w + c] = v;
                                    }
                                    z |= (v == 0.f) ? (1ull << c) : 0ull;
                                }
                                L.zmask[(size_t)r * W + w] = z;
                            }
                        }
                    }
                }
                wsync();
                continue;
            }
            const int sc = uni(L.row_star[fr]);
            if (lane == 0) L.row_prime[fr] = fc;
            if (sc < 0) { z0r = fr; z0c = fc; break; }
            if (lane == (fr >> 6)) rowcov |= 1ull << (fr & 63);
            if (lane == (sc >> 6)) colcov &= ~(1ull << (sc & 63));
        }
        wsync();
        // step 5: augmenting path (serial, lane 0); then erase primes
        if (lane == 0) {
            int pr = z0r, pc = z0c;
            for (long it = 0; it <= (long)n + m; ++it) {
                const int r2 = L.col_star[pc];
                L.row_star[pr] = pc;
                L.col_star[pc] = pr;
                if (r2 < 0) break;
                pr = r2;
                pc = L.row_prime[r2];
            }
        }
        wsync();
        for (int r = lane; r < n; r += kWave) L.row_prime[r] = -1;
        wsync();
    }
}

// One frame of one tracker: sort.py:244-296.  Dets::get(k, float[4]) yields the k-th detection of this class
// as the float32 row the reference builds (tracker_sort.py:45); Emit receives the rows of sort.py:286-288.
template <bool HELP = false, class Dets, class Emit>
__device__ int tracker_step(const TrackerMem& M, TrackerState& S, const MunkresMem& L, float* lds_cost,
                            int lds_cost_cap, const Dets& dets, int N, double iou_thr, int max_age, int min_hits,
                            int frame_global, long long id_base, Emit& emit, int* n_births, int* n_rows) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = lanemask_lt();
    const int cap = M.cap;
    S.frame_count += 1;
    WT_T0
    // ---- predict; tracks with a non-finite predicted box are dropped (sort.py:256-265) ----
    int T = 0;
    {
        const int T0 = S.n_tracks;
        for (int base = 0; base < T0; base += kWave) {
            const int i = base + lane;
            const bool act = i < T0;
            int slot = 0;
            bool ok = false;
            double b[4] = {0., 0., 0., 0.};
            if (act) {
                slot = M.order[i];
                kalman_predict(M, slot, b);
                const int tsu = M.tsu[slot];
                if (tsu > 0) M.streak[slot] = 0;
                M.tsu[slot] = tsu + 1;
                ok = finite_d(b[0]) && finite_d(b[1]) && finite_d(b[2]) && finite_d(b[3]);
            }
            const unsigned long long good = __ballot(act && ok);
            const unsigned long long bad = __ballot(act && !ok);
            if (act && ok) {
                const int np = T + __popcll(good & lt);
                M.order[np] = slot;
#pragma unroll
                for (int c = 0; c < 4; ++c) M.pbox[c * cap + np] = b[c];
            }
            if (act && !ok) M.freel[S.n_free + __popcll(bad & lt)] = slot;
            T += __popcll(good);
            S.n_free += __popcll(bad);
        }
    }
    for (int i = lane; i < T; i += kWave) M.trk_match[i] = -1;
    for (int k = lane; k < N; k += kWave) M.det_match[k] = -1;
    wsync();
    WT_TICK(0)
    // ---- associate (sort.py:193-230) ----
    if (T > 0 && N > 0) {
        const bool transposed = T < N;                 // Munkres works on rows <= cols
        const int n = transposed ? T : N, m = transposed ? N : T;
        const int ld = munkres_ld(m);
        const bool in_lds = (long)n * ld <= (long)lds_cost_cap;
        if (!in_lds && !M.cost_g) return kErrCapacity;
        float* C = in_lds ? lds_cost : M.cost_g;
        static_assert(sizeof(Dets) <= 64, "HelpJob carries the detection accessor bytewise");
        if (HELP && L.help) {
            // helper waves: every wave takes a quarter of the tracks (HelpJob); the predicted boxes written above are visible behind the barrier
            HelpJob* J = L.help;
            if (lane == 0) {
                J->T = T; J->N = N; J->transposed = transposed ? 1 : 0; J->cap = cap; J->ld = ld; J->C = C; J->c_lds = in_lds ? 1 : 0; J->pbox = M.pbox;
                __builtin_memcpy(J->dets, &dets, sizeof(Dets));
                J->cmd = HELP_IOU;
            }
            __syncthreads();
            help_iou_share<Dets>(J, 0);
        } else
        // lane = detection (its float32 row is fetched once: two dependent global loads), predicted track boxes are held
        // lane-wise in registers and broadcast one by one with readlane: no memory access in the N x T inner loop
        for (int dbase = 0; dbase < N; dbase += kWave) {
            const int d = dbase + lane;
            float db[4] = {0.f, 0.f, 0.f, 0.f};
            if (d < N) dets.get(d, db);
            for (int tbase = 0; tbase < T; tbase += kWave) {
                const int tl = tbase + lane;
                double tbx[4] = {0., 0., 0., 0.};
                if (tl < T) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) tbx[q] = M.pbox[q * cap + tl];
                }
                const int tcnt = (T - tbase) < kWave ? (T - tbase) : kWave;
                for (int tt = 0; tt < tcnt; ++tt) {
                    double tb[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned long long bits = readlane64((unsigned long long)__double_as_longlong(tbx[q]), tt);
                        tb[q] = __longlong_as_double((long long)bits);
                    }
                    const float v = (float)iou_det_trk(db, tb);      // stored float32 (sort.py:201,205)
                    if (d < N) {
                        const int t = tbase + tt;
                        const int r = transposed ? t : d, c = transposed ? d : t;
                        C[r * ld + c] = -v;                            // linear_assignment(-iou_matrix)
                    }
                }
            }
        }
        wsync();
        WT_TICK(1)
        MunkresMem L2 = L;
        L2.cost_in_lds = in_lds ? 1 : 0;
        const int rc = in_lds ? munkres_wave<HELP>(lds_cost, n, m, ld, L2) : munkres_wave<HELP>(M.cost_g, n, m, ld, L2);
        if (rc) return rc;
        WT_TICK(2)
        for (int d = lane; d < N; d += kWave) {
            const int t = transposed ? L.col_star[d] : L.row_star[d];
            if (t >= 0) {
                float db[4];
                dets.get(d, db);
                double tb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) tb[q] = M.pbox[q * cap + t];
                const float v = (float)iou_det_trk(db, tb);
                if ((double)v < iou_thr) {                             // sort.py:220 (f32 value vs python float)
                    M.det_match[d] = -2;
                } else {
                    M.det_match[d] = t;
                    M.trk_match[t] = d;
                }
            }
        }
        wsync();
    }
    WT_TICK(3)
    // ---- update matched tracks (sort.py:270-273) ----
    for (int i = lane; i < T; i += kWave) {
        const int d = M.trk_match[i];
        if (d >= 0) {
            const int slot = M.order[i];
            float db[4];
            dets.get(d, db);
            M.tsu[slot] = 0;
            M.streak[slot] = M.streak[slot] + 1;
            kalman_update(M, slot, db);
        }
    }
    WT_TICK(4)
    // ---- births (sort.py:276-278): never-assigned detections ascending, then threshold-rejected ones ----
    int nb = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int want = pass == 0 ? -1 : -2;
        for (int base = 0; base < N; base += kWave) {
            const int k = base + lane;
            const bool f = (k < N) && (M.det_match[k] == want);
            const unsigned long long mask = __ballot(f);
            if (f) M.new_list[nb + __popcll(mask & lt)] = k;
            nb += __popcll(mask);
        }
    }
    if (T + nb > cap || nb > S.n_free) return kErrCapacity;
    wsync();
    for (int k = lane; k < nb; k += kWave) {
        const int slot = M.freel[S.n_free - 1 - k];
        M.order[T + k] = slot;
        float db[4];
        dets.get(M.new_list[k], db);
        track_init(M, slot, db);
        M.gid[slot] = id_base + S.next_local + k;
        M.tsu[slot] = 0;
        M.streak[slot] = 0;
        M.bframe[slot] = frame_global;
        M.bk[slot] = k;
    }
    S.n_free -= nb;
    S.next_local += nb;
    const int T2 = T + nb;
    wsync();
    // ---- emit newest first (sort.py:279-290) ----
    int K = 0;
    for (int top = T2; top > 0; top -= kWave) {
        const int i = top - 1 - lane;
        bool e = false;
        double b[4] = {0., 0., 0., 0.}, conf = 0.;
        int slot = 0;
        if (i >= 0) {
            slot = M.order[i];
            if (M.tsu[slot] < 1 && (M.streak[slot] >= min_hits || S.frame_count <= min_hits)) {
                x_to_bbox(M.kx[0 * cap + slot], M.kx[1 * cap + slot], M.kx[2 * cap + slot], M.kx[3 * cap + slot], b);
                const double err = ((M.kP[0 * cap + slot] + M.kP[8 * cap + slot]) + M.kP[16 * cap + slot]) / 3.0;
                conf = exp(-err * 0.1);
                e = emit.accept(b, conf);
            }
        }
        const unsigned long long mask = __ballot(e);
        if (e) emit.write(K + __popcll(mask & lt), b, conf, M.gid[slot], M.bframe[slot], M.bk[slot]);
        K += __popcll(mask);
    }
    // ---- reap (sort.py:291-293) ----
    int Tn = 0;
    for (int base = 0; base < T2; base += kWave) {
        const int i = base + lane;
        const bool act = i < T2;
        int slot = 0;
        bool live = false;
        if (act) { slot = M.order[i]; live = !(M.tsu[slot] > max_age); }
        const unsigned long long good = __ballot(act && live);
        const unsigned long long bad = __ballot(act && !live);
        if (act && live) M.order[Tn + __popcll(good & lt)] = slot;
        if (act && !live) M.freel[S.n_free + __popcll(bad & lt)] = slot;
        Tn += __popcll(good);
        S.n_free += __popcll(bad);
    }
    S.n_tracks = Tn;
    *n_births = nb;
    *n_rows = K;
    wsync();
    WT_TICK(5)
    return 0;
}


}  // namespace wtdev
