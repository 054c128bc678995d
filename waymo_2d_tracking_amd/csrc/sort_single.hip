// Per-object SORT API on gfx950 (include/waymotrack.h): wt_sort_* (tracking/sort/sort.py:233-296 Sort),
// wt_associate_host (sort.py:193-230) and wt_linear_assignment_f32_host (scikit-learn 0.22.2 Munkres, call site
// sort.py:206).  Same device code as the batched engine (sort_device.h); the tracker state stays resident in
// device memory between calls, each call is one single-wavefront kernel launch.
#include "common.h"
#include "sort_device.h"
#include <vector>

using namespace wtdev;

namespace {

constexpr int kLdsCostFloats = 8192;
constexpr size_t kLdsMunkresMax = 120 * 1024;   // stars / primes / zero bitmaps (dynamic LDS, raised limit)

struct RowDets {
    const float* p;
    __device__ __forceinline__ void get(int k, float o[4]) const {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = p[5 * k + q];
    }
};

struct RowEmit {
    double* out;
    int cap;
    __device__ __forceinline__ bool accept(double*, double&) const { return true; }
    __device__ __forceinline__ void write(int j, const double b[4], double conf, long long gid, int, int) const {
        if (j >= cap) return;
        double* o = out + 6 * (size_t)j;
        o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3];
        o[4] = (double)(gid + 1);                                  // sort.py:288
        o[5] = conf;
    }
};

// state: [n_tracks, n_free, frame_count, next_local]; result: [rc, n_rows, n_births, n_tracks]
__global__ __launch_bounds__(kWave) void sort_step_kernel(TrackerMem M, int* state, const float* dets5, int N,
                                                          double iou_thr, int max_age, int min_hits, long long id_base,
                                                          double* out6, int out_cap, int* result, int lds_cost_cap,
                                                          int mk_small, int mk_big) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds_cost = reinterpret_cast<float*>(smem);
    MunkresMem L = munkres_mem(smem + (((size_t)lds_cost_cap * sizeof(float) + 15) / 16) * 16, mk_small, mk_big);
    TrackerState S = {state[0], state[1], state[2], state[3]};
    __syncthreads();
    RowDets dets = {dets5};
    RowEmit emit = {out6, out_cap};
    int nb = 0, nr = 0;
    S.next_local = 0;                    // ids come from id_base (+ rank of the birth inside this call)
    const int rc = tracker_step(M, S, L, lds_cost, lds_cost_cap, dets, N, iou_thr, max_age, min_hits, 0, id_base, emit,
                                &nb, &nr);
    if (threadIdx.x == 0) {
        state[0] = S.n_tracks; state[1] = S.n_free; state[2] = S.frame_count; state[3] = 0;
        result[0] = (rc == 0 && nr > out_cap) ? kErrCapacity : rc;
        result[1] = nr; result[2] = nb; result[3] = S.n_tracks;
    }
}

__global__ void sort_init_kernel(TrackerMem M, int* state) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M.cap; i += gridDim.x * blockDim.x) M.freel[i] = M.cap - 1 - i;
    if (blockIdx.x == 0 && threadIdx.x == 0) { state[0] = 0; state[1] = M.cap; state[2] = 0; state[3] = 0; }
}

// grow: copy the live tracks (list order) into slots 0..n-1 of a larger block
__global__ void sort_migrate_kernel(TrackerMem A, TrackerMem B, int* state) {
    const int n = state[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B.cap; i += gridDim.x * blockDim.x) {
        if (i < n) {
            const int s = A.order[i];
            for (int e = 0; e < 7; ++e) B.kx[e * B.cap + i] = A.kx[e * A.cap + s];
            for (int e = 0; e < 49; ++e) B.kP[e * B.cap + i] = A.kP[e * A.cap + s];
            B.gid[i] = A.gid[s]; B.tsu[i] = A.tsu[s]; B.streak[i] = A.streak[s];
            B.bframe[i] = A.bframe[s]; B.bk[i] = A.bk[s];
            B.order[i] = i;
        }
        if (i < B.cap - n) B.freel[i] = B.cap - 1 - i;
    }
}

__global__ void sort_migrate_finish_kernel(int* state, int new_cap) { state[1] = new_cap - state[0]; }

__global__ void sort_state_kernel(TrackerMem M, const int* state, long long* ids, double* x7, double* P49, int cap_out) {
    const int n = state[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n && i < cap_out; i += gridDim.x * blockDim.x) {
        const int s = M.order[i];
        ids[i] = M.gid[s];
        for (int e = 0; e < 7; ++e) x7[7 * (size_t)i + e] = M.kx[e * M.cap + s];
        for (int e = 0; e < 49; ++e) P49[49 * (size_t)i + e] = M.kP[e * M.cap + s];
    }
}

// Munkres on an arbitrary (n_rows x n_cols) float32 matrix: linear_assignment(X) incl. the transposition rule
__global__ __launch_bounds__(kWave) void assignment_kernel(const float* __restrict__ cost, int n_rows, int n_cols,
                                                           float* work, int* pairs, int* result, bool work_in_lds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const bool transposed = n_cols < n_rows;
    const int n = transposed ? n_cols : n_rows, m = transposed ? n_rows : n_cols;
    float* lds_cost = reinterpret_cast<float*>(smem);
    const int ld = munkres_ld(m);
    MunkresMem L = munkres_mem(smem + (work_in_lds ? (((size_t)n * ld * sizeof(float) + 15) / 16) * 16 : 0), n, m);
    float* C = work_in_lds ? lds_cost : work;
    for (int e = lane; e < n * m; e += kWave) {
        const int r = e / m, c = e - r * m;
        C[r * ld + c] = transposed ? cost[(size_t)c * n_cols + r] : cost[(size_t)r * n_cols + c];
    }
    __syncthreads();
    const int rc = work_in_lds ? munkres_wave(lds_cost, n, m, ld, L) : munkres_wave(work, n, m, ld, L);
    int k = 0;
    if (rc == 0) {
        const unsigned long long lt = lanemask_lt();
        const int cnt = transposed ? m : n;                 // iterate original rows ascending
        for (int base = 0; base < cnt; base += kWave) {
            const int i = base + lane;
            const int other = (i < cnt) ? (transposed ? L.col_star[i] : L.row_star[i]) : -1;
            const unsigned long long mask = __ballot(other >= 0);
            if (other >= 0) {
                const int p = k + __popcll(mask & lt);
                pairs[2 * p] = i;
                pairs[2 * p + 1] = other;
            }
            k += __popcll(mask);
        }
    }
    if (lane == 0) { result[0] = rc; result[1] = k; }
}

// associate_detections_to_trackers (sort.py:193-230)
__global__ __launch_bounds__(kWave) void associate_kernel(const float* __restrict__ dets5, int N,
                                                          const double* __restrict__ trks4, int T, double thr,
                                                          float* work, bool work_in_lds, int* det_match, int* trk_flag,
                                                          int* matches, int* ud, int* ut, int* result) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const unsigned long long lt = lanemask_lt();
    int nm = 0, nud = 0, nut = 0, rc = 0;
    if (T == 0) {                                                            // sort.py:199-200
        for (int d = lane; d < N; d += kWave) ud[d] = d;
        nud = N;
    } else {
        const bool transposed = T < N;
        const int n = transposed ? T : N, m = transposed ? N : T;
        float* lds_cost = reinterpret_cast<float*>(smem);
        const int ld = munkres_ld(m);
        MunkresMem L = munkres_mem(smem + (work_in_lds ? (((size_t)n * ld * sizeof(float) + 15) / 16) * 16 : 0), n > 0 ? n : 1, m);
        float* C = work_in_lds ? lds_cost : work;
        for (int r = 0; r < n; ++r)
            for (int c = lane; c < m; c += kWave) {
                const int d = transposed ? c : r, t = transposed ? r : c;
                float db[4]; double tb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { db[q] = dets5[5 * d + q]; tb[q] = trks4[4 * t + q]; }
                C[r * ld + c] = -(float)iou_det_trk(db, tb);
            }
        for (int t = lane; t < T; t += kWave) trk_flag[t] = 0;              // 0 never assigned, 1 matched, 2 rejected
        __syncthreads();
        if (N > 0) rc = work_in_lds ? munkres_wave(lds_cost, n, m, ld, L) : munkres_wave(work, n, m, ld, L);
        if (rc == 0) {
            for (int d = lane; d < N; d += kWave) {
                const int t = transposed ? L.col_star[d] : L.row_star[d];
                int v = -1;
                if (t >= 0) {
                    float db[4]; double tb[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { db[q] = dets5[5 * d + q]; tb[q] = trks4[4 * t + q]; }
                    const float io = (float)iou_det_trk(db, tb);
                    if ((double)io < thr) { v = -2 - t; trk_flag[t] = 2; } else { v = t; trk_flag[t] = 1; }
                }
                det_match[d] = v;
            }
            __syncthreads();
            for (int base = 0; base < N; base += kWave) {                   // sort.py:208-211
                const int d = base + lane;
                const bool f = d < N && det_match[d] == -1;
                const unsigned long long mask = __ballot(f);
                if (f) ud[nud + __popcll(mask & lt)] = d;
                nud += __popcll(mask);
            }
            for (int base = 0; base < T; base += kWave) {                   // sort.py:212-215
                const int t = base + lane;
                const bool f = t < T && trk_flag[t] == 0;
                const unsigned long long mask = __ballot(f);
                if (f) ut[nut + __popcll(mask & lt)] = t;
                nut += __popcll(mask);
            }
            for (int base = 0; base < N; base += kWave) {                   // sort.py:218-224, pairs sorted by det
                const int d = base + lane;
                const int v = d < N ? det_match[d] : -1;
                const bool rej = v <= -2, acc = v >= 0;
                const unsigned long long mr = __ballot(rej), ma = __ballot(acc);
                if (rej) { const int p = __popcll(mr & lt); ud[nud + p] = d; ut[nut + p] = -2 - v; }
                if (acc) { const int p = nm + __popcll(ma & lt); matches[2 * p] = d; matches[2 * p + 1] = v; }
                nud += __popcll(mr); nut += __popcll(mr); nm += __popcll(ma);
            }
        }
    }
    if (lane == 0) { result[0] = rc; result[1] = nm; result[2] = nud; result[3] = nut; }
}

int alloc_tracker(wt::DevBuf& buf, int cap, TrackerMem* M) {
    wt::Carver c0(nullptr);
    auto carve = [&](wt::Carver& cv) {
        TrackerMem m;
        m.kx = cv.take<double>((size_t)7 * cap);
        m.kP = cv.take<double>((size_t)49 * cap);
        m.pbox = cv.take<double>((size_t)4 * cap);
        m.gid = cv.take<long long>((size_t)cap);
        m.tsu = cv.take<int>((size_t)cap);
        m.streak = cv.take<int>((size_t)cap);
        m.bframe = cv.take<int>((size_t)cap);
        m.bk = cv.take<int>((size_t)cap);
        m.order = cv.take<int>((size_t)cap);
        m.freel = cv.take<int>((size_t)cap);
        m.trk_match = cv.take<int>((size_t)cap);
        m.det_match = nullptr; m.new_list = nullptr; m.cost_g = nullptr;
        m.cap = cap; m.capN = 0;
        return m;
    };
    (void)carve(c0);
    WT_TRY(buf.alloc(c0.off));
    wt::Carver c1(buf.p);
    *M = carve(c1);
    return WT_OK;
}

}  // namespace

struct wt_sort {
    int max_age, min_hits;
    wt_idctr* ctr;
    wt_idctr* own;
    wt::DevBuf mem, state, dets, dmatch, newl, out, result, cost;
    TrackerMem M;
    int n_tracks = 0;
    int det_cap = 0, out_cap = 0;
    size_t cost_cap = 0;
};

extern "C" {

int wt_sort_create(int max_age, int min_hits, wt_idctr* ctr, wt_sort** out) {
    if (!out) return WT_ERR_INVALID;
    *out = nullptr;
    WT_TRY(wt::ensure_device());
    wt_sort* s = new wt_sort;
    s->max_age = max_age; s->min_hits = min_hits;
    s->own = ctr ? nullptr : wt_idctr_create(0);
    s->ctr = ctr ? ctr : s->own;
    int rc = alloc_tracker(s->mem, 256, &s->M);
    if (rc == WT_OK) rc = s->state.alloc(sizeof(int) * 4);
    if (rc == WT_OK) rc = s->result.alloc(sizeof(int) * 4);
    if (rc != WT_OK) { wt_sort_destroy(s); return rc; }
    hipLaunchKernelGGL(sort_init_kernel, dim3(4), dim3(256), 0, nullptr, s->M, s->state.as<int>());
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { wt_sort_destroy(s); return wt::hip_fail(e, "sort_init_kernel"); }
    *out = s;
    return WT_OK;
}

void wt_sort_destroy(wt_sort* s) {
    if (!s) return;
    if (s->own) wt_idctr_destroy(s->own);
    delete s;
}

int wt_sort_update_host(wt_sort* s, const float* dets5, int n, double iou_threshold, double* out6, int cap, int* n_out) {
    if (!s || n < 0 || (n > 0 && !dets5) || !n_out) { wt::set_error("wt_sort_update_host: bad argument"); return WT_ERR_INVALID; }
    WT_TRY(wt::ensure_device());
    *n_out = 0;
    // capacity: every live track plus one new track per detection
    if (s->n_tracks + n > s->M.cap) {
        int new_cap = s->M.cap;
        while (s->n_tracks + n > new_cap) new_cap *= 2;
        wt::DevBuf nb;
        TrackerMem B;
        WT_TRY(alloc_tracker(nb, new_cap, &B));
        hipLaunchKernelGGL(sort_migrate_kernel, dim3(8), dim3(256), 0, nullptr, s->M, B, s->state.as<int>());
        hipLaunchKernelGGL(sort_migrate_finish_kernel, dim3(1), dim3(1), 0, nullptr, s->state.as<int>(), new_cap);
        WT_HIP(hipDeviceSynchronize());
        std::swap(s->mem.p, nb.p);
        std::swap(s->mem.bytes, nb.bytes);
        s->M = B;
    }
    if (n > s->det_cap) {
        const int c = n < 64 ? 64 : 2 * n;
        WT_TRY(s->dets.alloc(sizeof(float) * 5 * (size_t)c));
        WT_TRY(s->dmatch.alloc(sizeof(int) * (size_t)c));
        WT_TRY(s->newl.alloc(sizeof(int) * (size_t)c));
        s->det_cap = c;
    }
    const int rows_cap = s->n_tracks + n + 1;
    if (rows_cap > s->out_cap) { WT_TRY(s->out.alloc(sizeof(double) * 6 * (size_t)rows_cap * 2)); s->out_cap = rows_cap * 2; }
    if (n) WT_HIP(hipMemcpy(s->dets.p, dets5, sizeof(float) * 5 * (size_t)n, hipMemcpyHostToDevice));
    const int T = s->n_tracks;
    const int small = n < T ? n : T, big = n < T ? T : n;
    const size_t mk = munkres_lds_bytes(small > 0 ? small : 1, big > 0 ? big : 1);
    if (mk > kLdsMunkresMax) { wt::set_error("assignment of %d x %d exceeds the LDS budget", n, T); return WT_ERR_CAPACITY; }
    const size_t elems = (size_t)(small > 0 ? small : 1) * (size_t)((big > 0 ? big : 1) | 1);   // n x ld
    int lds_cost = (int)(elems <= (size_t)kLdsCostFloats ? elems : 0);
    TrackerMem M = s->M;
    M.det_match = s->dmatch.as<int>();
    M.new_list = s->newl.as<int>();
    M.capN = s->det_cap;
    if (elems > (size_t)kLdsCostFloats) {
        if (elems > s->cost_cap) { WT_TRY(s->cost.alloc(sizeof(float) * elems * 2)); s->cost_cap = elems * 2; }
        M.cost_g = s->cost.as<float>();
    }
    const long long id_base = wt_idctr_get(s->ctr);
    {
        const size_t lds = wt::align_up((size_t)lds_cost * sizeof(float), 16) + mk;
        if (lds > 48 * 1024)
            WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sort_step_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(sort_step_kernel, dim3(1), dim3(kWave), wt::align_up((size_t)lds_cost * sizeof(float), 16) + mk, nullptr, M,
                       s->state.as<int>(), s->dets.as<float>(), n, iou_threshold, s->max_age, s->min_hits, id_base,
                       s->out.as<double>(), s->out_cap, s->result.as<int>(), lds_cost, small > 0 ? small : 1,
                       big > 0 ? big : 1);
    WT_HIP(hipGetLastError());
    WT_HIP(hipDeviceSynchronize());
    int res[4];
    WT_HIP(hipMemcpy(res, s->result.p, sizeof(res), hipMemcpyDeviceToHost));
    if (res[0] != 0) { wt::set_error("SORT step kernel status %d", res[0]); return res[0]; }
    s->n_tracks = res[3];
    wt_idctr_set(s->ctr, id_base + res[2]);
    *n_out = res[1];
    if (res[1] > cap) { wt::set_error("output buffer holds %d rows, %d needed", cap, res[1]); return WT_ERR_CAPACITY; }
    if (res[1]) WT_HIP(hipMemcpy(out6, s->out.p, sizeof(double) * 6 * (size_t)res[1], hipMemcpyDeviceToHost));
    return WT_OK;
}

int wt_sort_num_tracks(const wt_sort* s) { return s ? s->n_tracks : 0; }

int wt_sort_state_host(wt_sort* s, int cap, int64_t* ids, double* x7, double* P49, int* n_tracks) {
    if (!s || !n_tracks) return WT_ERR_INVALID;
    WT_TRY(wt::ensure_device());
    *n_tracks = s->n_tracks;
    if (s->n_tracks > cap) { wt::set_error("state buffers hold %d tracks, %d live", cap, s->n_tracks); return WT_ERR_CAPACITY; }
    if (s->n_tracks == 0) return WT_OK;
    const size_t n = (size_t)s->n_tracks;
    wt::DevBuf di, dx, dp;
    WT_TRY(di.alloc(8 * n)); WT_TRY(dx.alloc(56 * n)); WT_TRY(dp.alloc(392 * n));
    hipLaunchKernelGGL(sort_state_kernel, dim3(4), dim3(256), 0, nullptr, s->M, s->state.as<int>(), di.as<long long>(),
                       dx.as<double>(), dp.as<double>(), cap);
    WT_HIP(hipDeviceSynchronize());
    WT_HIP(hipMemcpy(ids, di.p, 8 * n, hipMemcpyDeviceToHost));
    WT_HIP(hipMemcpy(x7, dx.p, 56 * n, hipMemcpyDeviceToHost));
    WT_HIP(hipMemcpy(P49, dp.p, 392 * n, hipMemcpyDeviceToHost));
    return WT_OK;
}

int wt_linear_assignment_f32_host(const float* cost, int n_rows, int n_cols, int* pairs, int* n_pairs) {
    if (!n_pairs) return WT_ERR_INVALID;
    *n_pairs = 0;
    WT_TRY(wt::ensure_device());
    if (n_rows <= 0 || n_cols <= 0) return WT_OK;
    const int small = n_rows < n_cols ? n_rows : n_cols, big = n_rows < n_cols ? n_cols : n_rows;
    const size_t elems = (size_t)small * (size_t)(big | 1);      // working matrix: n x ld
    const size_t mk = munkres_lds_bytes(small, big);
    if (mk > kLdsMunkresMax) { wt::set_error("assignment of %d x %d exceeds the LDS budget", n_rows, n_cols); return WT_ERR_CAPACITY; }
    const bool in_lds = elems <= (size_t)kLdsCostFloats;
    wt::DevBuf dc, dw, dp, dr;
    const size_t in_elems = (size_t)n_rows * (size_t)n_cols;
    {
        const size_t lds = (in_lds ? wt::align_up(4 * elems, 16) : 0) + mk;
        if (lds > 48 * 1024)
            WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assignment_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    WT_TRY(dc.alloc(4 * in_elems)); WT_TRY(dw.alloc(in_lds ? 16 : 4 * elems)); WT_TRY(dp.alloc(8 * (size_t)small)); WT_TRY(dr.alloc(16));
    WT_HIP(hipMemcpy(dc.p, cost, 4 * in_elems, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(assignment_kernel, dim3(1), dim3(kWave), (in_lds ? wt::align_up(4 * elems, 16) : 0) + mk, nullptr, dc.as<float>(),
                       n_rows, n_cols, dw.as<float>(), dp.as<int>(), dr.as<int>(), in_lds);
    WT_HIP(hipGetLastError());
    WT_HIP(hipDeviceSynchronize());
    int res[2];
    WT_HIP(hipMemcpy(res, dr.p, sizeof(res), hipMemcpyDeviceToHost));
    if (res[0]) { wt::set_error("assignment kernel status %d", res[0]); return res[0]; }
    *n_pairs = res[1];
    if (res[1]) WT_HIP(hipMemcpy(pairs, dp.p, 8 * (size_t)res[1], hipMemcpyDeviceToHost));
    return WT_OK;
}

int wt_associate_host(const float* dets5, int n, const double* trks4, int t, double iou_threshold,
                      int* matches, int* n_matches, int* unmatched_dets, int* n_unmatched_dets,
                      int* unmatched_trks, int* n_unmatched_trks) {
    if (!n_matches || !n_unmatched_dets || !n_unmatched_trks || n < 0 || t < 0) return WT_ERR_INVALID;
    *n_matches = 0; *n_unmatched_dets = 0; *n_unmatched_trks = 0;
    WT_TRY(wt::ensure_device());
    if (n == 0 && t == 0) return WT_OK;
    const int small = n < t ? n : t, big = n < t ? t : n;
    const size_t elems = (size_t)(small > 0 ? small : 1) * (size_t)((big > 0 ? big : 1) | 1);
    const size_t mk = munkres_lds_bytes(small > 0 ? small : 1, big > 0 ? big : 1);
    if (mk > kLdsMunkresMax) { wt::set_error("assignment of %d x %d exceeds the LDS budget", n, t); return WT_ERR_CAPACITY; }
    const bool in_lds = elems <= (size_t)kLdsCostFloats;
    wt::DevBuf dd, dt, dw, dm, df, dma, dud, dut, dr;
    WT_TRY(dd.alloc(20 * (size_t)n)); WT_TRY(dt.alloc(32 * (size_t)t)); WT_TRY(dw.alloc(in_lds ? 16 : 4 * elems));
    WT_TRY(dm.alloc(4 * (size_t)n)); WT_TRY(df.alloc(4 * (size_t)t)); WT_TRY(dma.alloc(8 * (size_t)(small + 1)));
    WT_TRY(dud.alloc(4 * (size_t)(n + 1))); WT_TRY(dut.alloc(4 * (size_t)(t + 1))); WT_TRY(dr.alloc(16));
    {
        const size_t lds = (in_lds ? wt::align_up(4 * elems, 16) : 0) + mk;
        if (lds > 48 * 1024)
            WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(associate_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (n) WT_HIP(hipMemcpy(dd.p, dets5, 20 * (size_t)n, hipMemcpyHostToDevice));
    if (t) WT_HIP(hipMemcpy(dt.p, trks4, 32 * (size_t)t, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(associate_kernel, dim3(1), dim3(kWave), (in_lds ? wt::align_up(4 * elems, 16) : 0) + mk, nullptr, dd.as<float>(), n,
                       dt.as<double>(), t, iou_threshold, dw.as<float>(), in_lds, dm.as<int>(), df.as<int>(),
                       dma.as<int>(), dud.as<int>(), dut.as<int>(), dr.as<int>());
    WT_HIP(hipGetLastError());
    WT_HIP(hipDeviceSynchronize());
    int res[4];
    WT_HIP(hipMemcpy(res, dr.p, sizeof(res), hipMemcpyDeviceToHost));
    if (res[0]) { wt::set_error("associate kernel status %d", res[0]); return res[0]; }
    *n_matches = res[1]; *n_unmatched_dets = res[2]; *n_unmatched_trks = res[3];
    if (res[1]) WT_HIP(hipMemcpy(matches, dma.p, 8 * (size_t)res[1], hipMemcpyDeviceToHost));
    if (res[2]) WT_HIP(hipMemcpy(unmatched_dets, dud.p, 4 * (size_t)res[2], hipMemcpyDeviceToHost));
    if (res[3]) WT_HIP(hipMemcpy(unmatched_trks, dut.p, 4 * (size_t)res[3], hipMemcpyDeviceToHost));
    return WT_OK;
}

}  // extern "C"

// ---- MultiClassTrackerSort (tracking/sort/tracker_sort.py:10-51): one Sort per class, created at first sight ----

struct wt_mct {
    int max_age = 1, min_hits = 0;
    wt_idctr* ctr = nullptr;
    wt_idctr* own = nullptr;
    std::vector<int> classes;          // first-seen order == dict order of self.trackers
    std::vector<wt_sort*> sorts;
    std::vector<float> rows;
};

extern "C" {

int wt_mct_create(int max_age, int min_hits, wt_idctr* ctr, wt_mct** out) {
    if (!out) return WT_ERR_INVALID;
    *out = nullptr;
    WT_TRY(wt::ensure_device());
    wt_mct* m = new wt_mct;
    m->max_age = max_age; m->min_hits = min_hits;
    m->own = ctr ? nullptr : wt_idctr_create(0);
    m->ctr = ctr ? ctr : m->own;
    *out = m;
    return WT_OK;
}

void wt_mct_destroy(wt_mct* m) {
    if (!m) return;
    for (wt_sort* s : m->sorts) wt_sort_destroy(s);
    if (m->own) wt_idctr_destroy(m->own);
    delete m;
}

int wt_mct_num_classes(const wt_mct* m) { return m ? (int)m->classes.size() : 0; }

wt_sort* wt_mct_tracker(wt_mct* m, int class_id) {
    if (!m) return nullptr;
    for (size_t k = 0; k < m->classes.size(); ++k)
        if (m->classes[k] == class_id) return m->sorts[k];
    return nullptr;
}

int wt_mct_track_host(wt_mct* m, const double* dets6, int n, const double* iou_thresholds, int n_thresholds,
                      double* out6, int cap, int32_t* out_classes, int32_t* out_counts, int class_cap, int* n_classes) {
    if (!m || n < 0 || (n > 0 && !dets6) || !iou_thresholds || !n_classes || (cap > 0 && !out6)) {
        wt::set_error("wt_mct_track_host: bad argument");
        return WT_ERR_INVALID;
    }
    *n_classes = 0;
    // tracker_sort.py:28-35: bucket by class; a new class gets its Sort the first time it is seen (detection order)
    for (int i = 0; i < n; ++i) {
        const int cls = (int)dets6[(size_t)i * 6 + 5];
        if (cls < 1 || cls > n_thresholds) {
            wt::set_error("wt_mct_track_host: class %d has no iou threshold (iou_thresholds[class-1], %d given)", cls, n_thresholds);
            return WT_ERR_INVALID;
        }
        bool known = false;
        for (int c : m->classes) known |= c == cls;
        if (!known) {
            wt_sort* s = nullptr;
            WT_TRY(wt_sort_create(m->max_age, m->min_hits, m->ctr, &s));
            m->classes.push_back(cls);
            m->sorts.push_back(s);
        }
    }
    if ((int)m->classes.size() > class_cap || !out_classes || !out_counts) {
        wt::set_error("wt_mct_track_host: %d classes do not fit class_cap %d", (int)m->classes.size(), class_cap);
        return WT_ERR_CAPACITY;
    }
    // :41-49: every known class, in first-seen order, is updated once per frame (empty array when it has no detections)
    int written = 0;
    for (size_t k = 0; k < m->classes.size(); ++k) {
        const int cls = m->classes[k];
        m->rows.clear();
        for (int i = 0; i < n; ++i) {
            if ((int)dets6[(size_t)i * 6 + 5] != cls) continue;
            for (int j = 0; j < 5; ++j) m->rows.push_back((float)dets6[(size_t)i * 6 + j]);   // np.array(..., dtype=np.float32)
        }
        int k_out = 0;
        WT_TRY(wt_sort_update_host(m->sorts[k], m->rows.data(), (int)(m->rows.size() / 5), iou_thresholds[cls - 1],
                                   out6 + (size_t)written * 6, cap - written, &k_out));
        out_classes[k] = cls;
        out_counts[k] = k_out;
        written += k_out;
    }
    *n_classes = (int)m->classes.size();
    return WT_OK;
}

}  // extern "C"
