// SORT engine on gfx950 (include/waymotrack.h, "SORT" section).
//
// Persistent kernel: grid = n_streams * n_classes wavefronts; wave (s, c) is the reference's
// `Sort` object of class c inside the MultiClassTrackerSort of stream s (tracking/sort/tracker_sort.py:22-51,
// tracking/utils.py:25-60) and iterates that stream's frames in-kernel.  Class trackers of a stream are
// independent except for two global orders, both recovered afterwards without any serialisation:
//   * track IDs  = running count of births in (stream, frame, class-first-seen, birth) order (sort.py:86,140),
//   * output row order = (stream, frame, class-first-seen, newest track first)        (utils.py:31-58).
// Each wave records births / emitted rows per (frame, class); `rank_classes_kernel` orders the classes of a
// stream by first appearance, `scan_kernel` turns the per-(frame, class-rank) counts into offsets and
// `finalize_kernel` scatters rows to their final position with their global id.
#include "common.h"
#include "sort_device.h"
#include <vector>

using namespace wtdev;

namespace {

constexpr int kLdsCostFloats = 8192;          // 32 KiB of cost matrix in LDS per wave (thousands of trackers: 4 workgroups per CU)
constexpr int kLdsCostFloatsFew = 36864;      // 144 KiB when every tracker can own a CU anyway (<= 256 trackers: the online pipeline's 5-20):
                                              // a 100 x 300 cost matrix then stays in LDS instead of global memory
constexpr size_t kLdsMunkresMax = 120 * 1024;   // stars / primes / zero bitmaps (dynamic LDS, raised limit)  // stars/primes/covers

// Persistent part of a set of trackers: survives between the chunks of the streaming entry points
// (wt_track_state_* / wt_track_chunk_dev); the one-shot entry point carves it from its workspace.
struct State {
    // per tracker
    double* kx; double* kP;
    long long* gid;                      // per track slot: birth ordinal inside its stream once resolved (streaming form)
    int *tsu, *streak, *bframe, *bk, *order, *freel;
    int* hdr;                            // [n_trackers][4] TrackerState between chunks
    long long* first_key;                // per tracker: (stream frame number << 32 | det position) of creation, or -1
    // per stream
    long long* s_frames;                 // frames consumed so far
    long long* s_births;                 // track ids consumed so far
    size_t bytes;
};

struct Workspace {                       // device pointers carved from one block (transient: one call)
    // per tracker
    double* pbox;
    int *trk_match, *det_match, *new_list, *det_idx;
    float* cost_g;
    // per (frame, class)
    int* cnt;                            // emitted rows
    int* births;                         // births
    long long* ibase;                    // first intermediate slot of the (frame, class) block
    // per intermediate slot
    double* irow;                        // [5][n_dets]: x1, y1, w, h, score
    int *ifr, *isc, *ij, *ibf, *ibk;
    long long* igid;                     // resolved birth ordinal of the emitting track (streaming form, ibf < 0)
    // ranked
    int* rank;                           // [n_streams * C]: class rank inside its stream or -1
    long long* rcnt;                     // [n_frames * C] ranked row counts -> exclusive scan
    long long* rbirths;                  // [n_frames * C] ranked births     -> exclusive scan
    double* thr_score; double* thr_iou;  // [C]
    int* err;
    long long* totals;                   // [2] rows, births
    size_t bytes;
};

State carve_state(void* base, int32_t n_streams, int C, int cap) {
    wt::Carver cv(base);
    State st;
    const size_t nt = (size_t)n_streams * (size_t)C;
    st.kx = cv.take<double>(nt * 7 * cap);
    st.kP = cv.take<double>(nt * 49 * cap);
    st.gid = cv.take<long long>(nt * cap);
    st.tsu = cv.take<int>(nt * cap);
    st.streak = cv.take<int>(nt * cap);
    st.bframe = cv.take<int>(nt * cap);
    st.bk = cv.take<int>(nt * cap);
    st.order = cv.take<int>(nt * cap);
    st.freel = cv.take<int>(nt * cap);
    st.hdr = cv.take<int>(nt * 4);
    st.first_key = cv.take<long long>(nt);
    st.s_frames = cv.take<long long>((size_t)n_streams);
    st.s_births = cv.take<long long>((size_t)n_streams);
    st.bytes = cv.off;
    return st;
}

Workspace carve_ws(void* base, int64_t n_dets, int64_t n_frames, int32_t n_streams, int C, int cap, int capN,
                   bool need_cost_g) {
    wt::Carver cv(base);
    Workspace w;
    const size_t nt = (size_t)n_streams * (size_t)C;
    w.pbox = cv.take<double>(nt * 4 * cap);
    w.trk_match = cv.take<int>(nt * cap);
    w.det_match = cv.take<int>(nt * capN);
    w.new_list = cv.take<int>(nt * capN);
    w.det_idx = cv.take<int>(nt * capN);
    w.cost_g = cv.take<float>(need_cost_g ? nt * (size_t)capN * (cap | 1) : 1);
    w.cnt = cv.take<int>((size_t)n_frames * C);
    w.births = cv.take<int>((size_t)n_frames * C);
    w.ibase = cv.take<long long>((size_t)n_frames * C);
    w.irow = cv.take<double>((size_t)n_dets * 5 + 1);
    w.ifr = cv.take<int>((size_t)n_dets + 1);
    w.isc = cv.take<int>((size_t)n_dets + 1);
    w.ij = cv.take<int>((size_t)n_dets + 1);
    w.ibf = cv.take<int>((size_t)n_dets + 1);
    w.ibk = cv.take<int>((size_t)n_dets + 1);
    w.igid = cv.take<long long>((size_t)n_dets + 1);
    w.rank = cv.take<int>(nt);
    w.rcnt = cv.take<long long>((size_t)n_frames * C + 1);
    w.rbirths = cv.take<long long>((size_t)n_frames * C + 1);
    w.thr_score = cv.take<double>((size_t)C);
    w.thr_iou = cv.take<double>((size_t)C);
    w.err = cv.take<int>(4);
    w.totals = cv.take<long long>(2);
    w.bytes = cv.off;
    return w;
}

__device__ __forceinline__ TrackerMem tracker_mem(const State& st, const Workspace& w, size_t tk, int cap, int capN,
                                                  bool cost_g) {
    TrackerMem M;
    M.kx = st.kx + tk * 7 * cap;
    M.kP = st.kP + tk * 49 * cap;
    M.pbox = w.pbox + tk * 4 * cap;
    M.gid = st.gid + tk * cap;
    M.tsu = st.tsu + tk * cap;
    M.streak = st.streak + tk * cap;
    M.bframe = st.bframe + tk * cap;
    M.bk = st.bk + tk * cap;
    M.order = st.order + tk * cap;
    M.freel = st.freel + tk * cap;
    M.trk_match = w.trk_match + tk * cap;
    M.det_match = w.det_match + tk * capN;
    M.new_list = w.new_list + tk * capN;
    M.cost_g = cost_g ? w.cost_g + tk * (size_t)capN * (cap | 1) : nullptr;
    M.cap = cap;
    M.capN = capN;
    return M;
}

// ------------------------------------------------------------------------------------------------
// batched streams
struct StreamDets {       // k-th detection of this class in the current frame -> float32 row (utils.py:33)
    const double *x, *y, *w, *h;
    const int* idx;
    long long d0;
    __device__ __forceinline__ void get(int k, float o[4]) const {
        const long long d = d0 + idx[k];
        const double xx = x[d], yy = y[d];
        o[0] = (float)xx;
        o[1] = (float)yy;
        o[2] = (float)(xx + w[d]);
        o[3] = (float)(yy + h[d]);
    }
};

struct StreamEmit {       // utils.py:38-58 clip / drop / confidence clip, into the intermediate slots
    double cw, ch;
    double* irow;
    int *ifr, *isc, *ij, *ibf, *ibk;
    long long* igid;
    long long n_slots, base;
    int f, sc;
    __device__ __forceinline__ static double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
    __device__ __forceinline__ bool accept(double b[4], double& conf) const {
        if (cw > 0) {
            b[0] = clipd(b[0], 0, cw); b[1] = clipd(b[1], 0, ch);
            b[2] = clipd(b[2], 0, cw); b[3] = clipd(b[3], 0, ch);
            if ((b[2] - b[0]) < 1 || (b[3] - b[1]) < 1) return false;
            conf = clipd(conf, 0.2, 1.0);
        }
        return true;
    }
    __device__ __forceinline__ void write(int j, const double b[4], double conf, long long gid, int bframe, int bk) const {
        const long long p = base + j;
        igid[p] = gid;
        irow[p] = b[0];
        irow[n_slots + p] = b[1];
        irow[2 * n_slots + p] = b[2] - b[0];
        irow[3 * n_slots + p] = b[3] - b[1];
        irow[4 * n_slots + p] = conf;
        ifr[p] = f; isc[p] = sc; ij[p] = j; ibf[p] = bframe; ibk[p] = bk;
    }
};

// blockDim.x == kWave: the tracker alone; blockDim.x == kHelpWaves * kWave: waves 1.. are helper waves (sort_device.h HelpJob), launched when the
// chip has far more CUs than trackers (config 1 at its stated size: 20 trackers)
// (HELP = false is the plain single-wave code without a trace of the helper branches: inlined into the hot loops they cost the 64-segment
//  batch 14 %)
template <bool HELP>
__global__ __launch_bounds__(HELP ? kHelpWaves * kWave : kWave) void sort_streams_kernel(
    const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ w,
    const double* __restrict__ h, const double* __restrict__ score, const int32_t* __restrict__ category,
    const int64_t* __restrict__ frame_det_offsets, const int64_t* __restrict__ stream_frame_offsets,
    const double* __restrict__ clip_w, const double* __restrict__ clip_h, int C, int max_age, int min_hits,
    int cap, int capN, int lds_cost_cap, bool have_cost_g, long long n_slots, Workspace ws, State st, int resume) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = lanemask_lt();
    const size_t tk = blockIdx.x;
    const int s = (int)(tk / C);
    const int c = (int)(tk % C) + 1;
    float* lds_cost = reinterpret_cast<float*>(smem);
    const size_t mk_off = (((size_t)lds_cost_cap * sizeof(float) + 15) / 16) * 16;
    MunkresMem L = munkres_mem(smem + mk_off, capN, cap);
    if constexpr (HELP) {                        // helper waves: the job descriptor lives behind the Munkres state
        HelpJob* J = reinterpret_cast<HelpJob*>(smem + mk_off + ((wtdev::munkres_lds_bytes(capN, cap) + 15) / 16) * 16);
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        if (threadIdx.x == 0) J->cmd = HELP_NOP;
        __syncthreads();
        if (wave > 0) { helper_loop<StreamDets>(J, wave); return; }
        L.help = J;
    }
    TrackerMem M = tracker_mem(st, ws, tk, cap, capN, have_cost_g);
    int* det_idx = ws.det_idx + tk * capN;
    TrackerState S = {0, cap, 0, 0};
    bool created = false;
    long long first_key = -1;
    long long frames_before = 0;                 // frames of this stream consumed by earlier chunks
    if (resume) {                                // streaming form: continue where the previous chunk stopped
        S.n_tracks = st.hdr[tk * 4 + 0]; S.n_free = st.hdr[tk * 4 + 1];
        S.frame_count = st.hdr[tk * 4 + 2]; S.next_local = st.hdr[tk * 4 + 3];
        first_key = st.first_key[tk];
        created = first_key >= 0;
        frames_before = st.s_frames[s];
    } else {
        for (int i = lane; i < cap; i += kWave) M.freel[i] = cap - 1 - i;
    }
    wsync();
    const double thr_iou = ws.thr_iou[c - 1];
    const double cw = clip_w ? clip_w[s] : 0.0, ch = clip_h ? clip_h[s] : 0.0;
    const long long f0 = stream_frame_offsets[s], f1 = stream_frame_offsets[s + 1];
    for (long long f = f0; f < f1; ++f) {
        const long long d0 = frame_det_offsets[f], d1 = frame_det_offsets[f + 1];
        // select this class's detections (utils.py:79,86 filters; tracker_sort.py:29-37 bucketing) and count the
        // valid detections of lower class ids: they own the intermediate slots in front of ours
        int N = 0, lower = 0, first_pos = -1;
        bool overflow = false;
        for (long long base = d0; base < d1; base += kWave) {
            const long long d = base + lane;
            bool mine = false, low = false;
            if (d < d1) {
                const int cat = category[d];
                const bool valid = (cat >= 1 && cat <= C) && !(w[d] < 1) && !(h[d] < 1) && !(score[d] < ws.thr_score[cat - 1]);
                mine = valid && cat == c;
                low = valid && cat < c;
            }
            const unsigned long long mm = __ballot(mine);
            if (mine) {
                const int k = N + __popcll(mm & lt);
                if (k < capN) det_idx[k] = (int)(d - d0);
            }
            if (mm && first_pos < 0) first_pos = (int)(base - d0) + __builtin_ctzll(mm);
            N += __popcll(mm);
            lower += __popcll(__ballot(low));
        }
        if (N > capN) overflow = true;
        if (!created) {
            if (N == 0) continue;
            created = true;
            first_key = ((f - f0 + frames_before) << 32) | (long long)first_pos;
        }
        wsync();
        StreamDets dets = {x, y, w, h, det_idx, d0};
        StreamEmit emit = {cw, ch, ws.irow, ws.ifr, ws.isc, ws.ij, ws.ibf, ws.ibk, ws.igid, n_slots, d0 + lower, (int)f, (int)tk};
        int nb = 0, nr = 0;
        int rc = overflow ? WT_ERR_CAPACITY
                          : tracker_step<HELP>(M, S, L, lds_cost, lds_cost_cap, dets, N, thr_iou, max_age, min_hits, (int)f,
                                         0ll, emit, &nb, &nr);
        if (rc) {
            if (lane == 0) atomicMax(ws.err, rc);
            break;
        }
        if (lane == 0) { ws.cnt[f * C + c - 1] = nr; ws.births[f * C + c - 1] = nb; }
    }
    if (lane == 0) {
        st.first_key[tk] = created ? first_key : -1;
        st.hdr[tk * 4 + 0] = S.n_tracks; st.hdr[tk * 4 + 1] = S.n_free;
        st.hdr[tk * 4 + 2] = S.frame_count; st.hdr[tk * 4 + 3] = S.next_local;
    }
    if constexpr (HELP) {                        // send the helper waves home
        if (lane == 0) L.help->cmd = HELP_EXIT;
        __syncthreads();
    }
}

// class rank inside its stream = order of first appearance (dict insertion order, tracker_sort.py:32-33,41)
__global__ void rank_classes_kernel(int n_streams, int C, Workspace ws, State st) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    for (int c = 0; c < C; ++c) {
        const long long k = st.first_key[(size_t)s * C + c];
        int r = -1;
        if (k >= 0) {
            r = 0;
            for (int o = 0; o < C; ++o) {
                const long long ko = st.first_key[(size_t)s * C + o];
                if (o != c && ko >= 0 && ko < k) ++r;
            }
        }
        ws.rank[(size_t)s * C + c] = r;
    }
}

// ranked per-(frame, class-rank) counts
__global__ void gather_ranked_kernel(long long n_frames, int n_streams, int C,
                                     const int64_t* __restrict__ stream_frame_offsets, Workspace ws) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_frames * C) return;
    const long long f = e / C;
    const int c = (int)(e - f * C);
    int lo = 0, hi = n_streams;                    // stream of frame f: last s with offsets[s] <= f
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (stream_frame_offsets[mid] <= f) lo = mid; else hi = mid;
    }
    const int r = ws.rank[(size_t)lo * C + c];
    if (r >= 0) {
        ws.rcnt[f * C + r] = ws.cnt[e];
        ws.rbirths[f * C + r] = ws.births[e];
    }
}

// single-block exclusive scan of two int64 arrays (n <= a few million: launch-latency sized)
__global__ __launch_bounds__(1024) void scan2_kernel(long long n, long long* a, long long* b, long long* totals) {
    __shared__ long long sa[1024], sb[1024];
    const int t = threadIdx.x;
    const long long chunk = (n + 1023) / 1024;
    const long long lo = t * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    long long xa = 0, xb = 0;
    for (long long i = lo; i < hi; ++i) { xa += a[i]; xb += b[i]; }
    sa[t] = xa; sb[t] = xb;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        long long va = 0, vb = 0;
        if (t >= o) { va = sa[t - o]; vb = sb[t - o]; }
        __syncthreads();
        sa[t] += va; sb[t] += vb;
        __syncthreads();
    }
    long long ra = sa[t] - xa, rb = sb[t] - xb;      // exclusive prefix of this thread's chunk
    for (long long i = lo; i < hi; ++i) {
        const long long va = a[i], vb = b[i];
        a[i] = ra; b[i] = rb;
        ra += va; rb += vb;
    }
    if (t == 1023) { totals[0] = sa[1023]; totals[1] = sb[1023]; }
}

// births of stream s in this call that precede (frame f, class rank r): exclusive scan value minus the stream's base
__device__ __forceinline__ long long stream_birth_rank(const Workspace& ws, const int64_t* sfo, int s, long long f, int r, int C) {
    return ws.rbirths[f * C + r] - ws.rbirths[(long long)sfo[s] * C];
}

// local_ids (streaming form): out_object_id = 0-based ordinal of the track's birth inside ITS stream, counted over all
// chunks so far (the reference's id minus one minus the births of the streams in front of it, sort.py:86,140-141)
__global__ void finalize_kernel(long long n_slots, int C, long long id_base, Workspace ws, State st, int local_ids,
                                const int64_t* __restrict__ stream_frame_offsets,
                                int64_t* __restrict__ out_frame, int32_t* __restrict__ out_category,
                                double* __restrict__ out_bbox4, double* __restrict__ out_score,
                                int64_t* __restrict__ out_object_id) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_slots) return;
    const int f = ws.ifr[p];
    if (f < 0) return;
    const int sc = ws.isc[p];
    const int r = ws.rank[sc];
    const long long dst = ws.rcnt[(long long)f * C + r] + ws.ij[p];
    out_frame[dst] = f;
    out_category[dst] = sc % C + 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) out_bbox4[4 * dst + q] = ws.irow[q * n_slots + p];
    out_score[dst] = ws.irow[4 * n_slots + p];
    const int bf = ws.ibf[p];
    if (!local_ids) {
        // sort.py:141 id = count++ ; :288 emits id + 1
        out_object_id[dst] = id_base + ws.rbirths[(long long)bf * C + r] + ws.ibk[p] + 1;
    } else if (bf < 0) {
        out_object_id[dst] = ws.igid[p];                          // born in an earlier chunk
    } else {
        const int s = sc / C;
        out_object_id[dst] = st.s_births[s] + stream_birth_rank(ws, stream_frame_offsets, s, bf, r, C) + ws.ibk[p];
    }
}

// streaming form, end of a chunk: tracks born in this chunk get their stream-local birth ordinal (their (frame, rank)
// coordinates are only meaningful inside the chunk)
__global__ __launch_bounds__(kWave) void resolve_births_kernel(int C, int cap, Workspace ws, State st,
                                                               const int64_t* __restrict__ stream_frame_offsets) {
    const size_t tk = blockIdx.x;
    const int s = (int)(tk / C);
    const int r = ws.rank[tk];
    const int n = st.hdr[tk * 4 + 0];
    for (int i = threadIdx.x; i < n; i += kWave) {
        const int slot = st.order[tk * cap + i];
        const int bf = st.bframe[tk * cap + slot];
        if (bf >= 0) {
            st.gid[tk * cap + slot] = st.s_births[s] + stream_birth_rank(ws, stream_frame_offsets, s, bf, r, C) + st.bk[tk * cap + slot];
            st.bframe[tk * cap + slot] = -1;
        }
    }
}

__global__ void advance_streams_kernel(int n_streams, int C, long long n_frames, Workspace ws, State st,
                                       const int64_t* __restrict__ stream_frame_offsets) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    const long long f0 = stream_frame_offsets[s], f1 = stream_frame_offsets[s + 1];
    if (f1 <= f0) return;
    const long long end = (f1 < n_frames) ? ws.rbirths[f1 * C] : ws.totals[1];
    st.s_births[s] += end - ws.rbirths[f0 * C];
    st.s_frames[s] += f1 - f0;
}

__global__ void state_init_kernel(int n_streams, int C, int cap, State st) {
    const long long nt = (long long)n_streams * C;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nt * cap) st.freel[i] = cap - 1 - (int)(i % cap);
    if (i < nt) {
        st.hdr[i * 4 + 0] = 0; st.hdr[i * 4 + 1] = cap; st.hdr[i * 4 + 2] = 0; st.hdr[i * 4 + 3] = 0;
        st.first_key[i] = -1;
    }
    if (i < n_streams) { st.s_frames[i] = 0; st.s_births[i] = 0; }
}

// reference ids of rows produced by the streaming form: id = id_base + (births of the streams in front) + ordinal + 1
__global__ __launch_bounds__(1024) void stream_prefix_kernel(int n_streams, State st, long long* __restrict__ prefix) {
    __shared__ long long sa[1024];
    const int t = threadIdx.x;
    const int chunk = (n_streams + 1023) / 1024;
    const int lo = t * chunk, hi = (lo + chunk < n_streams) ? lo + chunk : n_streams;
    long long x = 0;
    for (int i = lo; i < hi; ++i) x += st.s_births[i];
    sa[t] = x;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        long long v = 0;
        if (t >= o) v = sa[t - o];
        __syncthreads();
        sa[t] += v;
        __syncthreads();
    }
    long long run = sa[t] - x;
    for (int i = lo; i < hi; ++i) { prefix[i] = run; run += st.s_births[i]; }
    if (t == 1023) prefix[n_streams] = sa[1023];
}

__global__ void global_ids_kernel(long long n, const int32_t* __restrict__ row_stream, const int64_t* __restrict__ local_id,
                                  const long long* __restrict__ prefix, long long id_base, int64_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = id_base + prefix[row_stream[i]] + local_id[i] + 1;
}

__global__ void totals_kernel(const long long* totals, const int* err, int64_t* n_out, int64_t* n_births) {
    n_out[0] = err[0] ? -(int64_t)err[0] : totals[0];
    n_births[0] = totals[1];
}

int pick_caps(int64_t max_frame_dets, const wt_track_params* p, int* cap, int* capN, int* lds_cost, bool* cost_g,
              size_t* lds_bytes, int64_t n_streams) {
    if (!p || p->n_classes < 1 || !p->iou_threshold || !p->score_threshold || p->max_age < 0) {
        wt::set_error("bad wt_track_params");
        return WT_ERR_INVALID;
    }
    const int64_t n = max_frame_dets > 0 ? max_frame_dets : 1;
    const int64_t c = n * ((int64_t)p->max_age + 2);
    if (c > (1 << 20)) { wt::set_error("per-tracker capacity too large (%lld tracks)", (long long)c); return WT_ERR_CAPACITY; }
    *capN = (int)n;
    *cap = (int)c;
    const int64_t full = (int64_t)(*capN) * ((*cap) | 1);
    const size_t mk = wtdev::munkres_lds_bytes(*capN, *cap);
    if (mk > kLdsMunkresMax) { wt::set_error("frame with %lld detections exceeds the LDS budget of the assignment kernel", (long long)n); return WT_ERR_CAPACITY; }
    int64_t budget = kLdsCostFloats;
    if (n_streams > 0 && n_streams * (int64_t)p->n_classes <= 256) {           // few trackers: what the CU's 160 KiB leave next to the bitmaps
        const int64_t room = ((int64_t)160 * 1024 - 512 - (int64_t)mk - (int64_t)wtdev::help_lds_bytes() - 16) / 4;
        budget = room < kLdsCostFloatsFew ? room : kLdsCostFloatsFew;
        if (budget < kLdsCostFloats) budget = kLdsCostFloats;
    }
    *lds_cost = (int)(full < budget ? full : budget);
    *cost_g = full > budget;
    *lds_bytes = wt::align_up((size_t)(*lds_cost) * sizeof(float), 16) + mk;
    return WT_OK;
}

}  // namespace

namespace {
// one tracking pass over the given frames; `st` holds the trackers (fresh when !resume)
int run_tracking(int64_t n_dets, const double* x, const double* y, const double* w, const double* h,
                 const double* score, const int32_t* category, int64_t n_frames, const int64_t* frame_det_offsets,
                 int32_t n_streams, const int64_t* stream_frame_offsets, const double* clip_w, const double* clip_h,
                 const wt_track_params* params, int64_t id_base, int64_t* out_frame, int32_t* out_category,
                 double* out_bbox4, double* out_score, int64_t* out_object_id, int64_t* n_out_dev, int64_t* n_births_dev,
                 const Workspace& ws, const State& st, int resume, int cap, int capN, int lds_cost, bool cost_g, size_t lds,
                 hipStream_t stream) {
    const int C = params->n_classes;
    WT_HIP(hipMemcpyAsync(ws.thr_score, params->score_threshold, sizeof(double) * C, hipMemcpyHostToDevice, stream));
    WT_HIP(hipMemcpyAsync(ws.thr_iou, params->iou_threshold, sizeof(double) * C, hipMemcpyHostToDevice, stream));
    WT_HIP(hipMemsetAsync(ws.err, 0, sizeof(int) * 4, stream));
    WT_HIP(hipMemsetAsync(ws.cnt, 0, sizeof(int) * (size_t)n_frames * C, stream));
    WT_HIP(hipMemsetAsync(ws.births, 0, sizeof(int) * (size_t)n_frames * C, stream));
    WT_HIP(hipMemsetAsync(ws.ifr, 0xFF, sizeof(int) * ((size_t)n_dets + 1), stream));
    WT_HIP(hipMemsetAsync(ws.rcnt, 0, sizeof(long long) * ((size_t)n_frames * C + 1), stream));
    WT_HIP(hipMemsetAsync(ws.rbirths, 0, sizeof(long long) * ((size_t)n_frames * C + 1), stream));
    const unsigned n_trackers = (unsigned)n_streams * (unsigned)C;
    // helper waves (three more waves per tracker for the data-parallel phases) when every tracker can have a CU to itself anyway;
    // WT_SORT_HELPERS=0 switches them off (A/B)
    static const bool helpers_off = getenv("WT_SORT_HELPERS") && getenv("WT_SORT_HELPERS")[0] == '0';
    const bool helpers = !helpers_off && n_trackers <= 256 && (lds + 15) / 16 * 16 + wtdev::help_lds_bytes() <= (size_t)160 * 1024 - 256;
    if (helpers) lds = (lds + 15) / 16 * 16 + wtdev::help_lds_bytes();
    if (lds > 48 * 1024)
        WT_HIP(hipFuncSetAttribute(helpers ? reinterpret_cast<const void*>(sort_streams_kernel<true>) : reinterpret_cast<const void*>(sort_streams_kernel<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (helpers)
        hipLaunchKernelGGL(sort_streams_kernel<true>, dim3(n_trackers), dim3(kHelpWaves * kWave), lds, stream, x, y, w, h, score, category,
                           frame_det_offsets, stream_frame_offsets, clip_w, clip_h, C, (int)params->max_age,
                           (int)params->min_hits, cap, capN, lds_cost, cost_g, (long long)n_dets, ws, st, resume);
    else
        hipLaunchKernelGGL(sort_streams_kernel<false>, dim3(n_trackers), dim3(kWave), lds, stream, x, y, w, h, score, category,
                           frame_det_offsets, stream_frame_offsets, clip_w, clip_h, C, (int)params->max_age,
                           (int)params->min_hits, cap, capN, lds_cost, cost_g, (long long)n_dets, ws, st, resume);
    WT_HIP(hipGetLastError());
    hipLaunchKernelGGL(rank_classes_kernel, dim3((n_streams + 255) / 256), dim3(256), 0, stream, (int)n_streams, C, ws, st);
    const long long ne = (long long)n_frames * C;
    hipLaunchKernelGGL(gather_ranked_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, stream,
                       (long long)n_frames, (int)n_streams, C, stream_frame_offsets, ws);
    hipLaunchKernelGGL(scan2_kernel, dim3(1), dim3(1024), 0, stream, ne, ws.rcnt, ws.rbirths, ws.totals);
    if (n_dets > 0)
        hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n_dets + 255) / 256)), dim3(256), 0, stream,
                           (long long)n_dets, C, (long long)id_base, ws, st, resume, stream_frame_offsets, out_frame,
                           out_category, out_bbox4, out_score, out_object_id);
    if (resume) {
        hipLaunchKernelGGL(resolve_births_kernel, dim3(n_trackers), dim3(kWave), 0, stream, C, cap, ws, st,
                           stream_frame_offsets);
        hipLaunchKernelGGL(advance_streams_kernel, dim3((n_streams + 255) / 256), dim3(256), 0, stream, (int)n_streams, C,
                           (long long)n_frames, ws, st, stream_frame_offsets);
    }
    hipLaunchKernelGGL(totals_kernel, dim3(1), dim3(1), 0, stream, (const long long*)ws.totals, ws.err, n_out_dev,
                       n_births_dev);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

inline char* align256(void* p) {
    const uintptr_t mis = (uintptr_t)p & 255;
    return (char*)p + (mis ? 256 - mis : 0);
}
}  // namespace

extern "C" {

size_t wt_track_streams_workspace(int64_t n_dets, int64_t n_frames, int32_t n_streams, int64_t max_frame_dets,
                                  const wt_track_params* params) {
    int cap, capN, lds_cost; bool cost_g; size_t lds;
    if (pick_caps(max_frame_dets, params, &cap, &capN, &lds_cost, &cost_g, &lds, n_streams) != WT_OK) return 0;
    return carve_ws(nullptr, n_dets, n_frames, n_streams, params->n_classes, cap, capN, cost_g).bytes +
           carve_state(nullptr, n_streams, params->n_classes, cap).bytes + 256;
}

int wt_track_streams_dev(int64_t n_dets, const double* x, const double* y, const double* w, const double* h,
                         const double* score, const int32_t* category,
                         int64_t n_frames, const int64_t* frame_det_offsets,
                         int32_t n_streams, const int64_t* stream_frame_offsets,
                         const double* clip_w, const double* clip_h, int64_t max_frame_dets,
                         const wt_track_params* params, int64_t id_base,
                         int64_t* out_frame, int32_t* out_category, double* out_bbox4, double* out_score,
                         int64_t* out_object_id, int64_t* n_out_dev, int64_t* n_births_dev,
                         void* workspace, size_t workspace_bytes, void* stream_) {
    WT_TRY(wt::ensure_device());
    hipStream_t stream = (hipStream_t)stream_;
    int cap, capN, lds_cost; bool cost_g; size_t lds;
    WT_TRY(pick_caps(max_frame_dets, params, &cap, &capN, &lds_cost, &cost_g, &lds, n_streams));
    const int C = params->n_classes;
    if (n_streams <= 0 || n_frames <= 0) {
        WT_HIP(hipMemsetAsync(n_out_dev, 0, sizeof(int64_t), stream));
        WT_HIP(hipMemsetAsync(n_births_dev, 0, sizeof(int64_t), stream));
        return WT_OK;
    }
    char* base = align256(workspace);
    Workspace ws = carve_ws(base, n_dets, n_frames, n_streams, C, cap, capN, cost_g);
    State st = carve_state(base + ws.bytes, n_streams, C, cap);
    if (!workspace || workspace_bytes < ws.bytes + st.bytes + 256) {
        wt::set_error("tracking workspace too small: need %zu bytes, have %zu", ws.bytes + st.bytes + 256, workspace_bytes);
        return WT_ERR_CAPACITY;
    }
    return run_tracking(n_dets, x, y, w, h, score, category, n_frames, frame_det_offsets, n_streams, stream_frame_offsets,
                        clip_w, clip_h, params, id_base, out_frame, out_category, out_bbox4, out_score, out_object_id,
                        n_out_dev, n_births_dev, ws, st, 0, cap, capN, lds_cost, cost_g, lds, stream);
}

/* ---- streaming form: the trackers persist in `state` between chunks of frames ---- */
size_t wt_track_state_bytes(int32_t n_streams, int64_t max_frame_dets, const wt_track_params* params) {
    int cap, capN, lds_cost; bool cost_g; size_t lds;
    if (n_streams <= 0 || pick_caps(max_frame_dets, params, &cap, &capN, &lds_cost, &cost_g, &lds, n_streams) != WT_OK) return 0;
    return carve_state(nullptr, n_streams, params->n_classes, cap).bytes + 256;
}

int wt_track_state_init_dev(void* state, size_t state_bytes, int32_t n_streams, int64_t max_frame_dets,
                            const wt_track_params* params, void* stream_) {
    WT_TRY(wt::ensure_device());
    int cap, capN, lds_cost; bool cost_g; size_t lds;
    WT_TRY(pick_caps(max_frame_dets, params, &cap, &capN, &lds_cost, &cost_g, &lds, n_streams));
    if (n_streams <= 0) { wt::set_error("n_streams must be positive"); return WT_ERR_INVALID; }
    const int C = params->n_classes;
    State st = carve_state(align256(state), n_streams, C, cap);
    if (!state || state_bytes < st.bytes + 256) {
        wt::set_error("tracker state too small: need %zu bytes, have %zu", st.bytes + 256, state_bytes);
        return WT_ERR_CAPACITY;
    }
    const long long items = (long long)n_streams * C * cap;
    hipLaunchKernelGGL(state_init_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       (int)n_streams, C, cap, st);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

size_t wt_track_chunk_workspace(int64_t n_dets, int64_t n_frames, int32_t n_streams, int64_t max_frame_dets,
                                const wt_track_params* params) {
    int cap, capN, lds_cost; bool cost_g; size_t lds;
    if (pick_caps(max_frame_dets, params, &cap, &capN, &lds_cost, &cost_g, &lds, n_streams) != WT_OK) return 0;
    return carve_ws(nullptr, n_dets, n_frames, n_streams, params->n_classes, cap, capN, cost_g).bytes + 256;
}

int wt_track_chunk_dev(void* state, size_t state_bytes,
                       int64_t n_dets, const double* x, const double* y, const double* w, const double* h,
                       const double* score, const int32_t* category,
                       int64_t n_frames, const int64_t* frame_det_offsets,
                       int32_t n_streams, const int64_t* stream_frame_offsets,
                       const double* clip_w, const double* clip_h, int64_t max_frame_dets,
                       const wt_track_params* params,
                       int64_t* out_frame, int32_t* out_category, double* out_bbox4, double* out_score,
                       int64_t* out_local_id, int64_t* n_out_dev, int64_t* n_births_dev,
                       void* workspace, size_t workspace_bytes, void* stream_) {
    WT_TRY(wt::ensure_device());
    hipStream_t stream = (hipStream_t)stream_;
    int cap, capN, lds_cost; bool cost_g; size_t lds;
    WT_TRY(pick_caps(max_frame_dets, params, &cap, &capN, &lds_cost, &cost_g, &lds, n_streams));
    const int C = params->n_classes;
    if (n_streams <= 0) { wt::set_error("n_streams must be positive"); return WT_ERR_INVALID; }
    if (n_frames <= 0) {
        WT_HIP(hipMemsetAsync(n_out_dev, 0, sizeof(int64_t), stream));
        WT_HIP(hipMemsetAsync(n_births_dev, 0, sizeof(int64_t), stream));
        return WT_OK;
    }
    State st = carve_state(align256(state), n_streams, C, cap);
    if (!state || state_bytes < st.bytes + 256) {
        wt::set_error("tracker state too small: need %zu bytes, have %zu", st.bytes + 256, state_bytes);
        return WT_ERR_CAPACITY;
    }
    Workspace ws = carve_ws(align256(workspace), n_dets, n_frames, n_streams, C, cap, capN, cost_g);
    if (!workspace || workspace_bytes < ws.bytes + 256) {
        wt::set_error("tracking workspace too small: need %zu bytes, have %zu", ws.bytes + 256, workspace_bytes);
        return WT_ERR_CAPACITY;
    }
    return run_tracking(n_dets, x, y, w, h, score, category, n_frames, frame_det_offsets, n_streams, stream_frame_offsets,
                        clip_w, clip_h, params, 0, out_frame, out_category, out_bbox4, out_score, out_local_id,
                        n_out_dev, n_births_dev, ws, st, 1, cap, capN, lds_cost, cost_g, lds, stream);
}

int wt_track_global_ids_dev(const void* state, size_t state_bytes, int32_t n_streams, int64_t max_frame_dets,
                            const wt_track_params* params, int64_t n_rows, const int32_t* row_stream,
                            const int64_t* local_id, int64_t id_base, int64_t* out_object_id, int64_t* stream_birth_prefix,
                            void* stream_) {
    WT_TRY(wt::ensure_device());
    hipStream_t stream = (hipStream_t)stream_;
    int cap, capN, lds_cost; bool cost_g; size_t lds;
    WT_TRY(pick_caps(max_frame_dets, params, &cap, &capN, &lds_cost, &cost_g, &lds, n_streams));
    if (n_streams <= 0 || !stream_birth_prefix) { wt::set_error("bad arguments"); return WT_ERR_INVALID; }
    State st = carve_state(align256(const_cast<void*>(state)), n_streams, params->n_classes, cap);
    if (!state || state_bytes < st.bytes + 256) { wt::set_error("tracker state too small"); return WT_ERR_CAPACITY; }
    hipLaunchKernelGGL(stream_prefix_kernel, dim3(1), dim3(1024), 0, stream, (int)n_streams, st,
                       (long long*)stream_birth_prefix);
    if (n_rows > 0)
        hipLaunchKernelGGL(global_ids_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream,
                           (long long)n_rows, row_stream, local_id, (const long long*)stream_birth_prefix,
                           (long long)id_base, out_object_id);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

int wt_track_streams_host(int64_t n_dets, const double* x, const double* y, const double* w, const double* h,
                          const double* score, const int32_t* category,
                          int64_t n_frames, const int64_t* frame_det_offsets,
                          int32_t n_streams, const int64_t* stream_frame_offsets,
                          const double* clip_w, const double* clip_h,
                          const wt_track_params* params, int64_t id_base,
                          int64_t* out_frame, int32_t* out_category, double* out_bbox4, double* out_score,
                          int64_t* out_object_id, int64_t* n_out, int64_t* n_births) {
    WT_TRY(wt::ensure_device());
    *n_out = 0;
    *n_births = 0;
    if (n_streams <= 0 || n_frames <= 0) return WT_OK;
    if (!params) { wt::set_error("params is NULL"); return WT_ERR_INVALID; }
    int64_t max_frame = 0;
    for (int64_t f = 0; f < n_frames; ++f) {
        const int64_t c = frame_det_offsets[f + 1] - frame_det_offsets[f];
        if (c < 0) { wt::set_error("frame_det_offsets must be non-decreasing"); return WT_ERR_INVALID; }
        if (c > max_frame) max_frame = c;
    }
    if (frame_det_offsets[n_frames] != n_dets || stream_frame_offsets[n_streams] != n_frames) {
        wt::set_error("CSR offsets do not cover n_dets / n_frames");
        return WT_ERR_INVALID;
    }
    for (int64_t i = 0; i < n_dets; ++i)
        if (category[i] < 1 || category[i] > params->n_classes) {
            wt::set_error("category %d outside 1..%d", (int)category[i], (int)params->n_classes);
            return WT_ERR_INVALID;
        }
    const size_t nd = (size_t)n_dets;
    wt::DevBuf dx, dy, dw, dh, ds, dc, dfo, dso, dcw, dch, of, oc, ob, os, oi, dn, dws;
    WT_TRY(dx.alloc(8 * nd)); WT_TRY(dy.alloc(8 * nd)); WT_TRY(dw.alloc(8 * nd)); WT_TRY(dh.alloc(8 * nd));
    WT_TRY(ds.alloc(8 * nd)); WT_TRY(dc.alloc(4 * nd));
    WT_TRY(dfo.alloc(8 * (size_t)(n_frames + 1))); WT_TRY(dso.alloc(8 * (size_t)(n_streams + 1)));
    WT_TRY(dcw.alloc(8 * (size_t)n_streams)); WT_TRY(dch.alloc(8 * (size_t)n_streams));
    WT_TRY(of.alloc(8 * (nd + 1))); WT_TRY(oc.alloc(4 * (nd + 1))); WT_TRY(ob.alloc(32 * (nd + 1)));
    WT_TRY(os.alloc(8 * (nd + 1))); WT_TRY(oi.alloc(8 * (nd + 1))); WT_TRY(dn.alloc(16));
    if (nd) {
        WT_HIP(hipMemcpy(dx.p, x, 8 * nd, hipMemcpyHostToDevice)); WT_HIP(hipMemcpy(dy.p, y, 8 * nd, hipMemcpyHostToDevice));
        WT_HIP(hipMemcpy(dw.p, w, 8 * nd, hipMemcpyHostToDevice)); WT_HIP(hipMemcpy(dh.p, h, 8 * nd, hipMemcpyHostToDevice));
        WT_HIP(hipMemcpy(ds.p, score, 8 * nd, hipMemcpyHostToDevice));
        WT_HIP(hipMemcpy(dc.p, category, 4 * nd, hipMemcpyHostToDevice));
    }
    WT_HIP(hipMemcpy(dfo.p, frame_det_offsets, 8 * (size_t)(n_frames + 1), hipMemcpyHostToDevice));
    WT_HIP(hipMemcpy(dso.p, stream_frame_offsets, 8 * (size_t)(n_streams + 1), hipMemcpyHostToDevice));
    std::vector<double> zeros;
    if (!clip_w || !clip_h) zeros.assign((size_t)n_streams, 0.0);
    WT_HIP(hipMemcpy(dcw.p, clip_w ? clip_w : zeros.data(), 8 * (size_t)n_streams, hipMemcpyHostToDevice));
    WT_HIP(hipMemcpy(dch.p, clip_h ? clip_h : zeros.data(), 8 * (size_t)n_streams, hipMemcpyHostToDevice));
    const size_t wsb = wt_track_streams_workspace(n_dets, n_frames, n_streams, max_frame, params);
    if (!wsb) return WT_ERR_CAPACITY;
    WT_TRY(dws.alloc(wsb));
    WT_TRY(wt_track_streams_dev(n_dets, dx.as<double>(), dy.as<double>(), dw.as<double>(), dh.as<double>(),
                                ds.as<double>(), dc.as<int32_t>(), n_frames, dfo.as<int64_t>(), n_streams,
                                dso.as<int64_t>(), dcw.as<double>(), dch.as<double>(), max_frame, params, id_base,
                                of.as<int64_t>(), oc.as<int32_t>(), ob.as<double>(), os.as<double>(), oi.as<int64_t>(),
                                dn.as<int64_t>(), dn.as<int64_t>() + 1, dws.p, wsb, nullptr));
    WT_HIP(hipDeviceSynchronize());
    int64_t res[2] = {0, 0};
    WT_HIP(hipMemcpy(res, dn.p, 16, hipMemcpyDeviceToHost));
    if (res[0] < 0) {
        wt::set_error("SORT kernel reported status %lld (4 = capacity, 5 = assignment did not converge)", (long long)-res[0]);
        return (int)-res[0];
    }
    *n_out = res[0];
    *n_births = res[1];
    const size_t k = (size_t)res[0];
    if (k) {
        WT_HIP(hipMemcpy(out_frame, of.p, 8 * k, hipMemcpyDeviceToHost));
        WT_HIP(hipMemcpy(out_category, oc.p, 4 * k, hipMemcpyDeviceToHost));
        WT_HIP(hipMemcpy(out_bbox4, ob.p, 32 * k, hipMemcpyDeviceToHost));
        WT_HIP(hipMemcpy(out_score, os.p, 8 * k, hipMemcpyDeviceToHost));
        WT_HIP(hipMemcpy(out_object_id, oi.p, 8 * k, hipMemcpyDeviceToHost));
    }
    return WT_OK;
}

}  // extern "C"

#ifdef WT_PHASE_TIMING
// experiments only (-DWT_PHASE_TIMING): cycles per tracker_step phase summed over all waves
extern "C" int wt_debug_phase_cycles(unsigned long long* out12, int reset) {
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(wtdev::wt_phase), 96) != hipSuccess) return 1;
    if (reset) { unsigned long long z[12] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(wtdev::wt_phase), z, 96); }
    return 0;
}
#endif
