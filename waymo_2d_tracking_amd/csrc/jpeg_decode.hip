// GPU JPEG decode - see jpeg_core.h for the scheme.  One call = one image:
//   host   : marker parse, Huffman / quantisation tables, byte unstuffing into restart segments (a memchr pass), one
//            pinned staging blob -> ONE hipMemcpyAsync
//   device : jpeg_cand_kernel (x4), jpeg_map_kernel, jpeg_resolve_kernel, jpeg_pick_kernel - candidate (start -> exit) decodes
//                                     per subsequence and the true chain through them by a scan over index maps (jpeg_core.h)
//            jpeg_sync_kernel (x2)  - subsequence synchronisation by iteration (settles what the candidates left open; a
//                                     no-op pass otherwise), bitstream + tables in LDS
//            jpeg_scan_kernel       - chain check + first block index / DC predictions of every subsequence (segmented scan)
//            jpeg_write_kernel      - final decode pass, coefficients (int16, natural order, DC resolved) to HBM
//            jpeg_idct_kernel       - dequantise + jidctint.c islow, 8 threads per block, planes in u8
//            jpeg_color_kernel      - fancy h2v1 / h2v2 upsampling + YCbCr -> RGB, (H, W, 3) u8 out
// The call returns after its stream has drained (it has to read the "chain settled" flag); when the flag is not set
// (streams that do not synchronise inside a subsequence) it keeps launching sync rounds until it is and repeats the tail.
#include <mutex>
#include <vector>
#include "common.h"
#include "jpeg_core.h"
#include "jpeg_host.h"
#include "../../include/waymodet.h"

namespace {

using jd::Header;
using jd::HuffLut;
using jd::State;
using jdh::Layout;
using jdh::Parsed;

constexpr int kSyncThreads = 256;
constexpr int kLdsWords = kSyncThreads * jd::SUB_WORDS + 4;

// ---------------------------------------------------------------------------------------------------------------- host
int fail(const char* msg) {
    wt::set_error("wd_jpeg: %s", msg);
    return WT_ERR_INVALID;
}

// ------------------------------------------------------------------------------------------------------------- kernels
struct Dev {                         // device pointers into the blob + work buffers
    const Header* hd;
    const HuffLut* luts;
    const uint32_t* seg_first_sub;
    const uint32_t* seg_end_bit;
    const int32_t* sub_seg;
    State* start;
    State* exit;
    const uint32_t* stream;          // bytes as stored (big-endian bit order): words are byte-swapped on load
    int4* cnt;                       // per subsequence: blocks completed, DC differences summed per component
    int4* base;                      // per subsequence: first block index, DC predictions at its start
    State* cand_s;                   // [subsequence][CAND_MAX]: candidate start states ...
    State* cand_e;                   // ... their exit states ...
    int4* cand_c;                    // ... and counts
    int32_t* cand_n;                 // [3][subsequence]: number of candidates after launch l in buffer l % 3
    unsigned long long* cand_map;    // [subsequence]: candidate index of i - 1 -> candidate index of i (4 bits each)
    uint8_t* cand_pick;              // [subsequence]: index of the candidate on the true chain (CAND_NONE = unresolved)
    int16_t* coef;
    uint8_t* planes;
    int32_t* flags;                  // [0] chain not settled, [1] a segment came up short, [2] / [3] statistics
    int nsub, stream_words;
};

__device__ __forceinline__ void stage(const Dev& d, uint32_t* words, HuffLut* luts) {
    const uint32_t w0 = blockIdx.x * (uint32_t)(kSyncThreads * jd::SUB_WORDS);
    for (int j = threadIdx.x; j < kLdsWords; j += kSyncThreads) {
        const uint32_t g = w0 + (uint32_t)j;
        words[j] = g < (uint32_t)d.stream_words ? __builtin_bswap32(d.stream[g]) : 0xFFFFFFFFu;
    }
    const uint32_t* src = reinterpret_cast<const uint32_t*>(d.luts);
    uint32_t* dst = reinterpret_cast<uint32_t*>(luts);
    for (int j = threadIdx.x; j < (int)(4 * sizeof(HuffLut) / 4); j += kSyncThreads) dst[j] = src[j];
}

__global__ __launch_bounds__(kSyncThreads) void jpeg_sync_kernel(Dev d) {
    __shared__ uint32_t words[kLdsWords];
    __shared__ HuffLut luts[4];
    __shared__ State exits[kSyncThreads];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kSyncThreads + tid;
    const bool valid = i < d.nsub;
    {   // nothing to do for this workgroup (the usual case behind the candidate sets)?  Then leave before staging 66 KB.
        int need = 0;
        if (valid) {
            const int sg = d.sub_seg[i];
            State want;
            if (d.seg_first_sub[sg] == (uint32_t)i) { want.p = (uint32_t)i * jd::SUB_BITS; want.bk = 0; } else want = d.exit[i - 1];
            need = !jd::same(want, d.start[i]);
        }
        if (!__syncthreads_or(need)) return;
    }
    stage(d, words, luts);
    const jd::Sel sel = jd::make_sel(d.hd);
    const uint32_t w0 = blockIdx.x * (uint32_t)(kSyncThreads * jd::SUB_WORDS);
    int seg = 0;
    bool first = false;
    uint32_t seg_end = 0, bound = 0;
    State s_cur{jd::NO_STATE, jd::NO_STATE}, e_cur{0, 0}, from_prev{0, 0};
    jd::Counts c_cur{0, {0, 0, 0}};
    if (valid) {
        seg = d.sub_seg[i];
        first = d.seg_first_sub[seg] == (uint32_t)i;
        seg_end = d.seg_end_bit[seg];
        bound = (uint32_t)(i + 1) * jd::SUB_BITS;
        bound = bound < seg_end ? bound : seg_end;
        s_cur = d.start[i];
        e_cur = d.exit[i];
        const int4 c = d.cnt[i];
        c_cur.n = c.x; c_cur.dc[0] = c.y; c_cur.dc[1] = c.z; c_cur.dc[2] = c.w;
        if (tid == 0 && !first) from_prev = d.exit[i - 1];
    }
    exits[tid] = e_cur;
    __syncthreads();
    int iters = 0, decodes = 0;
    for (;;) {
        ++iters;
        State want;
        if (first) { want.p = (uint32_t)i * jd::SUB_BITS; want.bk = 0; }
        else {
            const State left = exits[tid > 0 ? tid - 1 : 0];
            want.p = tid == 0 ? from_prev.p : left.p;
            want.bk = tid == 0 ? from_prev.bk : left.bk;
        }
        __syncthreads();
        int changed = 0;
        if (valid && !jd::same(want, s_cur)) {
            s_cur = want;
            ++decodes;
            const State e = jd::run<false>(s_cur, bound, seg_end, words, w0, luts, sel, c_cur, nullptr, 0, 0, nullptr);
            if (!jd::same(e, e_cur)) { e_cur = e; exits[tid] = e; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }
    if (valid) { d.start[i] = s_cur; d.exit[i] = e_cur; d.cnt[i] = make_int4(c_cur.n, c_cur.dc[0], c_cur.dc[1], c_cur.dc[2]); }
    // statistics (wd_jpeg_last_stats): longest iteration count of a workgroup, subsequence decodes in total
    if (tid == 0) atomicMax(&d.flags[2], iters);
    if (decodes) atomicAdd(&d.flags[3], decodes);
}

constexpr int kCandSubs = kSyncThreads / jd::CAND_SLOTS;                 // subsequences per workgroup of jpeg_cand_kernel
constexpr int kCandWords = kCandSubs * jd::SUB_WORDS + 4;

// Candidate sets, one launch = one round (jpeg_core.h).  CAND_SLOTS threads per subsequence.  launch 0: slot h decodes from
// (first bit, block index h, DC next); later launches: slot r decodes from the r-th exit of the predecessor's candidates (as of the
// previous launch) that is not yet among this subsequence's starts.  Records are append-only; counts are double-buffered.
__global__ __launch_bounds__(kSyncThreads) void jpeg_cand_kernel(Dev d, int launch) {
    __shared__ uint32_t words[kCandWords];
    __shared__ HuffLut luts[4];
    const int tid = threadIdx.x, slot = tid % jd::CAND_SLOTS;
    const int i = blockIdx.x * kCandSubs + tid / jd::CAND_SLOTS;
    const uint32_t w0 = blockIdx.x * (uint32_t)(kCandSubs * jd::SUB_WORDS);
    const jd::Sel sel = jd::make_sel(d.hd);
    // what does this thread decode?  (decided from global memory first: in the later launches most workgroups have nothing to do
    // and leave before staging their 38 KB of bitstream and tables)
    State s{jd::NO_STATE, jd::NO_STATE};
    int at = -1;
    uint32_t seg_end = 0, bound = 0;
    if (i < d.nsub) {
        const int seg = d.sub_seg[i];
        const bool first = d.seg_first_sub[seg] == (uint32_t)i;
        seg_end = d.seg_end_bit[seg];
        bound = (uint32_t)(i + 1) * jd::SUB_BITS;
        bound = bound < seg_end ? bound : seg_end;
        int32_t* n_new = d.cand_n + (size_t)(launch % 3) * d.nsub;
        const int32_t* n_old = d.cand_n + (size_t)((launch + 2) % 3) * d.nsub;
        const int32_t* n_old2 = d.cand_n + (size_t)((launch + 1) % 3) * d.nsub;
        if (launch == 0) {
            const int n = first ? 1 : sel.bpm;
            if (slot == 0) n_new[i] = n;
            if (slot < n) { s.p = (uint32_t)i * jd::SUB_BITS; s.bk = first ? 0u : (uint32_t)slot << 8; at = slot; }
        } else {
            const int mine = n_old[i];
            int fresh = 0;
            if (!first) {
                const int theirs = n_old[i - 1], seen = launch >= 2 ? n_old2[i - 1] : 0;      // [seen, theirs): added by the previous launch
                if (theirs > seen) {
                    const State* pe = d.cand_e + (size_t)(i - 1) * jd::CAND_MAX;
                    const State* my_s = d.cand_s + (size_t)i * jd::CAND_MAX;
                    for (int a = seen; a < theirs; ++a) {
                        const State e = pe[a];
                        bool known = e.p == jd::NO_STATE;
                        for (int b = 0; b < mine; ++b) known |= jd::same(my_s[b], e);
                        for (int b = seen; b < a; ++b) known |= jd::same(pe[b], e);        // the same exit twice among the new ones
                        if (known) continue;
                        if (fresh == slot && mine + fresh < jd::CAND_MAX) { s = e; at = mine + fresh; }
                        ++fresh;
                    }
                }
            }
            fresh = fresh < jd::CAND_SLOTS ? fresh : jd::CAND_SLOTS;         // one decode per slot and launch
            if (slot == 0) n_new[i] = mine + fresh < jd::CAND_MAX ? mine + fresh : jd::CAND_MAX;
        }
        if (at >= 0 && (s.p < w0 * 32u || s.p > bound)) {                   // cannot be decoded from this workgroup's window: an empty record
            d.cand_s[(size_t)i * jd::CAND_MAX + at] = State{jd::NO_STATE, jd::NO_STATE};
            d.cand_e[(size_t)i * jd::CAND_MAX + at] = State{jd::NO_STATE, jd::NO_STATE};
            at = -1;
        }
    }
    if (!__syncthreads_or(at >= 0 ? 1 : 0)) return;
    for (int j = tid; j < kCandWords; j += kSyncThreads) {
        const uint32_t g = w0 + (uint32_t)j;
        words[j] = g < (uint32_t)d.stream_words ? __builtin_bswap32(d.stream[g]) : 0xFFFFFFFFu;
    }
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(d.luts);
        uint32_t* dst = reinterpret_cast<uint32_t*>(luts);
        for (int j = tid; j < (int)(4 * sizeof(HuffLut) / 4); j += kSyncThreads) dst[j] = src[j];
    }
    __syncthreads();
    if (at < 0) return;
    jd::Counts c{0, {0, 0, 0}};
    const State e = jd::run<false>(s, bound, seg_end, words, w0, luts, sel, c, nullptr, 0, 0, nullptr);
    d.cand_s[(size_t)i * jd::CAND_MAX + at] = s;
    d.cand_e[(size_t)i * jd::CAND_MAX + at] = e;
    d.cand_c[(size_t)i * jd::CAND_MAX + at] = make_int4(c.n, c.dc[0], c.dc[1], c.dc[2]);
}

// map of the link i - 1 -> i: candidate a of i - 1 -> the candidate of i that starts at a's exit (CAND_NONE if there is none)
__global__ __launch_bounds__(256) void jpeg_map_kernel(Dev d, int parity) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.nsub) return;
    const int32_t* cn = d.cand_n + (size_t)parity * d.nsub;
    uint64_t mm = jd::map_identity();
    if (d.seg_first_sub[d.sub_seg[i]] != (uint32_t)i) {
        const int theirs = cn[i - 1], mine = cn[i];
        State ms[jd::CAND_MAX];
#pragma unroll
        for (int b = 0; b < jd::CAND_MAX; ++b) ms[b] = b < mine ? d.cand_s[(size_t)i * jd::CAND_MAX + b] : State{jd::NO_STATE, jd::NO_STATE};
        mm = (uint64_t)jd::CAND_NONE << 60;                                   // CAND_NONE -> CAND_NONE
#pragma unroll
        for (int a = 0; a < 15; ++a) {
            uint32_t to = jd::CAND_NONE;
            if (a < theirs && a < jd::CAND_MAX) {
                const State e = d.cand_e[(size_t)(i - 1) * jd::CAND_MAX + a];
#pragma unroll
                for (int b = 0; b < jd::CAND_MAX; ++b) to = (e.p != jd::NO_STATE && jd::same(ms[b], e)) ? (uint32_t)b : to;
            }
            mm |= (uint64_t)to << (4 * a);
        }
    }
    d.cand_map[i] = mm;
}

// one workgroup: which candidate of every subsequence lies on the true chain?  A segment's first subsequence has one candidate,
// index 0; segmented scan over the link maps (composition), then start / exit / counts of every resolved subsequence are copied
// from its candidate.
__global__ __launch_bounds__(1024) void jpeg_resolve_kernel(Dev d) {
    constexpr int ITEMS = 8;
    __shared__ uint64_t maps[1024];
    __shared__ int flgs[1024];
    __shared__ uint64_t carry_s;
    const int tid = threadIdx.x, nsub = d.nsub;
    if (tid == 0) carry_s = jd::map_identity();
    __syncthreads();
    for (int c0 = 0; c0 < nsub; c0 += 1024 * ITEMS) {
        uint64_t m[ITEMS];
        int firsts = 0, any = 0;
        uint64_t total = jd::map_identity();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int i = c0 + tid * ITEMS + j;
            m[j] = jd::map_identity();
            if (i < nsub) {
                const int f = d.seg_first_sub[d.sub_seg[i]] == (uint32_t)i;
                firsts |= f << j;
                m[j] = d.cand_map[i];
                total = f ? jd::map_identity() : jd::map_compose(m[j], total);
                any |= f;
            }
        }
        maps[tid] = total;
        flgs[tid] = any;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            uint64_t a = maps[tid];
            int g = flgs[tid];
            if (tid >= off && !g) { a = jd::map_compose(a, maps[tid - off]); g = flgs[tid - off]; }
            __syncthreads();
            maps[tid] = a;
            flgs[tid] = g;
            __syncthreads();
        }
        const uint64_t chunk = carry_s;
        // candidate index of the subsequence in front of this thread's first item (0 at a segment start)
        uint32_t b = jd::map_at(chunk, 0);
        if (tid > 0) b = flgs[tid - 1] ? jd::map_at(maps[tid - 1], 0) : jd::map_at(maps[tid - 1], jd::map_at(chunk, 0));
        const uint64_t last = flgs[1023] ? maps[1023] : jd::map_compose(maps[1023], chunk);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int i = c0 + tid * ITEMS + j;
            if (i >= nsub) continue;
            b = ((firsts >> j) & 1) ? 0u : jd::map_at(m[j], b);
            d.cand_pick[i] = (uint8_t)b;
        }
        __syncthreads();
        if (tid == 0) carry_s = last;
        __syncthreads();
    }
}

// start / exit / counts of every resolved subsequence = its candidate on the true chain
__global__ __launch_bounds__(256) void jpeg_pick_kernel(Dev d, int parity) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.nsub) return;
    const uint32_t b = d.cand_pick[i];
    if (b == jd::CAND_NONE || (int)b >= d.cand_n[(size_t)parity * d.nsub + i]) return;
    const State s = d.cand_s[(size_t)i * jd::CAND_MAX + b];
    if (s.p == jd::NO_STATE) return;
    d.start[i] = s;
    d.exit[i] = d.cand_e[(size_t)i * jd::CAND_MAX + b];
    d.cnt[i] = d.cand_c[(size_t)i * jd::CAND_MAX + b];
}

// one workgroup: is every start state the exit state of its predecessor (or the known segment start)?  Running values of
// the sequential decoder at the start of every subsequence = segmented exclusive scan over (blocks, DC differences)
__global__ __launch_bounds__(1024) void jpeg_scan_kernel(Dev d) {
    constexpr int ITEMS = 4;
    __shared__ int4 vals[1024];
    __shared__ int flgs[1024];
    __shared__ int4 carry_s;
    const int tid = threadIdx.x;
    const Header* hd = d.hd;
    const int bpm = hd->bpm, ri = hd->ri, total = hd->total_blocks;
    if (tid == 0) carry_s = make_int4(0, 0, 0, 0);
    __syncthreads();
    auto add = [](const int4 a, const int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    int bad = 0, shortfall = 0;
    for (int c0 = 0; c0 < d.nsub; c0 += 1024 * ITEMS) {
        int4 v[ITEMS];
        int segs[ITEMS];
        int4 run = make_int4(0, 0, 0, 0);
        int any = 0, firsts = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int i = c0 + tid * ITEMS + j;
            int4 c = make_int4(0, 0, 0, 0);
            int f = 0;
            segs[j] = 0;
            if (i < d.nsub) {
                segs[j] = d.sub_seg[i];
                f = d.seg_first_sub[segs[j]] == (uint32_t)i;
                c = d.cnt[i];
                const State s = d.start[i];
                State want;
                if (f) { want.p = (uint32_t)i * jd::SUB_BITS; want.bk = 0; } else want = d.exit[i - 1];
                bad |= !jd::same(s, want);
            }
            run = f ? c : add(run, c);
            any |= f;
            firsts |= f << j;
            v[j] = run;                               // inclusive within the thread since its last segment start
        }
        vals[tid] = run;
        flgs[tid] = any;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {   // inclusive segmented scan over the thread totals
            int4 a = vals[tid];
            int g = flgs[tid];
            if (tid >= off && !g) { a = add(a, vals[tid - off]); g = flgs[tid - off]; }
            __syncthreads();
            vals[tid] = a;
            flgs[tid] = g;
            __syncthreads();
        }
        const int4 chunk = carry_s;
        int4 carry = chunk;                           // running values in front of this thread's first item
        if (tid > 0) carry = flgs[tid - 1] ? vals[tid - 1] : add(vals[tid - 1], chunk);
        const int4 last = flgs[1023] ? vals[1023] : add(vals[1023], chunk);
        bool seen = false;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int i = c0 + tid * ITEMS + j;
            if (i >= d.nsub) continue;
            seen |= (firsts >> j) & 1;
            const int4 incl = seen ? v[j] : add(v[j], carry);
            const int4 c = d.cnt[i];
            const int seg_block0 = segs[j] * ri * bpm;
            d.base[i] = make_int4(seg_block0 + incl.x - c.x, incl.y - c.y, incl.z - c.z, incl.w - c.w);
            if (d.seg_first_sub[segs[j] + 1] == (uint32_t)(i + 1)) {            // last subsequence of its segment
                int expect = total - seg_block0;
                expect = expect < ri * bpm ? expect : ri * bpm;
                shortfall |= incl.x < expect;
            }
        }
        __syncthreads();
        if (tid == 0) carry_s = last;
        __syncthreads();
    }
    if (bad) atomicOr(&d.flags[0], 1);
    if (shortfall) atomicOr(&d.flags[1], 1);
}

// final pass: decode from the settled start states and write the coefficients, DC prediction already resolved
__global__ __launch_bounds__(kSyncThreads) void jpeg_write_kernel(Dev d) {
    __shared__ uint32_t words[kLdsWords];
    __shared__ HuffLut luts[4];
    __shared__ uint8_t nat[80];                      // zig-zag -> natural order next to the stream (a constant-memory load per coefficient otherwise)
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kSyncThreads + tid;
    stage(d, words, luts);
    if (tid < 80) nat[tid] = (uint8_t)jd::natural(tid);
    const jd::Sel sel = jd::make_sel(d.hd);
    __syncthreads();
    if (i >= d.nsub) return;
    const Header* hd = d.hd;
    const uint32_t w0 = blockIdx.x * (uint32_t)(kSyncThreads * jd::SUB_WORDS);
    const int seg = d.sub_seg[i];
    const uint32_t seg_end = d.seg_end_bit[seg];
    uint32_t bound = (uint32_t)(i + 1) * jd::SUB_BITS;
    bound = bound < seg_end ? bound : seg_end;
    const int seg_block0 = seg * hd->ri * hd->bpm;
    int block_end = seg_block0 + hd->ri * hd->bpm;
    block_end = block_end < hd->total_blocks ? block_end : hd->total_blocks;
    const State s = d.start[i];
    if (s.p == jd::NO_STATE || s.p < w0 * 32u) return;                    // never for a settled chain
    const int4 b = d.base[i];
    const int32_t pred[3] = {b.y, b.z, b.w};
    jd::Counts c;
    (void)jd::run<true>(s, bound, seg_end, words, w0, luts, sel, c, d.coef, b.x, block_end, pred, (const uint8_t*)nat);
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jidctint.c: one 1-D pass of the "islow" inverse DCT on 8 values
__device__ __forceinline__ void islow_pass(const int* in, int* out, int out_shift) {
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * 4433;
    int tmp2 = z1 + z3 * -15137;
    int tmp3 = z1 + z2 * 6270;
    int tmp0 = (in[0] + in[4]) << 13;
    int tmp1 = (in[0] - in[4]) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 = z3 * -16069 + z5; z4 = z4 * -3196 + z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = descale(tmp10 + tmp3, out_shift); out[7] = descale(tmp10 - tmp3, out_shift);
    out[1] = descale(tmp11 + tmp2, out_shift); out[6] = descale(tmp11 - tmp2, out_shift);
    out[2] = descale(tmp12 + tmp1, out_shift); out[5] = descale(tmp12 - tmp1, out_shift);
    out[3] = descale(tmp13 + tmp0, out_shift); out[4] = descale(tmp13 - tmp0, out_shift);
}

// 32 blocks per workgroup, 8 threads per block: dequantise -> column pass -> row pass -> range limit -> u8 plane
__global__ __launch_bounds__(256) void jpeg_idct_kernel(Dev d) {
    constexpr int BS = 72, RS = 9;                   // LDS strides (ints): block, row - conflict-free in both passes
    __shared__ int ws[32 * BS];
    const Header* hd = d.hd;
    const int tid = threadIdx.x, lb = tid >> 3, l = tid & 7;
    const int b = blockIdx.x * 32 + lb;
    const bool valid = b < hd->total_blocks;
    int comp = 0, px = 0, py = 0;
    if (valid) {
        const int mcu = b / hd->bpm, j = b - mcu * hd->bpm;
        comp = hd->blk_comp[j];
        const int jj = j - hd->comp_off[comp];
        const int h = hd->comp_h[comp], v = hd->comp_v[comp];
        const int my = mcu / hd->mx, mxi = mcu - my * hd->mx;
        px = (mxi * h + jj % h) * 8;
        py = (my * v + jj / h) * 8;
        const uint4 raw = *reinterpret_cast<const uint4*>(d.coef + (size_t)b * 64 + l * 8);
        const uint16_t* q = hd->quant[hd->comp_tq[comp]] + l * 8;
        const uint32_t r[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ws[lb * BS + l * RS + 2 * k] = (int)(int16_t)(r[k] & 0xffff) * (int)q[2 * k];
            ws[lb * BS + l * RS + 2 * k + 1] = (int)(int16_t)(r[k] >> 16) * (int)q[2 * k + 1];
        }
    }
    __syncthreads();
    if (valid) {                                      // pass 1: column l
        int in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = ws[lb * BS + r * RS + l];
        islow_pass(in, out, 13 - 2);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[lb * BS + r * RS + l] = out[r];
    }
    __syncthreads();
    if (valid) {                                      // pass 2: row l
        int in[8], out[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) in[k] = ws[lb * BS + l * RS + k];
        islow_pass(in, out, 13 + 2 + 3);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int v = out[k] & 1023;              // RANGE_MASK on the table centred at 128
            const uint32_t u = (uint32_t)(v < 128 ? v + 128 : v < 512 ? 255 : v < 896 ? 0 : v - 896);
            if (k < 4) lo |= u << (8 * k); else hi |= u << (8 * (k - 4));
        }
        uint8_t* row = d.planes + hd->plane_off[comp] + (size_t)(py + l) * hd->plane_pitch[comp] + px;
        *reinterpret_cast<uint2*>(row) = make_uint2(lo, hi);
    }
}

__device__ __forceinline__ void ycc_to_rgb(int y, int cb, int cr, uint8_t* out) {
    cb -= 128; cr -= 128;
    int r = y + ((91881 * cr + 32768) >> 16);
    int g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    int b = y + ((116130 * cb + 32768) >> 16);
    out[0] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
    out[1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g);
    out[2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
}

// MODE 0: 4:4:4, 1: h2v1, 2: h2v2, 3: grayscale.  One thread = one chroma site = FH x FV output pixels.
template <int MODE>
__global__ __launch_bounds__(256) void jpeg_color_kernel(Dev d, uint8_t* __restrict__ rgb) {
    const Header* hd = d.hd;
    const int W = hd->width, H = hd->height;
    constexpr int FH = (MODE == 1 || MODE == 2) ? 2 : 1, FV = MODE == 2 ? 2 : 1;
    const int cw = (W + FH - 1) / FH, ch = (H + FV - 1) / FV;
    const int cx = blockIdx.x * 64 + (threadIdx.x & 63), cy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (cx >= cw || cy >= ch) return;
    const uint8_t* yp = d.planes + hd->plane_off[0];
    const int ypitch = hd->plane_pitch[0];
    if constexpr (MODE == 3) {
        const uint8_t v = yp[(size_t)cy * ypitch + cx];
        uint8_t* o = rgb + ((size_t)cy * W + cx) * 3;
        o[0] = v; o[1] = v; o[2] = v;
        return;
    } else {
    const uint8_t* cbp = d.planes + hd->plane_off[1];
    const uint8_t* crp = d.planes + hd->plane_off[2];
    const int cpitch = hd->plane_pitch[1];
    int cbv[FV][FH], crv[FV][FH];
    if constexpr (MODE == 0) {
        cbv[0][0] = cbp[(size_t)cy * cpitch + cx];
        crv[0][0] = crp[(size_t)cy * cpitch + cx];
    } else {
        const int dw = hd->dw[1], dh = hd->dh[1];
        const bool fancy = dw > 2;                    // jdsample.c: plain replication for very narrow components
        const int xl = cx > 0 ? cx - 1 : 0, xr = cx < dw - 1 ? cx + 1 : dw - 1;
        if constexpr (MODE == 1) {
            const uint8_t* rb = cbp + (size_t)cy * cpitch;
            const uint8_t* rr = crp + (size_t)cy * cpitch;
            const int b0 = rb[cx], r0 = rr[cx];
            if (!fancy) { cbv[0][0] = cbv[0][1] = b0; crv[0][0] = crv[0][1] = r0; }
            else {
                cbv[0][0] = cx == 0 ? b0 : (b0 * 3 + rb[xl] + 1) >> 2;
                cbv[0][1] = cx == dw - 1 ? b0 : (b0 * 3 + rb[xr] + 2) >> 2;
                crv[0][0] = cx == 0 ? r0 : (r0 * 3 + rr[xl] + 1) >> 2;
                crv[0][1] = cx == dw - 1 ? r0 : (r0 * 3 + rr[xr] + 2) >> 2;
            }
        } else {
            const int yu = cy > 0 ? cy - 1 : 0, yd = cy < dh - 1 ? cy + 1 : dh - 1;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const uint8_t* p = pl ? crp : cbp;
                int (*o)[FH] = pl ? crv : cbv;
                const uint8_t* r0 = p + (size_t)cy * cpitch;
                if (!fancy) { const int v = r0[cx]; o[0][0] = o[0][1] = o[1][0] = o[1][1] = v; continue; }
                const uint8_t* ru = p + (size_t)yu * cpitch;
                const uint8_t* rd = p + (size_t)yd * cpitch;
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const uint8_t* ro = v ? rd : ru;
                    const int sl = 3 * r0[xl] + ro[xl], sc = 3 * r0[cx] + ro[cx], sr = 3 * r0[xr] + ro[xr];
                    o[v][0] = cx == 0 ? (sc * 4 + 8) >> 4 : (sc * 3 + sl + 8) >> 4;
                    o[v][1] = cx == dw - 1 ? (sc * 4 + 7) >> 4 : (sc * 3 + sr + 7) >> 4;
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < FV; ++v) {
        const int y = cy * FV + v;
        if (y >= H) continue;
#pragma unroll
        for (int h = 0; h < FH; ++h) {
            const int x = cx * FH + h;
            if (x >= W) continue;
            ycc_to_rgb(yp[(size_t)y * ypitch + x], cbv[v][h], crv[v][h], rgb + ((size_t)y * W + x) * 3);
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------------- scratch contexts
struct Ctx {
    uint8_t* host = nullptr;         // pinned staging blob
    size_t host_bytes = 0;
    uint8_t* dev = nullptr;          // device blob
    size_t dev_bytes = 0;
    uint8_t* work = nullptr;         // nblk | base | flags | coef | planes
    size_t work_bytes = 0;
    int32_t* host_flags = nullptr;   // pinned, 4 ints
    bool busy = false;
};

std::mutex g_mu;
std::vector<Ctx*> g_ctx;

Ctx* acquire() {
    std::lock_guard<std::mutex> lock(g_mu);
    for (Ctx* c : g_ctx)
        if (!c->busy) { c->busy = true; return c; }
    Ctx* c = new Ctx();
    c->busy = true;
    g_ctx.push_back(c);
    return c;
}

// hands the context back; when work has been enqueued on `st` (dirty) the stream is drained first - on an error return queued kernels
// may still read or write the buffers the next holder of the context is about to memcpy into, grow or free
struct Release {
    Ctx* c;
    hipStream_t st = nullptr;
    bool dirty = false;
    ~Release() {
        if (dirty) (void)hipStreamSynchronize(st);
        std::lock_guard<std::mutex> lock(g_mu);
        c->busy = false;
    }
};

int grow(Ctx& c, size_t blob, size_t work) {
    if (!c.host_flags) WT_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.host_flags), 64, hipHostMallocDefault));
    if (blob > c.host_bytes) {
        if (c.host) (void)hipHostFree(c.host);
        c.host = nullptr; c.host_bytes = 0;
        const size_t want = wt::align_up(blob + blob / 4, 1 << 16);
        WT_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.host), want, hipHostMallocDefault));
        c.host_bytes = want;
    }
    if (blob > c.dev_bytes) {
        if (c.dev) (void)hipFree(c.dev);
        c.dev = nullptr; c.dev_bytes = 0;
        const size_t want = wt::align_up(blob + blob / 4, 1 << 16);
        WT_HIP(hipMalloc(reinterpret_cast<void**>(&c.dev), want));
        c.dev_bytes = want;
    }
    if (work > c.work_bytes) {
        if (c.work) (void)hipFree(c.work);
        c.work = nullptr; c.work_bytes = 0;
        const size_t want = wt::align_up(work + work / 4, 1 << 16);
        WT_HIP(hipMalloc(reinterpret_cast<void**>(&c.work), want));
        c.work_bytes = want;
    }
    return WT_OK;
}

thread_local int32_t t_stats[4] = {0, 0, 0, 0};

}  // namespace

extern "C" int wd_jpeg_last_stats(int32_t* out4) {
    if (!out4) return fail("no output");
    for (int k = 0; k < 4; ++k) out4[k] = t_stats[k];
    return WT_OK;
}

extern "C" int wd_jpeg_info(const uint8_t* data, int64_t n, int32_t* width, int32_t* height, int32_t* components,
                            int32_t* h_samp, int32_t* v_samp, int32_t* restart_interval) {
    if (!data || n < 4) return fail("no data");
    Parsed p;
    if (const char* e = jdh::parse(data, (size_t)n, p)) return fail(e);
    if (width) *width = p.hd.width;
    if (height) *height = p.hd.height;
    if (components) *components = p.hd.ncomp;
    if (h_samp) *h_samp = p.hd.hmax;
    if (v_samp) *v_samp = p.hd.vmax;
    if (restart_interval) *restart_interval = p.expected_segments > 1 ? p.hd.ri : 0;
    return WT_OK;
}

extern "C" int wd_jpeg_decode_rgb_u8(const uint8_t* data, int64_t n, uint8_t* rgb, int64_t capacity, int32_t* width,
                                     int32_t* height, int32_t* sync_rounds, void* stream) {
    WT_TRY(wt::ensure_device());
    if (!data || n < 4 || !rgb) return fail("no data / no output buffer");
    Parsed p;
    if (const char* e = jdh::parse(data, (size_t)n, p)) return fail(e);
    Header& hd = p.hd;
    if (width) *width = hd.width;
    if (height) *height = hd.height;
    if ((int64_t)hd.width * hd.height * 3 > capacity) {
        wt::set_error("wd_jpeg_decode_rgb_u8: output needs %lld bytes, capacity %lld", (long long)hd.width * hd.height * 3, (long long)capacity);
        return WT_ERR_CAPACITY;
    }
    // every restart marker takes two bytes of the scan: a header that announces more segments than the file can hold is refused
    // before anything is allocated for it (a few hundred bytes could otherwise reserve gigabytes in the grow-only contexts)
    if ((size_t)(p.expected_segments - 1) > ((size_t)n - p.scan_pos) / 2)
        return fail("restart interval / frame size announce more restart segments than the file holds (truncated or corrupt file)");
    const Layout L = jdh::layout_for((size_t)n, p.scan_pos, p.expected_segments);
    size_t planes_bytes = 0;
    for (int c = 0; c < hd.ncomp; ++c) planes_bytes = (size_t)hd.plane_off[c] + wt::align_up((size_t)hd.plane_pitch[c] * hd.plane_rows[c]);
    size_t woff = 0;
    auto take = [&](size_t b) { const size_t o = woff; woff += wt::align_up(b); return o; };
    const size_t o_cnt = take((size_t)L.max_sub * 16), o_base = take((size_t)L.max_sub * 16), o_flags = take(64);
    const size_t o_cn = take((size_t)L.max_sub * 12);
    const size_t o_zero_end = woff;                      // everything up to here is zeroed per call
    const size_t o_cs = take((size_t)L.max_sub * jd::CAND_MAX * sizeof(State)), o_ce = take((size_t)L.max_sub * jd::CAND_MAX * sizeof(State));
    const size_t o_cc = take((size_t)L.max_sub * jd::CAND_MAX * 16), o_cm = take((size_t)L.max_sub * 8), o_cp = take((size_t)L.max_sub);
    const size_t o_coef = take((size_t)hd.total_blocks * 128), o_planes = take(planes_bytes);
    Ctx* ctx = acquire();
    Release rel{ctx};
    WT_TRY(grow(*ctx, L.total, woff));
    if (const char* e = jdh::unstuff(data, (size_t)n, p.scan_pos, L, p.expected_segments, ctx->host, hd)) return fail(e);
    memcpy(ctx->host + L.header, &hd, sizeof(Header));
    memcpy(ctx->host + L.luts, p.luts, sizeof(p.luts));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t used = L.stream + (size_t)hd.nsub * jd::SUB_BYTES + 16;
    rel.st = st; rel.dirty = true;                     // from here on every return path drains the stream before the context is reused
    WT_HIP(hipMemcpyAsync(ctx->dev, ctx->host, used, hipMemcpyHostToDevice, st));
    WT_HIP(hipMemsetAsync(ctx->work, 0, o_zero_end, st));                                    // counts, bases, flags, candidate counts
    WT_HIP(hipMemsetAsync(ctx->work + o_coef, 0, (size_t)hd.total_blocks * 128, st));        // coefficients
    Dev d;
    d.hd = reinterpret_cast<const Header*>(ctx->dev + L.header);
    d.luts = reinterpret_cast<const HuffLut*>(ctx->dev + L.luts);
    d.seg_first_sub = reinterpret_cast<const uint32_t*>(ctx->dev + L.seg_first_sub);
    d.seg_end_bit = reinterpret_cast<const uint32_t*>(ctx->dev + L.seg_end_bit);
    d.sub_seg = reinterpret_cast<const int32_t*>(ctx->dev + L.sub_seg);
    d.start = reinterpret_cast<State*>(ctx->dev + L.start);
    d.exit = reinterpret_cast<State*>(ctx->dev + L.exit);
    d.stream = reinterpret_cast<const uint32_t*>(ctx->dev + L.stream);
    d.cnt = reinterpret_cast<int4*>(ctx->work + o_cnt);
    d.base = reinterpret_cast<int4*>(ctx->work + o_base);
    d.flags = reinterpret_cast<int32_t*>(ctx->work + o_flags);
    d.cand_n = reinterpret_cast<int32_t*>(ctx->work + o_cn);
    d.cand_s = reinterpret_cast<State*>(ctx->work + o_cs);
    d.cand_e = reinterpret_cast<State*>(ctx->work + o_ce);
    d.cand_c = reinterpret_cast<int4*>(ctx->work + o_cc);
    d.cand_map = reinterpret_cast<unsigned long long*>(ctx->work + o_cm);
    d.cand_pick = ctx->work + o_cp;
    d.coef = reinterpret_cast<int16_t*>(ctx->work + o_coef);
    d.planes = ctx->work + o_planes;
    d.nsub = hd.nsub;
    d.stream_words = hd.nsub * jd::SUB_WORDS + 4;      // 16 bytes of 1-bits follow the last subsequence
    const unsigned sync_grid = (unsigned)((hd.nsub + kSyncThreads - 1) / kSyncThreads);
    int rounds = 0;
    auto tail = [&]() -> int {
        hipLaunchKernelGGL(jpeg_scan_kernel, dim3(1), dim3(1024), 0, st, d);
        hipLaunchKernelGGL(jpeg_write_kernel, dim3(sync_grid), dim3(kSyncThreads), 0, st, d);
        hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((hd.total_blocks + 31) / 32)), dim3(256), 0, st, d);
        const int fh = hd.ncomp == 3 ? hd.hmax / hd.comp_h[1] : 1, fv = hd.ncomp == 3 ? hd.vmax / hd.comp_v[1] : 1;
        const dim3 grid((unsigned)(((hd.width + fh - 1) / fh + 63) / 64), (unsigned)(((hd.height + fv - 1) / fv + 3) / 4));
        if (hd.ncomp == 1) hipLaunchKernelGGL(jpeg_color_kernel<3>, grid, dim3(256), 0, st, d, rgb);
        else if (fh == 1) hipLaunchKernelGGL(jpeg_color_kernel<0>, grid, dim3(256), 0, st, d, rgb);
        else if (fv == 1) hipLaunchKernelGGL(jpeg_color_kernel<1>, grid, dim3(256), 0, st, d, rgb);
        else hipLaunchKernelGGL(jpeg_color_kernel<2>, grid, dim3(256), 0, st, d, rgb);
        WT_HIP(hipGetLastError());
        WT_HIP(hipMemcpyAsync(ctx->host_flags, d.flags, 16, hipMemcpyDeviceToHost, st));
        WT_HIP(hipStreamSynchronize(st));
        return WT_OK;
    };
    // candidate sets: block-index guesses, three closure launches, the true chain by a scan over index maps; then the plain
    // iteration (a no-op pass when the chain is already consistent) and the chain check
    const unsigned cand_grid = (unsigned)((hd.nsub + kCandSubs - 1) / kCandSubs);
    constexpr int kCandLaunches = 4;
    for (int l = 0; l < kCandLaunches; ++l) hipLaunchKernelGGL(jpeg_cand_kernel, dim3(cand_grid), dim3(kSyncThreads), 0, st, d, l);
    const unsigned per_sub_grid = (unsigned)((hd.nsub + 255) / 256);
    hipLaunchKernelGGL(jpeg_map_kernel, dim3(per_sub_grid), dim3(256), 0, st, d, (kCandLaunches - 1) % 3);
    hipLaunchKernelGGL(jpeg_resolve_kernel, dim3(1), dim3(1024), 0, st, d);
    hipLaunchKernelGGL(jpeg_pick_kernel, dim3(per_sub_grid), dim3(256), 0, st, d, (kCandLaunches - 1) % 3);
    for (; rounds < 2; ++rounds) hipLaunchKernelGGL(jpeg_sync_kernel, dim3(sync_grid), dim3(kSyncThreads), 0, st, d);
    WT_TRY(tail());
    while (ctx->host_flags[0]) {                      // chain not settled yet: more rounds, then the tail again
        if (rounds > hd.nsub + 3) return fail("internal: subsequence chain did not settle");
        WT_HIP(hipMemsetAsync(d.flags, 0, 8, st));
        WT_HIP(hipMemsetAsync(d.coef, 0, (size_t)hd.total_blocks * 128, st));
        for (int k = 0; k < 4; ++k, ++rounds) hipLaunchKernelGGL(jpeg_sync_kernel, dim3(sync_grid), dim3(kSyncThreads), 0, st, d);
        WT_TRY(tail());
    }
    if (sync_rounds) *sync_rounds = rounds;
    t_stats[0] = rounds; t_stats[1] = ctx->host_flags[2]; t_stats[2] = ctx->host_flags[3]; t_stats[3] = hd.nsub;
    if (ctx->host_flags[1]) return fail("entropy-coded data ends before the last block (truncated or corrupt file)");
    return WT_OK;
}
