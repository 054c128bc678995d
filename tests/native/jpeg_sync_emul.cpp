// TEST INFRASTRUCTURE: runs the GPU JPEG decoder's entropy stage on the CPU, thread by thread - the same jd::run,
// the same staging blob (jpeg_host.h) and the same round structure as jpeg_sync_kernel / jpeg_scan_kernel /
// jpeg_write_kernel in csrc/jpeg_decode.hip - so that the synchronisation scheme can be checked against
// the sequential restatement (oracle/jpeg_ref.py) without a GPU.  Not part of libwaymotrack.so.
//   extern "C" int jpeg_emul_coefficients(data, n, group, coef_out, capacity_blocks, &rounds, &total_blocks, err, errlen)
// `group` = threads per emulated workgroup (256 on the GPU; negative = plain iteration without the candidate sets); coefficients come
// back in scan order (block, 64) int16; geometry[8] = most iterations of a workgroup inside one launch, [9] = subsequence decodes.
#include <cstdio>
#include <vector>
#include "../../waymo_2d_tracking_amd/csrc/jpeg_host.h"

extern "C" int jpeg_emul_coefficients(const uint8_t* data, long n, int group_in, int16_t* coef, long capacity_blocks,
                                      int* rounds_out, int* total_blocks, int* geometry, char* err, int errlen) {
    using namespace jd;
    const int group = group_in < 0 ? -group_in : group_in;      // negative: plain iteration only (no candidate sets)
    const bool use_candidates = group_in > 0;
    jdh::Parsed p;
    if (const char* e = jdh::parse(data, (size_t)n, p)) { snprintf(err, errlen, "%s", e); return 1; }
    Header& hd = p.hd;
    const jdh::Layout L = jdh::layout_for((size_t)n, p.scan_pos, p.expected_segments);
    std::vector<uint8_t> blob(L.total + 64);
    if (const char* e = jdh::unstuff(data, (size_t)n, p.scan_pos, L, p.expected_segments, blob.data(), hd)) { snprintf(err, errlen, "%s", e); return 1; }
    *total_blocks = hd.total_blocks;
    if (geometry) {
        geometry[0] = hd.width; geometry[1] = hd.height; geometry[2] = hd.ncomp; geometry[3] = hd.bpm; geometry[4] = hd.mx; geometry[5] = hd.my;
        geometry[6] = hd.nsub; geometry[7] = hd.nseg; geometry[8] = geometry[9] = 0;
    }
    if (hd.total_blocks > capacity_blocks) { snprintf(err, errlen, "capacity"); return 2; }
    const uint32_t* seg_first = reinterpret_cast<const uint32_t*>(blob.data() + L.seg_first_sub);
    const uint32_t* seg_end = reinterpret_cast<const uint32_t*>(blob.data() + L.seg_end_bit);
    const int32_t* sub_seg = reinterpret_cast<const int32_t*>(blob.data() + L.sub_seg);
    State* start = reinterpret_cast<State*>(blob.data() + L.start);
    State* exits = reinterpret_cast<State*>(blob.data() + L.exit);
    const uint8_t* bytes = blob.data() + L.stream;
    const int nsub = hd.nsub;
    std::vector<uint32_t> words((size_t)nsub * SUB_WORDS + 4);
    for (size_t w = 0; w < words.size(); ++w) {
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) { const size_t bi = w * 4 + k; v = (v << 8) | (bi < (size_t)nsub * SUB_BYTES + 16 ? bytes[bi] : 0xFF); }
        words[w] = v;
    }
    std::vector<Counts> cnt(nsub, Counts{0, {0, 0, 0}}), base(nsub, Counts{0, {0, 0, 0}});
    const Sel sel = make_sel(&hd);
    auto bound_of = [&](int i) { const uint32_t b = (uint32_t)(i + 1) * SUB_BITS, e = seg_end[sub_seg[i]]; return b < e ? b : e; };
    auto is_first = [&](int i) { return seg_first[sub_seg[i]] == (uint32_t)i; };
    int rounds = 0, max_iters_seen = 0;
    long decodes = 0;
    // one launch of jpeg_sync_kernel: every workgroup iterates until its own exits stop changing; thread 0 of a group sees the exit
    // its predecessor group had BEFORE this launch (the kernel reads it at entry)
    auto sync_launch = [&]() {
        std::vector<State> before(exits, exits + nsub);
        for (int g0 = 0; g0 < nsub; g0 += group) {
            const int g1 = g0 + group < nsub ? g0 + group : nsub;
            for (int it = 1;; ++it) {
                std::vector<State> snap(exits + g0, exits + g1);
                bool changed = false;
                for (int i = g0; i < g1; ++i) {
                    State want;
                    if (is_first(i)) { want.p = (uint32_t)i * SUB_BITS; want.bk = 0; }
                    else want = i == g0 ? before[i - 1] : snap[i - 1 - g0];
                    if (same(want, start[i])) continue;
                    start[i] = want;
                    ++decodes;
                    const State e = run<false>(want, bound_of(i), seg_end[sub_seg[i]], words.data(), 0u, p.luts, sel, cnt[i], nullptr, 0, 0, nullptr);
                    if (!same(e, exits[i])) { exits[i] = e; changed = true; }
                }
                max_iters_seen = it > max_iters_seen ? it : max_iters_seen;
                if (!changed) break;
            }
        }
        ++rounds;
    };
    // jpeg_cand_kernel x 4 + jpeg_resolve_kernel (candidate sets, jpeg_core.h)
    if (use_candidates) {
        struct Rec { State s, e; Counts c; };
        std::vector<std::vector<Rec>> cand(nsub);
        auto decode_from = [&](int i, State st) {
            Rec r;
            r.s = st;
            r.c = Counts{0, {0, 0, 0}};
            r.e = run<false>(st, bound_of(i), seg_end[sub_seg[i]], words.data(), 0u, p.luts, sel, r.c, nullptr, 0, 0, nullptr);
            ++decodes;
            return r;
        };
        for (int i = 0; i < nsub; ++i) {                                   // launch 0
            const int nh = is_first(i) ? 1 : hd.bpm;
            for (int h = 0; h < nh && h < CAND_SLOTS; ++h) cand[i].push_back(decode_from(i, State{(uint32_t)i * SUB_BITS, is_first(i) ? 0u : (uint32_t)h << 8}));
        }
        for (int l = 1; l < 4; ++l) {
            std::vector<size_t> n_old(nsub);
            for (int i = 0; i < nsub; ++i) n_old[i] = cand[i].size();
            for (int i = nsub - 1; i >= 1; --i) {                          // in place: only entries below the predecessor's old count are read
                if (is_first(i)) continue;
                int fresh = 0;
                const size_t mine = n_old[i];
                for (size_t a = 0; a < n_old[i - 1]; ++a) {
                    const State e = cand[i - 1][a].e;
                    bool known = false;
                    for (size_t b2 = 0; b2 < mine; ++b2) known |= same(cand[i][b2].s, e);
                    for (size_t b2 = 0; b2 < a; ++b2) known |= same(cand[i - 1][b2].e, e);
                    if (known) continue;
                    if (fresh < CAND_SLOTS && mine + fresh < (size_t)CAND_MAX && e.p <= bound_of(i)) cand[i].push_back(decode_from(i, e));
                    else if (fresh < CAND_SLOTS && mine + fresh < (size_t)CAND_MAX) cand[i].push_back(Rec{State{NO_STATE, NO_STATE}, State{NO_STATE, NO_STATE}, Counts{0, {0, 0, 0}}});
                    ++fresh;
                }
                if (cand[i].size() > (size_t)CAND_MAX) cand[i].resize(CAND_MAX);
            }
        }
        uint32_t b = 0;                                                    // jpeg_resolve_kernel, walked in order
        for (int i = 0; i < nsub; ++i) {
            if (is_first(i)) b = 0;
            else if (b != CAND_NONE) {
                const State e = cand[i - 1][b].e;
                uint32_t to = CAND_NONE;
                for (size_t k = 0; k < cand[i].size(); ++k) if (same(cand[i][k].s, e)) to = (uint32_t)k;
                b = to;
            }
            if (b != CAND_NONE && b < cand[i].size() && cand[i][b].s.p != NO_STATE) { start[i] = cand[i][b].s; exits[i] = cand[i][b].e; cnt[i] = cand[i][b].c; }
        }
        if (map_at(map_compose(map_identity(), map_identity()), 7) != 7 || map_at(map_compose(0xF000000000000000ull | 0x21ull, 0xF000000000000000ull | 0x10ull), 1) != 2 ||
            map_at(map_compose(0xF000000000000000ull | 0x21ull, 0xF000000000000000ull | 0x10ull), 15) != CAND_NONE) {
            snprintf(err, errlen, "map_compose"); return 5;
        }
    }
    for (;;) {
        sync_launch();
        sync_launch();
        bool bad = false;
        for (int i = 0; i < nsub; ++i) {
            State want;
            if (is_first(i)) { want.p = (uint32_t)i * SUB_BITS; want.bk = 0; } else want = exits[i - 1];
            bad |= !same(want, start[i]);
        }
        if (!bad) break;
        sync_launch(); sync_launch();
        if (rounds > nsub + 3) { snprintf(err, errlen, "chain did not settle"); return 3; }
    }
    if (geometry) { geometry[8] = max_iters_seen; geometry[9] = (int)(decodes > 0x7fffffff ? 0x7fffffff : decodes); }
    *rounds_out = rounds;
    int shortfall = 0;
    for (int s = 0; s < hd.nseg; ++s) {                                   // jpeg_scan_kernel: running values per segment
        Counts acc{0, {0, 0, 0}};
        const int b0 = s * hd.ri * hd.bpm;
        for (uint32_t i = seg_first[s]; i < seg_first[s + 1]; ++i) {
            base[i] = acc;
            base[i].n += b0;
            acc.n += cnt[i].n;
            for (int c = 0; c < 3; ++c) acc.dc[c] += cnt[i].dc[c];
        }
        int expect = hd.total_blocks - b0;
        expect = expect < hd.ri * hd.bpm ? expect : hd.ri * hd.bpm;
        shortfall |= acc.n < expect;
    }
    std::fill(coef, coef + (size_t)hd.total_blocks * 64, (int16_t)0);
    for (int i = 0; i < nsub; ++i) {                                      // jpeg_write_kernel
        const int s = sub_seg[i];
        int block_end = (s + 1) * hd.ri * hd.bpm;
        block_end = block_end < hd.total_blocks ? block_end : hd.total_blocks;
        Counts c;
        (void)run<true>(start[i], bound_of(i), seg_end[s], words.data(), 0u, p.luts, sel, c, coef, base[i].n, block_end, base[i].dc);
    }
    if (shortfall) { snprintf(err, errlen, "segment came up short"); return 4; }
    return 0;
}
