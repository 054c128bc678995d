// TEST INFRASTRUCTURE: runs the GPU JPEG decoder's entropy stage on the CPU, thread by thread - the same jd::run,
// the same staging blob (jpeg_host.h) and the same round structure as jpeg_sync_kernel / jpeg_scan_kernel /
// jpeg_write_kernel in csrc/jpeg_decode.hip - so that the synchronisation scheme can be checked against
// the sequential restatement (oracle/jpeg_ref.py) without a GPU.  Not part of libwaymotrack.so.
//   extern "C" int jpeg_emul_coefficients(data, n, group, coef_out, capacity_blocks, &rounds, &total_blocks, err, errlen)
// `group` = threads per emulated workgroup (256 on the GPU); coefficients come back in scan order (block, 64) int16.
#include <cstdio>
#include <vector>
#include "../../waymo_2d_tracking_amd/csrc/jpeg_host.h"

extern "C" int jpeg_emul_coefficients(const uint8_t* data, long n, int group, int16_t* coef, long capacity_blocks,
                                      int* rounds_out, int* total_blocks, int* geometry, char* err, int errlen) {
    using namespace jd;
    jdh::Parsed p;
    if (const char* e = jdh::parse(data, (size_t)n, p)) { snprintf(err, errlen, "%s", e); return 1; }
    Header& hd = p.hd;
    const jdh::Layout L = jdh::layout_for((size_t)n, p.scan_pos, p.expected_segments);
    std::vector<uint8_t> blob(L.total + 64);
    if (const char* e = jdh::unstuff(data, (size_t)n, p.scan_pos, L, p.expected_segments, blob.data(), hd)) { snprintf(err, errlen, "%s", e); return 1; }
    *total_blocks = hd.total_blocks;
    if (geometry) {
        geometry[0] = hd.width; geometry[1] = hd.height; geometry[2] = hd.ncomp; geometry[3] = hd.bpm; geometry[4] = hd.mx; geometry[5] = hd.my;
        geometry[6] = hd.nsub; geometry[7] = hd.nseg;
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
    int rounds = 0;
    for (;;) {
        // one launch: every workgroup iterates until its own exits stop changing; thread 0 of a group sees the exit its
        // predecessor group had BEFORE this launch (the kernel reads it at entry)
        std::vector<State> before(exits, exits + nsub);
        for (int g0 = 0; g0 < nsub; g0 += group) {
            const int g1 = g0 + group < nsub ? g0 + group : nsub;
            for (;;) {
                std::vector<State> snap(exits + g0, exits + g1);
                bool changed = false;
                for (int i = g0; i < g1; ++i) {
                    State want;
                    if (is_first(i)) { want.p = (uint32_t)i * SUB_BITS; want.bk = 0; }
                    else want = i == g0 ? before[i - 1] : snap[i - 1 - g0];
                    if (same(want, start[i])) continue;
                    start[i] = want;
                    const State e = run<false>(want, bound_of(i), seg_end[sub_seg[i]], words.data(), 0u, p.luts, sel, cnt[i], nullptr, 0, 0, nullptr);
                    if (!same(e, exits[i])) { exits[i] = e; changed = true; }
                }
                if (!changed) break;
            }
        }
        ++rounds;
        if (rounds < 3) continue;
        bool bad = false;
        for (int i = 0; i < nsub; ++i) {
            State want;
            if (is_first(i)) { want.p = (uint32_t)i * SUB_BITS; want.bk = 0; } else want = exits[i - 1];
            bad |= !same(want, start[i]);
        }
        if (!bad) break;
        if (rounds > nsub + 3) { snprintf(err, errlen, "chain did not settle"); return 3; }
    }
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
