// Baseline JPEG on the GPU ("GPU JPEG decode" in front of the fused pre-processing kernel): the
// pieces shared by the kernels (jpeg_decode.hip) and by the host-side emulation used in the CPU tests
// (tests/native/jpeg_sync_emul.cpp compiles this header with g++ and runs the same synchronisation algorithm thread by
// thread).  Replaces `PIL.Image.open(path).convert('RGB')` of the reference's loader (detnet/data/coco.py image read,
// detnet/inference.py:170 ToRGB), i.e. libjpeg-turbo at its defaults: Huffman baseline (T.81 Annex F / jdhuff.c),
// JDCT_ISLOW (jidctint.c), fancy upsampling (jdsample.c), jdcolor.c YCbCr -> RGB.  Bit-exact with PIL by test.
//
// Entropy decoding in parallel (after Weissenberger & Schmidt, "Accelerating JPEG decompression on GPUs", 2021): the
// unstuffed scan is cut into SUBSEQUENCES of 1024 bits, one thread each.  A thread that starts decoding at an arbitrary
// bit with an arbitrary (block-in-MCU, zig-zag index) state produces garbage for a while and then falls into step with
// the true decoder (Huffman codes self-synchronise); so every thread decodes its subsequence from the EXIT state of its
// predecessor, repeatedly, until no exit state changes any more - at that point thread i's start state is exactly the
// state of the sequential decoder at that bit (induction from the first subsequence of each restart segment, whose start
// is known).  A last pass decodes once more from the settled start states and writes the coefficients.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define JD_HD __host__ __device__ __forceinline__
#else
#define JD_HD inline
#endif

namespace jd {

#ifndef JD_SUB_BITS
#define JD_SUB_BITS 1024
#endif
constexpr int SUB_BITS = JD_SUB_BITS;              // bits per subsequence (1024 = 128 bytes = 32 words)
constexpr int SUB_BYTES = SUB_BITS / 8;
constexpr int SUB_WORDS = SUB_BITS / 32;
constexpr int MAX_BPM = 10;                        // blocks per MCU (T.81 B.2.3)
constexpr uint32_t NO_STATE = 0xFFFFFFFFu;

// decoder table of one Huffman table.  Codes of <= FAST_BITS bits: one lookup with the next FAST_BITS bits (12 bits: 8 KB per
// table in LDS; longer codes are ~0.3 % of the symbols, so that a 64-lane wave rarely has to walk the second path).  Longer
// codes: canonical codes are ordered, so with limit[l] = (first code value after the codes of length l) left-justified to 16
// bits - a non-decreasing sequence - the length of the code at the top of a 16-bit window w is
// FAST_BITS + 1 + #{l > FAST_BITS : w >= limit[l]}: independent compares instead of jdhuff.c's bit-by-bit maxcode loop (a chain of
// dependent reads on a GPU).
constexpr int FAST_BITS = 12;
constexpr int SLOW_LENS = 16 - FAST_BITS;
struct alignas(16) HuffLut {
    uint16_t fast[1 << FAST_BITS];                 // (length << 8) | symbol, 0 = longer code / no code
    uint32_t limit[SLOW_LENS];                     // lengths FAST_BITS + 1 .. 16
    uint32_t valoff[SLOW_LENS];                    // (vals index of the first code of length l - that code) mod 2^16
    uint8_t vals[256];
};

struct Header {
    int32_t width, height, ncomp, bpm;             // bpm = blocks per MCU
    int32_t mx, my;                                // MCUs per row / column
    int32_t ri;                                    // MCUs per restart segment (mx * my without DRI)
    int32_t nseg, nsub, total_blocks;
    int32_t hmax, vmax;
    int32_t comp_h[4], comp_v[4], comp_tq[4], comp_dc[4], comp_ac[4], comp_off[4], comp_nblk[4];
    int32_t plane_off[4], plane_pitch[4], plane_rows[4];   // u8 component planes (whole MCUs)
    int32_t dw[4], dh[4];                          // downsampled_width / height of libjpeg (what upsampling sees)
    uint8_t blk_comp[16];                          // block-in-MCU -> component
    uint16_t quant[4][64];                         // natural order
};

// natural-order position of zig-zag index k (entries past 63 guard corrupt runs, like jpeg_natural_order)
JD_HD int natural(int k) {
    constexpr uint8_t t[80] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7,
                               14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46,
                               53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
    return t[k];
}

// state of the sequential decoder between two symbols
struct State {
    uint32_t p;                                    // absolute bit position in the (padded) stream
    uint32_t bk;                                   // (block in MCU << 8) | next zig-zag index (0 = DC symbol next)
};
JD_HD bool same(const State& a, const State& b) { return a.p == b.p && a.bk == b.bk; }

// one Huffman symbol from the top of `win`; returns the code length (0 = no code: only in padding or out of step)
template <class LutPtr>
JD_HD int symbol(LutPtr lut, uint32_t win, int& sym) {
    const uint32_t f = lut->fast[win >> (32 - FAST_BITS)];
    if (f) { sym = (int)(f & 255); return (int)(f >> 8); }
    const uint32_t w16 = win >> 16;
    uint32_t vo[SLOW_LENS];
    int l = FAST_BITS + 1;
#pragma unroll
    for (int j = 0; j < SLOW_LENS; ++j) {
        l += w16 >= lut->limit[j] ? 1 : 0;
        vo[j] = lut->valoff[j];
    }
    if (l > 16) return 0;
    uint32_t v = vo[0];
#pragma unroll
    for (int j = 1; j < SLOW_LENS; ++j) v = l - (FAST_BITS + 1) == j ? vo[j] : v;
    sym = lut->vals[((win >> (32 - l)) + v) & 255];
    return l;
}

JD_HD int extend(uint32_t r, int s) { return (int)r < (1 << (s - 1)) ? (int)r - (1 << s) + 1 : (int)r; }

// table selection without memory reads: bit b of dc / ac = Huffman table of block-in-MCU b, 2 bits per block = component
struct Sel {
    uint32_t dc, ac, comp;
    int bpm;
};
template <class HeaderPtr>
JD_HD Sel make_sel(HeaderPtr hd) {
    Sel s;
    s.dc = s.ac = s.comp = 0;
    s.bpm = hd->bpm;
    for (int b = 0; b < hd->bpm; ++b) {
        const int c = hd->blk_comp[b];
        s.dc |= (uint32_t)hd->comp_dc[c] << b;
        s.ac |= (uint32_t)hd->comp_ac[c] << b;
        s.comp |= (uint32_t)c << (2 * b);
    }
    return s;
}

// what a subsequence contributes to the sequential decoder's running values: blocks completed, DC differences per component
struct Counts {
    int32_t n, dc[3];
};

// Decode from `st` while the position is inside [.., bound): bound = end of the thread's subsequence or of its restart
// segment, whichever comes first; a symbol that does not fit before `seg_end` ends the segment (padding bits).
// Returns the exit state; `cnt` = blocks completed and the sum of the DC differences decoded, per component.
// WRITE: coefficients go to coef[(block0 + completed) * 64 + natural position] as long as the block index stays below
// block_end, DC as the running prediction that starts at pred0[component].
// `nat` (WRITE): the zig-zag -> natural-order table in fast memory (LDS on the GPU), 80 entries; natural() otherwise.
// The bit window lives in registers (64 bits, refilled one word at a time with the next word already in flight): one
// dependent LDS access per symbol - the table lookup.
template <bool WRITE, class WordPtr, class LutPtr, class NatPtr = const uint8_t*>
JD_HD State run(State st, uint32_t bound, uint32_t seg_end, WordPtr words, uint32_t w0, LutPtr luts, const Sel sel,
                Counts& cnt, int16_t* __restrict__ coef, int block0, int block_end, const int32_t* pred0, NatPtr nat = nullptr) {
    uint32_t p = st.p;
    int blk = (int)(st.bk >> 8), k = (int)(st.bk & 255);
    int n = 0, d0 = 0, d1 = 0, d2 = 0;
    int pr0 = 0, pr1 = 0, pr2 = 0;
    if (WRITE) { pr0 = pred0[0]; pr1 = pred0[1]; pr2 = pred0[2]; }
    uint32_t wi = (p >> 5) - w0;
    uint32_t used = p & 31;
    uint32_t hi = words[wi], lo = words[wi + 1], ahead = words[wi + 2];      // 64-bit window + the next word in flight
    while (p < bound) {
        const uint32_t win = used ? (hi << used) | (lo >> (32 - used)) : hi;
        int sym = 0;
        const uint32_t tsel = k == 0 ? ((sel.dc >> blk) & 1u) : 2u + ((sel.ac >> blk) & 1u);
        const int len = symbol(luts + tsel, win, sym);
        if (len == 0) { p = bound; break; }                                 // no such code: padding, or an out-of-step thread
        const int s = k == 0 ? (sym > 16 ? 16 : sym) : (sym & 15);
        if (p + (uint32_t)(len + s) > seg_end) { p = seg_end; break; }
        const uint32_t extra = s ? (win << len) >> (32 - s) : 0u;          // len + s <= 32
        p += (uint32_t)(len + s);
        used += (uint32_t)(len + s);
        if (used >= 32) {
            used -= 32;
            ++wi;
            hi = lo;
            lo = ahead;
            ahead = words[wi + 2];
        }
        bool done = false;
        if (k == 0) {
            const int diff = s ? extend(extra, s) : 0;
            const uint32_t comp = (sel.comp >> (2 * blk)) & 3u;
            d0 += comp == 0 ? diff : 0;
            d1 += comp == 1 ? diff : 0;
            d2 += comp == 2 ? diff : 0;
            if (WRITE) {
                pr0 += comp == 0 ? diff : 0;
                pr1 += comp == 1 ? diff : 0;
                pr2 += comp == 2 ? diff : 0;
                if (block0 + n < block_end) coef[(size_t)(block0 + n) * 64] = (int16_t)(comp == 0 ? pr0 : comp == 1 ? pr1 : pr2);
            }
            k = 1;
        } else {
            const int r = sym >> 4;
            if (s) {
                k += r;
                if (WRITE && block0 + n < block_end) coef[(size_t)(block0 + n) * 64 + (nat ? nat[k] : natural(k))] = (int16_t)extend(extra, s);
                ++k;
            } else if (r == 15) {
                k += 16;
            } else {
                done = true;
            }
        }
        if (done || k >= 64) {
            k = 0;
            ++n;
            blk = blk + 1 == sel.bpm ? 0 : blk + 1;
        }
    }
    cnt.n = n; cnt.dc[0] = d0; cnt.dc[1] = d1; cnt.dc[2] = d2;
    State out;
    out.p = p;
    out.bk = ((uint32_t)blk << 8) | (uint32_t)k;
    return out;
}

// ---- candidate sets (jpeg_cand_kernel / jpeg_resolve_kernel) -----------------------------------------------------------------
// The iteration "decode from the predecessor's exit until nothing changes" moves the truth one subsequence per pass through every
// stretch of subsequences that did not fall into step on their own - 12 to 14 passes on a 4:2:0 frame, because a stream needs the
// block-in-MCU index (Y Y Y Y Cb Cr use different tables) to line up as well, and a single guess achieves that in 53 % of the
// subsequences only.  But the true exit is AMONG the exits of the guesses "block index h at my first bit" in 97 % of them.  So
// every subsequence keeps a small set of candidate (start state -> exit state) decodes: first the block-index guesses, then, launch
// by launch, every exit of its predecessor's candidates that it has not started from yet.  After three such launches the sets are
// closed on a photo-like frame (every candidate exit of i - 1 is a candidate start of i), the true chain is a path through them,
// and which candidate it is in every subsequence follows from a segmented scan over index maps (composition of maps is
// associative).  Streams that do not fall into step inside one subsequence (quality-100 noise) leave the chain unresolved; the
// plain iteration and the chain check behind it remain the judge in every case.
constexpr int CAND_MAX = 12;                       // candidates per subsequence
constexpr int CAND_SLOTS = 8;                      // threads per subsequence in jpeg_cand_kernel (>= blocks per MCU of the supported files)
constexpr uint32_t CAND_NONE = 15;
JD_HD uint64_t map_identity() { return 0xFEDCBA9876543210ull; }
JD_HD uint32_t map_at(uint64_t m, uint32_t x) { return (uint32_t)(m >> (4 * x)) & 15u; }
JD_HD uint64_t map_compose(uint64_t later, uint64_t earlier) {               // x -> later[earlier[x]]; CAND_NONE stays CAND_NONE
    uint64_t r = 0;
    for (int x = 0; x < 16; ++x) r |= (uint64_t)map_at(later, map_at(earlier, (uint32_t)x)) << (4 * x);
    return r;
}

}  // namespace jd
