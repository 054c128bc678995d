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

constexpr int SUB_BITS = 1024;                     // bits per subsequence (= 128 bytes = 32 words)
constexpr int SUB_WORDS = SUB_BITS / 32;
constexpr int MAX_BPM = 10;                        // blocks per MCU (T.81 B.2.3)
constexpr uint32_t NO_STATE = 0xFFFFFFFFu;

// decoder table of one Huffman table: 8-bit first-level lookup, canonical search for the longer codes (jdhuff.c)
struct HuffLut {
    uint16_t fast[256];                            // (length << 8) | symbol for codes of <= 8 bits, 0 = longer / invalid
    int32_t maxcode[18];                           // largest code of length l (-1 if none); [17] = sentinel
    int32_t valoff[18];                            // vals index of the first code of length l minus that code
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

// 32 bits of the stream starting at bit p; `words` holds the stream as big-endian 32-bit words relative to word `w0`
template <class WordPtr>
JD_HD uint32_t peek32(WordPtr words, uint32_t w0, uint32_t p) {
    const uint32_t wi = (p >> 5) - w0, sh = p & 31;
    const uint64_t two = ((uint64_t)words[wi] << 32) | words[wi + 1];
    return (uint32_t)((two << sh) >> 32);
}

// one Huffman symbol from the top of `win`; returns the code length (0 = no code: only in padding or out of step)
template <class LutPtr>
JD_HD int symbol(LutPtr lut, uint32_t win, int& sym) {
    const uint32_t f = lut->fast[win >> 24];
    if (f) { sym = (int)(f & 255); return (int)(f >> 8); }
    int l = 9;
    int32_t code = (int32_t)(win >> 23);
    while (l <= 16 && code > lut->maxcode[l]) { ++l; code = (int32_t)(win >> (32 - l)); }
    if (l > 16) return 0;
    sym = lut->vals[(code + lut->valoff[l]) & 255];
    return l;
}

JD_HD int extend(uint32_t r, int s) { return (int)r < (1 << (s - 1)) ? (int)r - (1 << s) + 1 : (int)r; }

// Decode from `st` while the position is inside [.., bound): bound = end of the thread's subsequence or of its restart
// segment, whichever comes first; a symbol that does not fit before `seg_end` ends the segment (padding bits).
// Returns the exit state; `nblk` counts the blocks completed.  WRITE: coefficients (DC still as difference) go to
// coef[(block0 + completed) * 64 + natural position] as long as the block index stays below block_end.
template <bool WRITE, class WordPtr, class LutPtr, class HeaderPtr>
JD_HD State run(State st, uint32_t bound, uint32_t seg_end, WordPtr words, uint32_t w0, LutPtr luts, HeaderPtr hd,
                int& nblk, int16_t* __restrict__ coef, int block0, int block_end) {
    uint32_t p = st.p;
    int blk = (int)(st.bk >> 8), k = (int)(st.bk & 255);
    const int bpm = hd->bpm;
    int n = 0;
    int comp = hd->blk_comp[blk];
    while (p < bound) {
        const uint32_t win = peek32(words, w0, p);
        int sym = 0;
        const int len = symbol(luts + (k == 0 ? hd->comp_dc[comp] : 2 + hd->comp_ac[comp]), win, sym);
        if (len == 0) { p = bound; break; }                                 // no such code: padding, or an out-of-step thread
        const int s = k == 0 ? (sym > 16 ? 16 : sym) : (sym & 15);
        if (p + (uint32_t)(len + s) > seg_end) { p = seg_end; break; }
        const uint32_t extra = s ? (uint32_t)(((uint64_t)win << len) >> (32 - s)) & ((1u << s) - 1u) : 0u;
        p += (uint32_t)(len + s);
        bool done = false;
        if (k == 0) {
            if (WRITE && block0 + n < block_end) coef[(size_t)(block0 + n) * 64] = (int16_t)(s ? extend(extra, s) : 0);
            k = 1;
        } else {
            const int r = sym >> 4;
            if (s) {
                k += r;
                if (WRITE && block0 + n < block_end) coef[(size_t)(block0 + n) * 64 + natural(k)] = (int16_t)extend(extra, s);
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
            blk = blk + 1 == bpm ? 0 : blk + 1;
            comp = hd->blk_comp[blk];
        }
    }
    nblk = n;
    State out;
    out.p = p;
    out.bk = ((uint32_t)blk << 8) | (uint32_t)k;
    return out;
}

}  // namespace jd
