// Host side of the GPU JPEG decoder (plain C++, no HIP): marker parse, decoder tables, staging-blob layout, byte
// unstuffing into restart segments.  Shared by jpeg_decode.hip and the CPU emulation of the tests.  Functions return
// nullptr or the error text.
#pragma once
#include <cstddef>
#include "jpeg_core.h"

namespace jdh {

using jd::Header;
using jd::HuffLut;
using jd::State;

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
constexpr size_t SB = jd::SUB_BYTES;

struct Parsed {
    Header hd;
    HuffLut luts[4];                 // dc0, dc1, ac0, ac1
    size_t scan_pos = 0;             // first byte of entropy-coded data
    int expected_segments = 0;
};

inline void derive(const uint8_t* bits, const uint8_t* vals, int nvals, HuffLut& lut) {
    constexpr int FB = jd::FAST_BITS;
    memset(&lut, 0, sizeof(lut));
    memcpy(lut.vals, vals, (size_t)nvals);
    uint32_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        if (l > FB) lut.valoff[l - FB - 1] = ((uint32_t)k - code) & 0xffffu;
        for (int i = 0; i < bits[l - 1]; ++i, ++k, ++code) {
            if (l <= FB) {
                const uint32_t lo = code << (FB - l);
                for (uint32_t j = 0; j < (1u << (FB - l)); ++j)
                    if (lo + j < (1u << FB)) lut.fast[lo + j] = (uint16_t)((l << 8) | vals[k]);
            }
        }
        if (l > FB) lut.limit[l - FB - 1] = code << (16 - l);            // an over-full table (corrupt DHT) only makes codes unreachable
        code <<= 1;
    }
}

inline const char* parse(const uint8_t* d, size_t n, Parsed& out) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return ("not a JPEG file (no SOI)");
    memset(&out.hd, 0, sizeof(out.hd));
    Header& hd = out.hd;
    bool have_jfif = false;
    bool have_sof = false, have_q[4] = {false, false, false, false}, have_h[4] = {false, false, false, false};
    int comp_id[4] = {0, 0, 0, 0};
    int dri = 0, adobe_transform = -1;
    size_t pos = 2;
    for (;;) {
        while (pos < n && d[pos] != 0xFF) ++pos;
        while (pos < n && d[pos] == 0xFF) ++pos;
        if (pos >= n) return ("truncated before the scan");
        const int m = d[pos++];
        if (m == 0xD9) return ("no scan in the file");
        if ((m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (pos + 2 > n) return ("truncated marker segment");
        const size_t len = ((size_t)d[pos] << 8) | d[pos + 1];
        if (len < 2 || pos + len > n) return ("bad marker segment length");
        const uint8_t* s = d + pos + 2;
        const size_t sl = len - 2;
        if (m == 0xDB) {
            size_t i = 0;
            while (i < sl) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > sl) return ("bad DQT");
                for (int k = 0; k < 64; ++k) {
                    const int v = pq ? ((s[i] << 8) | s[i + 1]) : s[i];
                    i += pq ? 2 : 1;
                    hd.quant[tq][jd::natural(k)] = (uint16_t)v;
                }
                have_q[tq] = true;
            }
        } else if (m == 0xC4) {
            size_t i = 0;
            while (i < sl) {
                if (i + 17 > sl) return ("bad DHT");
                const int tc = s[i] >> 4, th = s[i] & 15;
                int nv = 0;
                for (int k = 0; k < 16; ++k) nv += s[i + 1 + k];
                if (tc > 1 || th > 1) return ("unsupported: Huffman table id > 1 (not baseline)");
                if (nv > 256 || i + 17 + (size_t)nv > sl) return ("bad DHT");
                derive(s + i + 1, s + i + 17, nv, out.luts[tc * 2 + th]);
                have_h[tc * 2 + th] = true;
                i += 17 + (size_t)nv;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6) return ("bad SOF");
            if (s[0] != 8) return ("unsupported: sample precision is not 8 bits");
            hd.height = (s[1] << 8) | s[2];
            hd.width = (s[3] << 8) | s[4];
            hd.ncomp = s[5];
            if (hd.width < 1 || hd.height < 1) return ("unsupported: zero image dimension (DNL)");
            if (hd.ncomp != 1 && hd.ncomp != 3) return ("unsupported: component count (only grayscale and YCbCr)");
            if (sl < 6 + 3 * (size_t)hd.ncomp) return ("bad SOF");
            for (int c = 0; c < hd.ncomp; ++c) {
                comp_id[c] = s[6 + 3 * c];
                hd.comp_h[c] = s[7 + 3 * c] >> 4;
                hd.comp_v[c] = s[7 + 3 * c] & 15;
                hd.comp_tq[c] = s[8 + 3 * c];
                if (hd.comp_h[c] < 1 || hd.comp_h[c] > 4 || hd.comp_v[c] < 1 || hd.comp_v[c] > 4 || hd.comp_tq[c] > 3) return ("bad SOF component");
            }
            have_sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC8 && m != 0xCC) {
            return ("unsupported: progressive / lossless / arithmetic-coded JPEG (only baseline Huffman is decoded on the GPU)");
        } else if (m == 0xE0) {
            if (sl >= 5 && memcmp(s, "JFIF", 5) == 0) have_jfif = true;
        } else if (m == 0xEE) {
            // Adobe APP14: transform 0 with three components = the samples are RGB, not YCbCr (jdapimin.c default_decompress_parms)
            if (sl >= 12 && memcmp(s, "Adobe", 5) == 0) adobe_transform = s[11];
        } else if (m == 0xDD) {
            if (sl < 2) return ("bad DRI");
            dri = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!have_sof) return ("SOS before SOF");
            if (sl < 1 || s[0] != hd.ncomp || sl < 1 + 2 * (size_t)hd.ncomp + 3) return ("unsupported: multi-scan (non-interleaved) file");
            for (int c = 0; c < hd.ncomp; ++c) {
                int ci = -1;
                for (int k = 0; k < hd.ncomp; ++k) if (comp_id[k] == s[1 + 2 * c]) ci = k;
                if (ci != c) return ("unsupported: scan component order differs from the frame header");
                hd.comp_dc[c] = s[2 + 2 * c] >> 4;
                hd.comp_ac[c] = s[2 + 2 * c] & 15;
                if (hd.comp_dc[c] > 1 || hd.comp_ac[c] > 1 || !have_h[hd.comp_dc[c]] || !have_h[2 + hd.comp_ac[c]]) return ("scan refers to a missing Huffman table");
                if (!have_q[hd.comp_tq[c]]) return ("frame refers to a missing quantisation table");
            }
            pos += len;
            break;
        }
        pos += len;
    }
    out.scan_pos = pos;
    if (hd.ncomp == 1) hd.comp_h[0] = hd.comp_v[0] = 1;              // a single-component scan is never interleaved (T.81 A.2.2)
    hd.hmax = hd.vmax = 1;
    for (int c = 0; c < hd.ncomp; ++c) {
        hd.hmax = hd.comp_h[c] > hd.hmax ? hd.comp_h[c] : hd.hmax;
        hd.vmax = hd.comp_v[c] > hd.vmax ? hd.comp_v[c] : hd.vmax;
    }
    if (hd.ncomp == 3) {
        // libjpeg's colour-space guess: Adobe transform 0, or (without JFIF / Adobe markers) component ids 'R' 'G' 'B'
        if (adobe_transform == 0 || (adobe_transform < 0 && !have_jfif && comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B'))
            return ("unsupported: RGB-coded JPEG (no YCbCr transform)");
        if (hd.comp_h[0] != hd.hmax || hd.comp_v[0] != hd.vmax || hd.comp_h[1] != hd.comp_h[2] || hd.comp_v[1] != hd.comp_v[2])
            return ("unsupported: luma is subsampled or the chroma planes differ");
        const int fh = hd.hmax / hd.comp_h[1], fv = hd.vmax / hd.comp_v[1];
        if (hd.hmax % hd.comp_h[1] || hd.vmax % hd.comp_v[1] || !((fh == 1 && fv == 1) || (fh == 2 && fv == 1) || (fh == 2 && fv == 2)))
            return ("unsupported: chroma subsampling other than 4:4:4, 4:2:2 (h2v1), 4:2:0 (h2v2)");
    }
    hd.mx = (hd.width + 8 * hd.hmax - 1) / (8 * hd.hmax);
    hd.my = (hd.height + 8 * hd.vmax - 1) / (8 * hd.vmax);
    hd.bpm = 0;
    int plane = 0;
    for (int c = 0; c < hd.ncomp; ++c) {
        hd.comp_off[c] = hd.bpm;
        hd.comp_nblk[c] = hd.comp_h[c] * hd.comp_v[c];
        for (int j = 0; j < hd.comp_nblk[c]; ++j) {
            // launch 0 of the candidate kernel starts one decode per block-in-MCU index with CAND_SLOTS threads per subsequence
            if (hd.bpm >= jd::MAX_BPM || hd.bpm >= jd::CAND_SLOTS) return ("unsupported: more than 8 blocks per MCU");
            hd.blk_comp[hd.bpm++] = (uint8_t)c;
        }
        hd.plane_off[c] = plane;
        hd.plane_pitch[c] = hd.mx * hd.comp_h[c] * 8;
        hd.plane_rows[c] = hd.my * hd.comp_v[c] * 8;
        plane += (int)align_up((size_t)hd.plane_pitch[c] * hd.plane_rows[c]);
        hd.dw[c] = (hd.width * hd.comp_h[c] + hd.hmax - 1) / hd.hmax;
        hd.dh[c] = (hd.height * hd.comp_v[c] + hd.vmax - 1) / hd.vmax;
    }
    const long mcus = (long)hd.mx * hd.my;
    if (mcus * hd.bpm > (1L << 23)) return ("image too large (more than 2^23 blocks)");
    hd.ri = dri > 0 && dri < mcus ? dri : (int)mcus;
    hd.total_blocks = (int)(mcus * hd.bpm);
    out.expected_segments = (int)((mcus + hd.ri - 1) / hd.ri);
    return nullptr;
}

// staging blob layout (host pinned and device, same offsets)
struct Layout {
    size_t header, luts, seg_first_sub, seg_end_bit, sub_seg, start, exit, stream, total;
    int max_sub;
};

inline Layout layout_for(size_t file_bytes, size_t scan_pos, int nseg) {
    Layout L;
    const size_t scan_max = file_bytes - scan_pos;
    L.max_sub = (int)(scan_max / SB + (size_t)nseg + 2);
    size_t off = 0;
    auto take = [&](size_t b) { const size_t o = off; off += align_up(b); return o; };
    L.header = take(sizeof(Header));
    L.luts = take(4 * sizeof(HuffLut));
    L.seg_first_sub = take(((size_t)nseg + 1) * 4);
    L.seg_end_bit = take((size_t)nseg * 4);
    L.sub_seg = take((size_t)L.max_sub * 4);
    L.start = take((size_t)L.max_sub * sizeof(State));
    L.exit = take((size_t)L.max_sub * sizeof(State));
    L.stream = take((size_t)L.max_sub * SB + 16);
    L.total = off;
    return L;
}

// Unstuff the entropy-coded bytes into `stream`, one subsequence-aligned run per restart segment, padded with 1-bits.
inline const char* unstuff(const uint8_t* d, size_t n, size_t pos, const Layout& L, int nseg_expected, uint8_t* blob, Header& hd) {
    uint8_t* stream = blob + L.stream;
    uint32_t* seg_first = reinterpret_cast<uint32_t*>(blob + L.seg_first_sub);
    uint32_t* seg_end = reinterpret_cast<uint32_t*>(blob + L.seg_end_bit);
    const size_t cap = (size_t)L.max_sub * SB;
    size_t w = 0;                     // write offset in stream
    int seg = 0;
    seg_first[0] = 0;
    bool end = false;
    while (!end) {
        const uint8_t* q = pos < n ? (const uint8_t*)memchr(d + pos, 0xFF, n - pos) : nullptr;
        const size_t run = q ? (size_t)(q - (d + pos)) : n - pos;
        if (w + run + 1 > cap) return ("internal: scan staging overflow");
        memcpy(stream + w, d + pos, run);
        w += run;
        pos += run;
        if (!q || pos + 1 >= n) { end = true; break; }
        const int nb = d[pos + 1];
        if (nb == 0) { stream[w++] = 0xFF; pos += 2; continue; }
        if (nb == 0xFF) { pos += 1; continue; }                           // fill byte
        if (nb >= 0xD0 && nb <= 0xD7) {
            // a restart segment without a single byte (two markers in a row): its blocks would stay zero unnoticed
            if (w == (size_t)seg_first[seg] * SB) return ("empty restart segment (truncated or corrupt file)");
            seg_end[seg] = (uint32_t)(w * 8);
            const size_t padded = (w + SB - 1) / SB * SB;
            memset(stream + w, 0xFF, padded - w);
            w = padded;
            ++seg;
            if (seg >= nseg_expected) return ("more restart markers than the restart interval allows");
            seg_first[seg] = (uint32_t)(w / SB);
            pos += 2;
            continue;
        }
        end = true;                                                         // EOI or any other marker ends the scan
    }
    if (seg > 0 && w == (size_t)seg_first[seg] * SB) return ("empty restart segment (truncated or corrupt file)");
    seg_end[seg] = (uint32_t)(w * 8);
    const size_t padded = (w + SB - 1) / SB * SB;
    memset(stream + w, 0xFF, padded - w + 16);
    w = padded;
    ++seg;
    if (seg != nseg_expected) return ("restart markers do not match the restart interval (truncated or corrupt file)");
    seg_first[seg] = (uint32_t)(w / SB);
    hd.nseg = seg;
    hd.nsub = (int)(w / SB);
    if (hd.nsub < 1) return ("empty scan");
    // per-subsequence tables: segment, start (unknown), exit (a guess: the next boundary, block 0, DC next)
    int32_t* sub_seg = reinterpret_cast<int32_t*>(blob + L.sub_seg);
    State* st = reinterpret_cast<State*>(blob + L.start);
    State* ex = reinterpret_cast<State*>(blob + L.exit);
    for (int s = 0; s < seg; ++s)
        for (uint32_t i = seg_first[s]; i < seg_first[s + 1]; ++i) {
            sub_seg[i] = s;
            st[i].p = jd::NO_STATE; st[i].bk = jd::NO_STATE;
            const uint32_t b = (i + 1) * (uint32_t)jd::SUB_BITS;
            ex[i].p = b < seg_end[s] ? b : seg_end[s];
            ex[i].bk = 0;
        }
    return nullptr;
}

}  // namespace jdh
