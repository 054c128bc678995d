"""TEST INFRASTRUCTURE - sequential restatement of baseline JPEG decoding as the reference's image loader performs it.

The reference decodes its frames with PIL (`Image.open(...).convert('RGB')`: detnet/data/coco.py via
detnet/inference.py:170 `ToRGB`), i.e. with the libjpeg-turbo bundled in Pillow at its defaults: Huffman baseline
(ITU T.81 Annex F), `JDCT_ISLOW` integer inverse DCT (jidctint.c: CONST_BITS 13, PASS1_BITS 2), "fancy" triangle chroma
upsampling (jdsample.c h2v1 / h2v2) and the 16-bit fixed-point YCbCr -> RGB tables of jdcolor.c.  libjpeg-turbo is a
third-party dependency that is not part of /root/reference; this file restates those published algorithms and is pinned
against PIL itself (tests/test_jpeg_ref.py: bit-exact on synthetic images over sizes, subsamplings, qualities and restart
intervals).  Only tests/, smoke() and bench.py's cpu_baseline leg may import it; the product path is csrc/jpeg_decode.hip.
"""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14,
                   21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53,
                   60, 61, 54, 47, 55, 62, 63])


class Unsupported(ValueError):
    pass


def parse(data):
    """Markers of a baseline, single-scan, Huffman-coded file -> dict (quantisation tables in natural order)."""
    data = bytes(data)
    if data[:2] != b'\xff\xd8':
        raise ValueError('not a JPEG file')
    pos = 2
    qt, dc, ac = {}, {}, {}
    out = dict(restart_interval=0)
    while True:
        while data[pos] != 0xFF:
            pos += 1
        while data[pos] == 0xFF:
            pos += 1
        m = data[pos]
        pos += 1
        if m == 0xD9:
            raise ValueError('no scan')
        if 0xD0 <= m <= 0xD7 or m == 0x01:
            continue
        n = (data[pos] << 8) | data[pos + 1]
        seg = data[pos + 2:pos + n]
        if m == 0xDB:
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                t = np.zeros(64, np.int64)
                for k in range(64):
                    if pq:
                        t[ZIGZAG[k]] = (seg[i] << 8) | seg[i + 1]
                        i += 2
                    else:
                        t[ZIGZAG[k]] = seg[i]
                        i += 1
                qt[tq] = t
        elif m == 0xC4:
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                bits = list(seg[i + 1:i + 17])
                vals = list(seg[i + 17:i + 17 + sum(bits)])
                i += 17 + sum(bits)
                (ac if tc else dc)[th] = (bits, vals)
        elif m in (0xC0, 0xC1):
            if seg[0] != 8:
                raise Unsupported('sample precision %d' % seg[0])
            out['height'], out['width'] = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            out['comps'] = [dict(id=seg[6 + 3 * c], h=seg[7 + 3 * c] >> 4, v=seg[7 + 3 * c] & 15, tq=seg[8 + 3 * c]) for c in range(seg[5])]
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise Unsupported('SOF%d (progressive / lossless / arithmetic) is not baseline' % (m - 0xC0))
        elif m == 0xE0 and seg[:5] == b'JFIF\x00':
            out['jfif'] = True
        elif m == 0xEE and seg[:5] == b'Adobe' and len(seg) >= 12:
            out['adobe_transform'] = seg[11]
        elif m == 0xDD:
            out['restart_interval'] = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            ns = seg[0]
            if ns != len(out['comps']):
                raise Unsupported('non-interleaved multi-scan file')
            for c in range(ns):
                comp = [k for k in out['comps'] if k['id'] == seg[1 + 2 * c]][0]
                comp['td'], comp['ta'] = seg[2 + 2 * c] >> 4, seg[2 + 2 * c] & 15
            pos += n
            break
        pos += n
    out.update(qt=qt, dc=dc, ac=ac)
    ids = [c['id'] for c in out['comps']]
    if len(ids) == 3 and (out.get('adobe_transform') == 0 or ('adobe_transform' not in out and not out.get('jfif') and ids == [82, 71, 66])):
        raise Unsupported('RGB-coded JPEG (no YCbCr transform; jdapimin.c default_decompress_parms)')
    # entropy-coded data: unstuff, cut at restart markers
    segs, cur = [], bytearray()
    while pos < len(data):
        b = data[pos]
        if b != 0xFF:
            cur.append(b)
            pos += 1
            continue
        nb = data[pos + 1] if pos + 1 < len(data) else 0xD9
        if nb == 0:
            cur.append(0xFF)
            pos += 2
        elif 0xD0 <= nb <= 0xD7:
            segs.append(bytes(cur))
            cur = bytearray()
            pos += 2
        elif nb == 0xFF:
            pos += 1
        else:
            break
    segs.append(bytes(cur))
    out['segments'] = segs
    return out


def _derive(bits, vals):
    """canonical code table: {(length, code): symbol} (T.81 Annex C)."""
    table, code, k = {}, 0, 0
    for l in range(1, 17):
        for _ in range(bits[l - 1]):
            table[(l, code)] = vals[k]
            code += 1
            k += 1
        code <<= 1
    return table


class _Bits(object):
    def __init__(self, data):
        self.data, self.pos, self.n = data, 0, 8 * len(data)

    def bit(self):
        if self.pos >= self.n:
            return 1                                                    # libjpeg pads a short segment (it warns and feeds zeros; never reached by valid files)
        b = (self.data[self.pos >> 3] >> (7 - (self.pos & 7))) & 1
        self.pos += 1
        return b

    def bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v


def _symbol(br, table):
    code = 0
    for l in range(1, 17):
        code = (code << 1) | br.bit()
        if (l, code) in table:
            return table[(l, code)]
    raise ValueError('bad Huffman code')


def _extend(r, s):
    return r - (1 << s) + 1 if r < (1 << (s - 1)) else r


def geometry(info):
    comps = info['comps']
    if len(comps) == 1:
        hmax = vmax = 1
        shape = [(1, 1)]
    else:
        hmax, vmax = max(c['h'] for c in comps), max(c['v'] for c in comps)
        shape = [(c['h'], c['v']) for c in comps]
    W, H = info['width'], info['height']
    mx, my = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    return hmax, vmax, shape, mx, my


def decode_coefficients(info):
    """Sequential Huffman decode (T.81 F.2.2; jdhuff.c decode_mcu_slow) -> per component (blocks_y, blocks_x, 64) int16
    in natural order, DC prediction resolved."""
    hmax, vmax, shape, mx, my = geometry(info)
    comps = info['comps']
    dct = [_derive(*info['dc'][c['td']]) for c in comps]
    act = [_derive(*info['ac'][c['ta']]) for c in comps]
    planes = [np.zeros((my * v, mx * h, 64), np.int16) for h, v in shape]
    ri = info['restart_interval'] or mx * my
    mcu = 0
    for seg in info['segments']:
        br = _Bits(seg)
        pred = [0] * len(comps)
        for _ in range(ri):
            if mcu >= mx * my:
                break
            my_i, mx_i = divmod(mcu, mx)
            for ci, (h, v) in enumerate(shape):
                for by in range(v):
                    for bx in range(h):
                        blk = planes[ci][my_i * v + by, mx_i * h + bx]
                        s = _symbol(br, dct[ci])
                        diff = _extend(br.bits(s), s) if s else 0
                        pred[ci] += diff
                        blk[0] = np.int64(pred[ci]).astype(np.int16)
                        k = 1
                        while k < 64:
                            rs = _symbol(br, act[ci])
                            r, s = rs >> 4, rs & 15
                            if s:
                                k += r
                                blk[ZIGZAG[min(k, 63)]] = _extend(br.bits(s), s)      # jpeg_natural_order's guard entries (corrupt runs)
                            elif r == 15:
                                k += 15
                            else:
                                break
                            k += 1
            mcu += 1
    return planes


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _pass(c0, c1, c2, c3, c4, c5, c6, c7, shift_even, out_shift):
    z2, z3 = c2, c6
    z1 = (z2 + z3) * 4433
    tmp2 = z1 + z3 * -15137
    tmp3 = z1 + z2 * 6270
    tmp0 = (c0 + c4) << shift_even
    tmp1 = (c0 - c4) << shift_even
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = c7, c5, c3, c1
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * 9633
    tmp0, tmp1, tmp2, tmp3 = tmp0 * 2446, tmp1 * 16819, tmp2 * 25172, tmp3 * 12299
    z1, z2, z3, z4 = z1 * -7373, z2 * -20995, z3 * -16069 + z5, z4 * -3196 + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    return [_descale(v, out_shift) for v in (tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0,
                                             tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3)]


def idct_islow(coef, quant):
    """jidctint.c jpeg_idct_islow on (..., 64) natural-order coefficients -> (..., 8, 8) uint8."""
    x = coef.astype(np.int64) * quant.astype(np.int64)
    x = x.reshape(x.shape[:-1] + (8, 8))
    cols = _pass(*[x[..., r, :] for r in range(8)], 13, 13 - 2)                  # pass 1: columns; ws[row][col]
    ws = np.stack(cols, axis=-2)
    rows = _pass(*[ws[..., :, c] for c in range(8)], 13, 13 + 2 + 3)              # pass 2: rows
    v = np.stack(rows, axis=-1) & 1023                                          # RANGE_MASK, table centred on 128
    return np.where(v < 128, v + 128, np.where(v < 512, 255, np.where(v < 896, 0, v - 896))).astype(np.uint8)


def component_planes(info, planes=None):
    hmax, vmax, shape, mx, my = geometry(info)
    planes = decode_coefficients(info) if planes is None else planes
    out = []
    for c, p in zip(info['comps'], planes):
        px = idct_islow(p, info['qt'][c['tq']])                                   # (by, bx, 8, 8)
        out.append(px.transpose(0, 2, 1, 3).reshape(p.shape[0] * 8, p.shape[1] * 8))
    return out


def _h2v1_fancy(row):
    """jdsample.c h2v1_fancy_upsample on (rows, w)."""
    a = row.astype(np.int64)
    w = a.shape[1]
    out = np.empty((a.shape[0], 2 * w), np.int64)
    left = np.concatenate([a[:, :1], a[:, :-1]], 1)
    right = np.concatenate([a[:, 1:], a[:, -1:]], 1)
    out[:, 0::2] = (a * 3 + left + 1) >> 2
    out[:, 1::2] = (a * 3 + right + 2) >> 2
    out[:, 0] = a[:, 0]
    out[:, -1] = a[:, -1]
    return out


def _h2v2_fancy(p):
    """jdsample.c h2v2_fancy_upsample on (h, w): rows first (3/4 nearer, 1/4 further, no rounding), then columns."""
    a = p.astype(np.int64)
    h, w = a.shape
    up = np.concatenate([a[:1], a[:-1]], 0)
    dn = np.concatenate([a[1:], a[-1:]], 0)
    out = np.empty((2 * h, 2 * w), np.int64)
    for v, other in ((0, up), (1, dn)):
        s = a * 3 + other                                                        # thiscolsum
        last = np.concatenate([s[:, :1], s[:, :-1]], 1)
        nxt = np.concatenate([s[:, 1:], s[:, -1:]], 1)
        even = (s * 3 + last + 8) >> 4
        odd = (s * 3 + nxt + 7) >> 4
        even[:, 0] = (s[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (s[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out


def decode_rgb(data):
    """bytes of a JPEG file -> (H, W, 3) uint8 as `PIL.Image.open(...).convert('RGB')` returns it."""
    info = parse(data)
    hmax, vmax, shape, mx, my = geometry(info)
    W, H = info['width'], info['height']
    planes = component_planes(info)
    if len(planes) == 1:
        y = planes[0][:H, :W]
        return np.stack([y, y, y], -1)
    if len(planes) != 3:
        raise Unsupported('%d components' % len(planes))
    full = []
    for (h, v), p in zip(shape, planes):
        dw, dh = -(-W * h // hmax), -(-H * v // vmax)                          # downsampled_width / height
        p = p[:dh, :dw]
        if (h, v) == (hmax, vmax):
            f = p.astype(np.int64)
        elif (2 * h, v) == (hmax, vmax):
            f = _h2v1_fancy(p) if dw > 2 else np.repeat(p, 2, 1).astype(np.int64)
        elif (2 * h, 2 * v) == (hmax, vmax):
            f = _h2v2_fancy(p) if dw > 2 else np.repeat(np.repeat(p, 2, 0), 2, 1).astype(np.int64)
        else:
            raise Unsupported('sampling factors %dx%d of %dx%d' % (h, v, hmax, vmax))
        full.append(f[:H, :W])
    y, cb, cr = full
    cb, cr = cb - 128, cr - 128
    fix = lambda x: int(x * 65536 + 0.5)
    r = y + ((fix(1.40200) * cr + 32768) >> 16)
    g = y + ((-fix(0.34414) * cb + 32768 - fix(0.71414) * cr) >> 16)
    b = y + ((fix(1.77200) * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)
