"""File formats either side of the hot path: what an engine exports (AOV planes, environment maps) and what an
offline run leaves on disk (HDR and tone-mapped images).  Pure Python + numpy + zlib — no imaging library is installed here.

  read_hdr          Radiance RGBE (.hdr): the format of the reference example's environments (example/public/hdr/*.hdr, loaded there
                    by three's RGBELoader) -> scene.environment for rfx_set_environment
  read_exr/write_exr  OpenEXR, scanline, compression NONE / RLE / ZIPS / ZIP / PIZ / PXR24, HALF / FLOAT / UINT channels, arbitrary layer.channel names —
                    the usual container of renderer AOVs; `exr_to_dump_planes` maps its layers onto the dump's attribute planes
  read_pfm/write_pfm  Portable Float Map (the simplest HDR interchange format)
  write_png         8-bit RGB(A) PNG
  tonemap           linear radiance -> display: ACES filmic (the reference example's renderer.toneMapping, example/main.js) or plain
                    clamp, then the sRGB transfer function

All images are (H, W, C) float32 with ROW 0 = BOTTOM, the orientation of every plane in this package (GL texture rows); file formats
that store the top row first are flipped on the way in and out.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np


# ------------------------------------------------------------------------------------------------ Radiance .hdr (RGBE)
def read_hdr(path: str) -> np.ndarray:
    """-> (H, W, 3) float32 linear radiance, row 0 = bottom.  New-style RLE and flat scanlines, -Y +X orientation (the only one
    written in practice)."""
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    if not data.startswith(b"#?"):
        raise ValueError("%s: not a Radiance file" % path)
    fmt_ok = False
    while True:
        end = data.index(b"\n", pos)
        line = data[pos:end]
        pos = end + 1
        if line.startswith(b"FORMAT="):
            fmt_ok = line.strip() == b"FORMAT=32-bit_rle_rgbe"
        if line == b"":
            break
    if not fmt_ok:
        raise ValueError("%s: only FORMAT=32-bit_rle_rgbe is supported" % path)
    end = data.index(b"\n", pos)
    res = data[pos:end].split()
    pos = end + 1
    if len(res) != 4 or res[0] != b"-Y" or res[2] != b"+X":
        raise ValueError("%s: unsupported orientation %r" % (path, res))
    H, W = int(res[1]), int(res[3])
    rgbe = np.empty((H, W, 4), np.uint8)
    buf = np.frombuffer(data, np.uint8)
    for y in range(H):
        if W < 8 or W > 0x7FFF or buf[pos] != 2 or buf[pos + 1] != 2 or (buf[pos + 2] & 0x80):
            rgbe[y] = buf[pos:pos + 4 * W].reshape(W, 4)  # flat scanline
            pos += 4 * W
            continue
        if (int(buf[pos + 2]) << 8 | int(buf[pos + 3])) != W:
            raise ValueError("%s: scanline width mismatch" % path)
        pos += 4
        for c in range(4):
            x = 0
            row = rgbe[y, :, c]
            while x < W:
                n = int(buf[pos]); pos += 1
                if n > 128:  # run
                    n -= 128
                    row[x:x + n] = buf[pos]; pos += 1
                else:  # literal
                    row[x:x + n] = buf[pos:pos + n]; pos += n
                x += n
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0).astype(np.float32)  # 2^(e-128) / 256
    out = rgbe[..., :3].astype(np.float32) * scale[..., None]
    return np.ascontiguousarray(out[::-1])  # file rows run top -> bottom


def environment_from_hdr(path: str) -> np.ndarray:
    """(H, W, 4) float32 equirectangular scene.environment for Context.set_environment (alpha 1), as RGBELoader hands it to three."""
    rgb = read_hdr(path)
    out = np.ones(rgb.shape[:2] + (4,), np.float32)
    out[..., :3] = rgb
    return out


# ------------------------------------------------------------------------------------------------ PFM
def write_pfm(path: str, img: np.ndarray):
    img = np.asarray(img, np.float32)
    if img.ndim == 2:
        img = img[..., None]
    if img.shape[2] not in (1, 3):
        raise ValueError("PFM holds 1 or 3 channels")
    with open(path, "wb") as f:
        f.write(b"%s\n%d %d\n-1.0\n" % (b"PF" if img.shape[2] == 3 else b"Pf", img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, "<f4").tobytes())  # PFM rows run bottom -> top: this package's orientation


def read_pfm(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        kind = f.readline().strip()
        w, h = [int(v) for v in f.readline().split()]
        scale = float(f.readline())
        ch = {b"PF": 3, b"Pf": 1}[kind]
        a = np.frombuffer(f.read(w * h * ch * 4), "<f4" if scale < 0 else ">f4").reshape(h, w, ch)
    return np.ascontiguousarray(a.astype(np.float32))


# ------------------------------------------------------------------------------------------------ PNG
def write_png(path: str, rgb8: np.ndarray):
    """rgb8: (H, W, 3|4) uint8, row 0 = bottom."""
    a = np.ascontiguousarray(np.asarray(rgb8, np.uint8)[::-1])
    h, w, c = a.shape
    if c not in (3, 4):
        raise ValueError("PNG: 3 or 4 channels")
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * c)], axis=1).tobytes()  # filter type 0 per scanline

    def chunk(tag, payload):
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 6, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


def read_png(path: str) -> np.ndarray:
    """8-bit RGB / RGBA, non-interlaced (what write_png writes and what the reference's blue-noise asset is) -> (H, W, C) uint8, row 0 = bottom."""
    with open(path, "rb") as f:
        d = f.read()
    if d[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, hdr = 8, b"", None
    while pos < len(d):
        n, tag = struct.unpack(">I4s", d[pos:pos + 8])
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", d[pos + 8:pos + 8 + n])
        elif tag == b"IDAT":
            idat += d[pos + 8:pos + 8 + n]
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype not in (2, 6) or interlace:
        raise ValueError("PNG: only 8-bit non-interlaced RGB / RGBA")
    c = 3 if ctype == 2 else 4
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * c)
    out = np.zeros((h, w * c), np.uint8)
    for y in range(h):  # undo the per-scanline filters
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        prev = out[y - 1].astype(np.int32) if y else np.zeros(w * c, np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(w * c, np.int32)
            for i in range(w * c):
                a = cur[i - c] if i >= c else 0
                b = prev[i]
                cc = prev[i - c] if i >= c else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - cc), abs(a - cc), abs(a + b - 2 * cc)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
                cur[i] = (line[i] + p) & 255
        out[y] = cur
    return np.ascontiguousarray(out.reshape(h, w, c)[::-1])


def tonemap(linear: np.ndarray, operator: str = "aces", exposure: float = 1.0) -> np.ndarray:
    """(H, W, >=3) linear radiance -> (H, W, 3) uint8 sRGB.  "aces": three.js ACESFilmicToneMapping (the reference example's renderer
    setting, example/main.js: RRT+ODT fit by Stephen Hill, exposure / 0.6); "linear": clamp only."""
    c = np.nan_to_num(np.asarray(linear, np.float64)[..., :3], nan=0.0, posinf=65504.0, neginf=0.0)
    c = np.clip(c, 0.0, 65504.0) * exposure
    if operator == "aces":
        m_in = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
        m_out = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
        c = (c / 0.6) @ m_in.T
        c = (c * (c + 0.0245786) - 0.000090537) / (c * (0.983729 * c + 0.4329510) + 0.238081)
        c = c @ m_out.T
    elif operator != "linear":
        raise ValueError("tonemap operator %r" % (operator,))
    c = np.clip(np.nan_to_num(c, nan=0.0, posinf=1.0, neginf=0.0), 0.0, 1.0)
    s = np.where(c <= 0.0031308, c * 12.92, 1.055 * np.power(c, 1.0 / 2.4) - 0.055)  # sRGB OETF
    return (s * 255.0 + 0.5).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ OpenEXR (scanline)
_PT_UINT, _PT_HALF, _PT_FLOAT = 0, 1, 2
_PT_DTYPE = {_PT_UINT: "<u4", _PT_HALF: "<f2", _PT_FLOAT: "<f4"}
_COMP_NONE, _COMP_RLE, _COMP_ZIPS, _COMP_ZIP, _COMP_PIZ, _COMP_PXR24 = 0, 1, 2, 3, 4, 5


# ---- PIZ (OpenEXR's default for renderer output): per 32-scanline block, the samples as 16-bit words, channel-planar; (1) a bitmap of
# the values that occur + a lookup table onto a dense range, (2) a 2-D Haar-like wavelet per channel plane (14-bit or modulo-16-bit
# arithmetic), (3) canonical Huffman coding of all words with a run-length escape symbol.  Restated from the published format (OpenEXR
# "Technical Introduction" and the reference implementation's ImfPizCompressor / ImfWav / ImfHuf, named for the reader: nothing copied).
_PIZ_BITMAP = 8192
_HUF_ENCSIZE = (1 << 16) + 1  # 65536 word values + the run-length symbol
_SHORT_ZERO, _LONG_ZERO = 59, 63
_SHORTEST_LONG = 2 + _LONG_ZERO - _SHORT_ZERO  # 6
_LONGEST_LONG = 255 + _SHORTEST_LONG


def _wav_pairs(a, b, w14, decode):
    """the 1-D wavelet step on two equally shaped uint16 arrays -> (l, h) when encoding, (a, b) when decoding"""
    if w14:
        if not decode:  # wenc14: m = (a + b) >> 1 (signed 16-bit), d = a - b
            sa, sb = a.astype(np.int16).astype(np.int32), b.astype(np.int16).astype(np.int32)
            return ((sa + sb) >> 1).astype(np.int16).view(np.uint16), (sa - sb).astype(np.int16).view(np.uint16)
        ls, hi = a.astype(np.int16).astype(np.int32), b.astype(np.int16).astype(np.int32)
        ai = ls + (hi & 1) + (hi >> 1)
        return ai.astype(np.int16).view(np.uint16), (ai - hi).astype(np.int16).view(np.uint16)
    a32, b32 = a.astype(np.int32), b.astype(np.int32)
    if not decode:  # wenc16: modulo arithmetic
        ao = (a32 + 32768) & 65535
        m = (ao + b32) >> 1
        d = ao - b32
        m = np.where(d < 0, (m + 32768) & 65535, m)
        return m.astype(np.uint16), (d & 65535).astype(np.uint16)
    bb = (a32 - (b32 >> 1)) & 65535
    aa = (b32 + bb - 32768) & 65535
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2(plane, mx, decode):
    """in-place 2-D wavelet transform of one (ny, nx) uint16 plane (a view: strides carry a FLOAT channel's interleaving)"""
    ny, nx = plane.shape
    w14 = mx < (1 << 14)
    n = min(nx, ny)
    levels = []
    p = 1
    while 2 * p <= n:
        levels.append(p)
        p *= 2
    for p in (reversed(levels) if decode else levels):
        p2 = 2 * p
        ys, xs = ny - ny % p2 if ny % p2 < p else ny - ny % p2, None  # (rows / columns taking part are derived below)
        yq = np.arange(0, ny - p2 + 1, p2) if ny >= p2 else np.zeros(0, int)
        xq = np.arange(0, nx - p2 + 1, p2) if nx >= p2 else np.zeros(0, int)
        if yq.size and xq.size:
            i00 = plane[np.ix_(yq, xq)]; i01 = plane[np.ix_(yq, xq + p)]; i10 = plane[np.ix_(yq + p, xq)]; i11 = plane[np.ix_(yq + p, xq + p)]
            if not decode:
                a, b = _wav_pairs(i00, i01, w14, False); c, d = _wav_pairs(i10, i11, w14, False)
                o00, o10 = _wav_pairs(a, c, w14, False); o01, o11 = _wav_pairs(b, d, w14, False)
            else:
                a, c = _wav_pairs(i00, i10, w14, True); b, d = _wav_pairs(i01, i11, w14, True)
                o00, o01 = _wav_pairs(a, b, w14, True); o10, o11 = _wav_pairs(c, d, w14, True)
            plane[np.ix_(yq, xq)] = o00; plane[np.ix_(yq, xq + p)] = o01; plane[np.ix_(yq + p, xq)] = o10; plane[np.ix_(yq + p, xq + p)] = o11
        if (nx & p) and yq.size:  # odd column: 1-D in y
            x = xq[-1] + p2 if xq.size else 0
            lo, hi = _wav_pairs(plane[yq, x], plane[yq + p, x], w14, decode)
            plane[yq, x], plane[yq + p, x] = lo, hi
        if ny & p:  # odd line: 1-D in x
            y = yq[-1] + p2 if yq.size else 0
            if xq.size:
                lo, hi = _wav_pairs(plane[y, xq], plane[y, xq + p], w14, decode)
                plane[y, xq], plane[y, xq + p] = lo, hi


def _huf_canonical(lengths):
    """code lengths (0 = unused) -> codes, assigned as the format prescribes: numerically ascending within a length, LONGER codes first"""
    lengths = np.asarray(lengths, np.int64)
    count = np.bincount(lengths, minlength=59)
    start = np.zeros(59, np.int64)
    c = 0
    for i in range(58, 0, -1):
        start[i] = c
        c = (c + count[i]) >> 1
    codes = np.zeros(lengths.size, np.int64)
    for ln in np.nonzero(count[1:])[0] + 1:
        idx = np.nonzero(lengths == ln)[0]
        codes[idx] = start[ln] + np.arange(idx.size)
    return codes


class _BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, nbits, value):
        self.acc = (self.acc << nbits) | int(value)
        self.n += nbits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 255)
        self.acc &= (1 << self.n) - 1

    def finish(self):
        nbits = len(self.out) * 8 + self.n
        if self.n:
            self.out.append((self.acc << (8 - self.n)) & 255)
            self.acc, self.n = 0, 0
        return bytes(self.out), nbits


def _huf_compress(words):
    """uint16 words -> the Huffman block (20-byte header, packed code-length table, bit stream)"""
    import heapq
    words = np.asarray(words, np.uint16)
    if words.size == 0:
        return b""
    freq = np.bincount(words, minlength=_HUF_ENCSIZE).astype(np.int64)
    used = np.nonzero(freq)[0]
    im, iM = int(used[0]), int(used[-1]) + 1
    freq[iM] = 1  # the run-length symbol
    syms = np.nonzero(freq)[0]
    lengths = np.zeros(_HUF_ENCSIZE, np.int64)
    if syms.size == 1:
        lengths[syms[0]] = 1
    else:  # plain Huffman: repeatedly join the two rarest subtrees, every symbol under a join gets one bit longer
        heap = [(int(freq[s]), int(s), [int(s)]) for s in syms]
        heapq.heapify(heap)
        while len(heap) > 1:
            fa, ka, la = heapq.heappop(heap)
            fb, kb, lb = heapq.heappop(heap)
            for s in la:
                lengths[s] += 1
            for s in lb:
                lengths[s] += 1
            heapq.heappush(heap, (fa + fb, min(ka, kb), la + lb))
    if lengths.max() > 58:
        raise ValueError("PIZ: Huffman code longer than 58 bits")
    codes = _huf_canonical(lengths)
    tw = _BitWriter()  # code lengths of im..iM, 6 bits each, runs of zeros collapsed
    i = im
    while i <= iM:
        ln = int(lengths[i])
        if ln == 0:
            run = 1
            while i < iM and run < _LONGEST_LONG and lengths[i + 1] == 0:
                i += 1
                run += 1
            if run >= 2:
                if run >= _SHORTEST_LONG:
                    tw.put(6, _LONG_ZERO)
                    tw.put(8, run - _SHORTEST_LONG)
                else:
                    tw.put(6, _SHORT_ZERO + run - 2)
                i += 1
                continue
        tw.put(6, ln)
        i += 1
    table, _ = tw.finish()
    bw = _BitWriter()
    rl_len, rl_code = int(lengths[iM]), int(codes[iM])
    w = words.astype(np.int64)
    change = np.nonzero(np.diff(w))[0] + 1  # run starts
    starts = np.concatenate([[0], change])
    ends = np.concatenate([change, [w.size]])
    for s0, e0 in zip(starts, ends):
        sym = int(w[s0]); ln, code = int(lengths[sym]), int(codes[sym])
        left = int(e0 - s0)
        while left > 0:  # a run of `cs` ADDITIONAL repeats, at most 255 per escape
            cs = min(left - 1, 255)
            if ln + rl_len + 8 < ln * cs:
                bw.put(ln, code); bw.put(rl_len, rl_code); bw.put(8, cs)
            else:
                for _ in range(cs + 1):
                    bw.put(ln, code)
            left -= cs + 1
    data, nbits = bw.finish()
    return struct.pack("<IIIII", im, iM, len(table), nbits, 0) + table + data


def _huf_uncompress(buf, nwords):
    """the Huffman block -> nwords uint16 words"""
    out = np.zeros(nwords, np.uint16)
    if nwords == 0 or len(buf) == 0:
        return out
    im, iM, tlen, nbits, _ = struct.unpack("<IIIII", buf[:20])
    if im >= _HUF_ENCSIZE or iM >= _HUF_ENCSIZE:
        raise ValueError("PIZ: corrupt Huffman header")
    table = buf[20:20 + tlen]
    tb = int.from_bytes(table, "big")
    tbits = len(table) * 8
    pos = 0

    def get(n):
        nonlocal pos
        v = (tb >> (tbits - pos - n)) & ((1 << n) - 1)
        pos += n
        return v

    lengths = np.zeros(_HUF_ENCSIZE, np.int64)
    i = im
    while i <= iM:
        ln = get(6)
        if ln == _LONG_ZERO:
            i += get(8) + _SHORTEST_LONG
        elif ln >= _SHORT_ZERO:
            i += ln - _SHORT_ZERO + 2
        else:
            lengths[i] = ln
            i += 1
    codes = _huf_canonical(lengths)
    # decoding table on the first 14 bits (every code of <= 14 bits fills its range), longer codes by (length, code) lookup
    DEC = 14
    fast_sym = np.full(1 << DEC, -1, np.int64)
    fast_len = np.zeros(1 << DEC, np.int64)
    long_codes = {}
    for sym in np.nonzero(lengths)[0]:
        ln, code = int(lengths[sym]), int(codes[sym])
        if ln <= DEC:
            lo = code << (DEC - ln)
            fast_sym[lo:lo + (1 << (DEC - ln))] = sym
            fast_len[lo:lo + (1 << (DEC - ln))] = ln
        else:
            long_codes[(ln, code)] = int(sym)
    max_len = int(lengths.max())
    data = buf[20 + tlen:20 + tlen + (nbits + 7) // 8]
    db = int.from_bytes(data + b"\0" * 8, "big")  # (8 bytes of slack: the 14-bit window may run past the last code)
    total = (len(data) + 8) * 8
    fs, fl = fast_sym.tolist(), fast_len.tolist()
    bp, n, res = 0, 0, out  # bit position, words produced
    outl = [0] * nwords
    mask = (1 << DEC) - 1
    while bp < nbits and n <= nwords:
        window = (db >> (total - bp - DEC)) & mask
        sym = fs[window]
        if sym >= 0:
            bp += fl[window]
        else:
            ln = DEC + 1
            while True:
                if ln > max_len:
                    raise ValueError("PIZ: invalid Huffman code")
                code = (db >> (total - bp - ln)) & ((1 << ln) - 1)
                sym = long_codes.get((ln, code), -1)
                if sym >= 0:
                    break
                ln += 1
            bp += ln
        if sym == iM:  # run-length escape: repeat the previous word
            cs = (db >> (total - bp - 8)) & 255
            bp += 8
            if n == 0 or n + cs > nwords:
                raise ValueError("PIZ: corrupt run")
            outl[n:n + cs] = [outl[n - 1]] * cs
            n += cs
        else:
            if n >= nwords:
                raise ValueError("PIZ: more words than the block holds")
            outl[n] = sym
            n += 1
    if n != nwords:
        raise ValueError("PIZ: %d of %d words decoded" % (n, nwords))
    res[:] = outl
    return res


def _piz_channel_views(words, chans, W, rows):
    """per channel: the (rows, W) uint16 planes of its words inside a block's channel-planar buffer (FLOAT / UINT: two interleaved planes)"""
    views, p = [], 0
    for _, pt in chans:
        size = 1 if pt == _PT_HALF else 2
        block = words[p:p + rows * W * size].reshape(rows, W * size)
        views.append([block[:, j::size] for j in range(size)])
        p += rows * W * size
    return views


def _piz_compress_block(raw, chans, W, rows):
    """raw: the block's scanline-interleaved bytes (per line: every channel's W samples) -> PIZ bytes"""
    lines = np.frombuffer(raw, "<u2")
    sizes = [1 if pt == _PT_HALF else 2 for _, pt in chans]
    per_line = W * sum(sizes)
    words = np.empty(lines.size, np.uint16)
    p = 0
    for ci, size in enumerate(sizes):  # channel-planar order
        off = W * sum(sizes[:ci])
        words[p:p + rows * W * size] = lines.reshape(rows, per_line)[:, off:off + W * size].reshape(-1)
        p += rows * W * size
    bitmap = np.zeros(65536, bool)
    bitmap[words] = True
    bitmap[0] = False
    lut = np.zeros(65536, np.uint16)
    keep = bitmap.copy()
    keep[0] = True
    lut[keep] = np.arange(int(keep.sum()), dtype=np.uint16)
    mx = int(keep.sum()) - 1
    words = lut[words]
    packed = np.packbits(bitmap.reshape(_PIZ_BITMAP, 8)[:, ::-1], axis=1).reshape(-1)  # bit i of byte i >> 3 (LSB first)
    nz = np.nonzero(packed)[0]
    mn, mxb = (int(nz[0]), int(nz[-1])) if nz.size else (_PIZ_BITMAP - 1, 0)
    head = struct.pack("<HH", mn, mxb) + (packed[mn:mxb + 1].tobytes() if mn <= mxb else b"")
    for planes in _piz_channel_views(words, chans, W, rows):
        for pl in planes:
            _wav2(pl, mx, False)
    huf = _huf_compress(words)
    return head + struct.pack("<i", len(huf)) + huf


def _piz_uncompress_block(buf, chans, W, rows):
    """PIZ bytes -> the block's scanline-interleaved bytes"""
    sizes = [1 if pt == _PT_HALF else 2 for _, pt in chans]
    nwords = rows * W * sum(sizes)
    mn, mxb = struct.unpack("<HH", buf[:4])
    pos = 4
    packed = np.zeros(_PIZ_BITMAP, np.uint8)
    if mn <= mxb:
        packed[mn:mxb + 1] = np.frombuffer(buf, np.uint8, mxb - mn + 1, pos)
        pos += mxb - mn + 1
    bitmap = np.unpackbits(packed.reshape(-1, 1), axis=1)[:, ::-1].reshape(-1).astype(bool)
    bitmap[0] = True
    rev = np.zeros(65536, np.uint16)
    vals = np.nonzero(bitmap)[0]
    rev[:vals.size] = vals
    mx = vals.size - 1
    (hlen,) = struct.unpack("<i", buf[pos:pos + 4])
    pos += 4
    words = _huf_uncompress(buf[pos:pos + hlen], nwords)
    for planes in _piz_channel_views(words, chans, W, rows):
        for pl in planes:
            _wav2(pl, mx, True)
    words = rev[words]
    per_line = W * sum(sizes)
    lines = np.empty((rows, per_line), np.uint16)
    p = 0
    for ci, size in enumerate(sizes):
        off = W * sum(sizes[:ci])
        lines[:, off:off + W * size] = words[p:p + rows * W * size].reshape(rows, W * size)
        p += rows * W * size
    return lines.astype("<u2").tobytes()


# ---- the byte shuffle ZIP / ZIPS / RLE share: even bytes then odd bytes, each byte replaced by its difference to its predecessor + 128
def _exr_predict(raw: bytes) -> bytes:
    a = np.frombuffer(raw, np.uint8)
    re = np.concatenate([a[0::2], a[1::2]])
    d = re.astype(np.int16)
    d[1:] = d[1:] - re[:-1].astype(np.int16) + 128
    return (d & 255).astype(np.uint8).tobytes()


def _exr_unpredict(buf: bytes) -> bytes:
    a = np.frombuffer(buf, np.uint8).astype(np.int32)
    a = ((np.cumsum(a - 128) + 128) & 255).astype(np.uint8)  # t[i] = t[i-1] + d[i] - 128, t[0] = d[0]
    half_n = (a.size + 1) // 2
    re = np.empty(a.size, np.uint8)
    re[0::2], re[1::2] = a[:half_n], a[half_n:]
    return re.tobytes()


# ---- RLE: signed count bytes — n >= 0: the next byte n + 1 times; n < 0: -n literal bytes (runs of 3..128, literals up to 127)
def _rle_compress(buf: bytes) -> bytes:
    out, i, n = bytearray(), 0, len(buf)
    while i < n:
        j = i + 1
        while j < n and buf[j] == buf[i] and j - i < 128:
            j += 1
        if j - i >= 3:
            out += bytes([j - i - 1, buf[i]])
            i = j
            continue
        j = i  # a literal stretch: up to the next run of three, at most 127 bytes
        while j < n and j - i < 127 and not (j + 2 < n and buf[j] == buf[j + 1] == buf[j + 2]):
            j += 1
        out += bytes([(i - j) & 255]) + buf[i:j]
        i = j
    return bytes(out)


def _rle_uncompress(buf: bytes, nbytes: int) -> bytes:
    out, i = bytearray(), 0
    while i < len(buf):
        c = buf[i] - 256 if buf[i] > 127 else buf[i]
        i += 1
        if c < 0:
            out += buf[i:i - c]
            i += -c
        else:
            out += bytes([buf[i]]) * (c + 1)
            i += 1
        if len(out) > nbytes:
            raise ValueError("RLE: more bytes than the block holds")
    if len(out) != nbytes:
        raise ValueError("RLE: %d of %d bytes decoded" % (len(out), nbytes))
    return bytes(out)


# ---- PXR24: per 16-scanline block, zlib over byte PLANES of pixel-to-pixel differences, per scanline and channel: HALF 2 planes, UINT 4,
# FLOAT 3 — a float is first cut to 24 bits (sign, exponent, 15 mantissa bits, rounded): lossy for FLOAT, exact for HALF / UINT
def _float_to_f24(f: np.ndarray) -> np.ndarray:
    u = f.astype(np.float32).view(np.uint32).astype(np.uint64)
    one, eight = np.uint64(1), np.uint64(8)
    s, e, m = u & np.uint64(0x80000000), u & np.uint64(0x7F800000), u & np.uint64(0x007FFFFF)
    rounded = ((e | m) + (m & np.uint64(0x80))) >> eight      # round the mantissa half up; may carry into the exponent ...
    rounded = np.where(rounded >= np.uint64(0x7F8000), (e | m) >> eight, rounded)  # ... but never into infinity: truncate instead
    nan = (e >> eight) | (m >> eight) | np.where((m >> eight) == 0, one, np.uint64(0))  # a NaN keeps a mantissa bit
    r = np.where(e == np.uint64(0x7F800000), np.where(m != 0, nan, e >> eight), rounded)
    return ((s >> eight) | r).astype(np.uint32)


def _pxr24_planes(pt):
    return {_PT_UINT: 4, _PT_HALF: 2, _PT_FLOAT: 3}[pt]


def _pxr24_compress_block(raw: bytes, chans, W: int, rows: int) -> bytes:
    planes, p = [], 0
    for _ in range(rows):
        for _, pt in chans:
            dt = np.dtype(_PT_DTYPE[pt])
            v = np.frombuffer(raw, dt, W, p)
            p += W * dt.itemsize
            k = _pxr24_planes(pt)
            if pt == _PT_FLOAT:
                px = _float_to_f24(v).astype(np.int64)
            else:
                px = v.view(np.uint16 if pt == _PT_HALF else np.uint32).astype(np.int64)
            diff = (np.diff(px, prepend=0) & ((1 << (8 * k)) - 1)).astype(np.uint64)
            for b in range(k - 1, -1, -1):
                planes.append(((diff >> np.uint64(8 * b)) & np.uint64(255)).astype(np.uint8))
    return zlib.compress(np.concatenate(planes).tobytes(), 4)


def _pxr24_uncompress_block(buf: bytes, chans, W: int, rows: int) -> bytes:
    a = np.frombuffer(zlib.decompress(buf), np.uint8)
    out, p = [], 0
    for _ in range(rows):
        for _, pt in chans:
            k = _pxr24_planes(pt)
            if p + k * W > a.size:
                raise ValueError("PXR24: block too short")
            diff = np.zeros(W, np.uint64)
            for b in range(k):
                diff = (diff << np.uint64(8)) | a[p:p + W].astype(np.uint64)
                p += W
            px = np.cumsum(diff) & np.uint64((1 << (8 * k)) - 1)
            if pt == _PT_FLOAT:
                out.append((px.astype(np.uint32) << np.uint32(8)).tobytes())
            elif pt == _PT_HALF:
                out.append(px.astype(np.uint16).tobytes())
            else:
                out.append(px.astype(np.uint32).tobytes())
    if p != a.size:
        raise ValueError("PXR24: %d bytes left over" % (a.size - p))
    return b"".join(out)


def _exr_pack_block(raw: bytes, comp: int, chans, W: int, rows: int) -> bytes:
    """one chunk's scanline-interleaved samples (per line: every channel's W samples) -> what the file stores (raw when that is not smaller)"""
    if comp == _COMP_NONE:
        return raw
    if comp == _COMP_PIZ:
        packed = _piz_compress_block(raw, chans, W, rows)
    elif comp == _COMP_PXR24:
        packed = _pxr24_compress_block(raw, chans, W, rows)
    elif comp == _COMP_RLE:
        packed = _rle_compress(_exr_predict(raw))
    else:
        packed = zlib.compress(_exr_predict(raw), 4)
    return packed if len(packed) < len(raw) else raw


def _exr_unpack_block(buf: bytes, comp: int, chans, W: int, rows: int, nbytes: int) -> bytes:
    if comp == _COMP_NONE or len(buf) >= nbytes:  # (a chunk that did not shrink is stored as it is)
        return buf
    if comp == _COMP_PIZ:
        return _piz_uncompress_block(buf, chans, W, rows)
    if comp == _COMP_PXR24:
        return _pxr24_uncompress_block(buf, chans, W, rows)
    if comp == _COMP_RLE:
        return _exr_unpredict(_rle_uncompress(buf, nbytes))
    return _exr_unpredict(zlib.decompress(buf))


def _exr_attr(name: bytes, typ: bytes, payload: bytes) -> bytes:
    return name + b"\0" + typ + b"\0" + struct.pack("<i", len(payload)) + payload


def write_exr(path: str, channels: dict, compression: str = "zip", half: bool = False, tiles=None):
    """channels: name -> (H, W) array (row 0 = bottom); e.g. {"R": .., "G": .., "B": ..} or layered AOVs {"normal.X": .., "depth.Z": ..}.
    float32 (or half=True: binary16) samples, compression "none" | "rle" | "zips" | "zip" | "piz" | "pxr24" (the last one keeps 24 bits
    of a float32 sample: lossy).  A scanline file, or with tiles=(tile_w, tile_h) a single-level tiled one (what some renderers write
    by default)."""
    names = sorted(channels)  # the format requires alphabetical channel order
    planes = [np.asarray(channels[n]) for n in names]
    H, W = planes[0].shape
    pt = _PT_HALF if half else _PT_FLOAT
    comp = {"none": _COMP_NONE, "rle": _COMP_RLE, "zips": _COMP_ZIPS, "zip": _COMP_ZIP, "piz": _COMP_PIZ, "pxr24": _COMP_PXR24}[compression]
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", pt, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", 0, 0, W - 1, H - 1)
    header = (b"\x76\x2f\x31\x01" + struct.pack("<i", 2 | (0x200 if tiles else 0)) +
              _exr_attr(b"channels", b"chlist", chlist) + _exr_attr(b"compression", b"compression", bytes([comp])) +
              _exr_attr(b"dataWindow", b"box2i", box) + _exr_attr(b"displayWindow", b"box2i", box) +
              _exr_attr(b"lineOrder", b"lineOrder", b"\0") + _exr_attr(b"pixelAspectRatio", b"float", struct.pack("<f", 1.0)) +
              _exr_attr(b"screenWindowCenter", b"v2f", struct.pack("<ff", 0.0, 0.0)) + _exr_attr(b"screenWindowWidth", b"float", struct.pack("<f", 1.0)) +
              (_exr_attr(b"tiles", b"tiledesc", struct.pack("<IIB", tiles[0], tiles[1], 0)) if tiles else b"") + b"\0")
    dt = _PT_DTYPE[pt]
    chans = [(n, pt) for n in names]
    top_down = [np.ascontiguousarray(p[::-1].astype(dt)) for p in planes]  # EXR y = 0 is the TOP row
    blocks = []
    if tiles:
        tw, th = tiles
        for ty in range((H + th - 1) // th):
            for tx in range((W + tw - 1) // tw):
                y0, y1, x0, x1 = ty * th, min(H, ty * th + th), tx * tw, min(W, tx * tw + tw)
                raw = b"".join(tp[y, x0:x1].tobytes() for y in range(y0, y1) for tp in top_down)
                raw = _exr_pack_block(raw, comp, chans, x1 - x0, y1 - y0)
                blocks.append(struct.pack("<iiiii", tx, ty, 0, 0, len(raw)) + raw)
    else:
        per_block = {_COMP_NONE: 1, _COMP_RLE: 1, _COMP_ZIPS: 1, _COMP_ZIP: 16, _COMP_PIZ: 32, _COMP_PXR24: 16}[comp]
        for y0 in range(0, H, per_block):
            y1 = min(H, y0 + per_block)
            raw = b"".join(tp[y].tobytes() for y in range(y0, y1) for tp in top_down)
            raw = _exr_pack_block(raw, comp, chans, W, y1 - y0)
            blocks.append(struct.pack("<ii", y0, len(raw)) + raw)
    table_pos = len(header)
    offs, pos = [], table_pos + 8 * len(blocks)
    for b in blocks:
        offs.append(pos)
        pos += len(b)
    with open(path, "wb") as f:
        f.write(header)
        f.write(struct.pack("<%dQ" % len(offs), *offs))
        for b in blocks:
            f.write(b)


def _exr_parse_header(d: bytes, pos: int):
    """-> (attributes {name: (type, payload)}, position after the header's terminating zero)"""
    attrs = {}
    while d[pos] != 0:
        e = d.index(b"\0", pos); name = d[pos:e]; pos = e + 1
        e = d.index(b"\0", pos); typ = d[pos:e]; pos = e + 1
        n = struct.unpack("<i", d[pos:pos + 4])[0]; pos += 4
        attrs[name] = (typ, d[pos:pos + n]); pos += n
    return attrs, pos + 1


def _exr_part_layout(path, attrs, tiled):
    chans, p, cl = [], 0, attrs[b"channels"][1]
    while cl[p] != 0:
        e = cl.index(b"\0", p); nm = cl[p:e].decode(); p = e + 1
        pt, _, xs, ys = struct.unpack("<iB3xii", cl[p:p + 16]); p += 16
        if xs != 1 or ys != 1:
            raise ValueError("%s: subsampled channels are not supported" % path)
        chans.append((nm, pt))
    comp = attrs[b"compression"][1][0]
    if comp not in (_COMP_NONE, _COMP_RLE, _COMP_ZIPS, _COMP_ZIP, _COMP_PIZ, _COMP_PXR24):
        raise ValueError("%s: compression %d not supported (NONE / RLE / ZIPS / ZIP / PIZ / PXR24 are)" % (path, comp))
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs[b"dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    lay = dict(chans=chans, comp=comp, W=W, H=H, y0=y0, y1=y1, tiled=tiled)
    if tiled:
        tw, th, mode = struct.unpack("<IIB", attrs[b"tiles"][1][:9])
        if mode & 15 == 2:
            raise ValueError("%s: rip-mapped tiles are not supported" % path)
        lay.update(tw=tw, th=th, ntx=(W + tw - 1) // tw, nty=(H + th - 1) // th)
        lay["level0_chunks"] = lay["ntx"] * lay["nty"]  # level 0 comes first in the offset table (the only level, or the mip chain's base)
        lay["chunks"] = struct.unpack("<i", attrs[b"chunkCount"][1])[0] if b"chunkCount" in attrs else lay["level0_chunks"]
        if (mode & 15) and b"chunkCount" not in attrs:  # a single-part mip-mapped file: the table's length follows from the level sizes
            n, w, h, up = 0, W, H, mode >> 4
            while True:
                n += ((w + tw - 1) // tw) * ((h + th - 1) // th)
                if w == 1 and h == 1:
                    break
                w, h = max(1, (w + up) // 2), max(1, (h + up) // 2)
            lay["chunks"] = n
    else:
        lay["per_block"] = {_COMP_ZIP: 16, _COMP_PIZ: 32, _COMP_PXR24: 16}.get(comp, 1)
        lay["chunks"] = lay["level0_chunks"] = (H + lay["per_block"] - 1) // lay["per_block"]
    return lay


def _exr_read_part(path, d, lay, table_pos, part_prefix):
    chans, comp, W, H = lay["chans"], lay["comp"], lay["W"], lay["H"]
    out = {nm: np.empty((H, W), np.uint32 if pt == _PT_UINT else np.float32) for nm, pt in chans}
    px_bytes = sum(np.dtype(_PT_DTYPE[pt]).itemsize for _, pt in chans)

    def scatter(raw, bx, by, cols, rows):  # a chunk's samples (per line: every channel's `cols` samples) into the planes
        p = 0
        for r in range(rows):
            for nm, pt in chans:
                dt = np.dtype(_PT_DTYPE[pt])
                out[nm][H - 1 - (by + r), bx:bx + cols] = np.frombuffer(raw, dt, cols, p)
                p += cols * dt.itemsize

    for o in struct.unpack("<%dQ" % lay["level0_chunks"], d[table_pos:table_pos + 8 * lay["level0_chunks"]]):
        o += part_prefix  # (multi-part files: every chunk starts with its part number)
        if lay["tiled"]:
            tx, ty, lx, ly, n = struct.unpack("<iiiii", d[o:o + 20])
            if lx or ly or not (0 <= tx < lay["ntx"] and 0 <= ty < lay["nty"]):
                raise ValueError("%s: unexpected tile (%d, %d) of level (%d, %d)" % (path, tx, ty, lx, ly))
            cols, rows = min(lay["tw"], W - tx * lay["tw"]), min(lay["th"], H - ty * lay["th"])
            scatter(_exr_unpack_block(d[o + 20:o + 20 + n], comp, chans, cols, rows, cols * rows * px_bytes), tx * lay["tw"], ty * lay["th"], cols, rows)
        else:
            by, n = struct.unpack("<ii", d[o:o + 8])
            rows = min(lay["per_block"], lay["y1"] - by + 1)
            scatter(_exr_unpack_block(d[o + 8:o + 8 + n], comp, chans, W, rows, rows * W * px_bytes), 0, by - lay["y0"], W, rows)
    return out


def read_exr(path: str) -> dict:
    """-> {channel name: (H, W) float32 (UINT channels: uint32)}, row 0 = bottom.  Flat images: scanline or tiled (level 0 of a mip-mapped
    one), single-part or multi-part, compression NONE / RLE / ZIPS / ZIP / PIZ (the format's default) / PXR24, no subsampling — every
    lossless scheme of the format plus PXR24; the lossy block codecs for beauty passes (B44, DWA) and deep data are refused.
    Multi-part files (one AOV per part is common): a part's channels come back as "<part name>.<channel>" unless the channel name already
    carries a layer prefix; all parts must have the same size."""
    with open(path, "rb") as f:
        d = f.read()
    if d[:4] != b"\x76\x2f\x31\x01":
        raise ValueError("%s: not an OpenEXR file" % path)
    version = struct.unpack("<i", d[4:8])[0]
    if version & 0x800:
        raise ValueError("%s: deep data is not supported" % path)
    if not version & 0x1000:  # single part
        attrs, pos = _exr_parse_header(d, 8)
        return _exr_read_part(path, d, _exr_part_layout(path, attrs, bool(version & 0x200)), pos, 0)
    parts, pos = [], 8
    while d[pos] != 0:  # headers until the empty one
        attrs, pos = _exr_parse_header(d, pos)
        typ = attrs.get(b"type", (b"", b"scanlineimage"))[1].rstrip(b"\0")
        if typ not in (b"scanlineimage", b"tiledimage"):
            raise ValueError("%s: part type %r is not supported" % (path, typ))
        parts.append((attrs[b"name"][1].rstrip(b"\0").decode(), _exr_part_layout(path, attrs, typ == b"tiledimage")))
    pos += 1
    out = {}
    for name, lay in parts:
        planes = _exr_read_part(path, d, lay, pos, 4)
        pos += 8 * lay["chunks"]
        for ch, v in planes.items():
            key = ch if ("." in ch or not name) else name + "." + ch
            if key in out:
                raise ValueError("%s: channel %s appears in two parts" % (path, key))
            if out and v.shape != next(iter(out.values())).shape:
                raise ValueError("%s: parts of different sizes" % path)
            out[key] = v
    return out


def write_exr_multipart(path: str, parts: dict, compression: str = "zip", half: bool = False):
    """parts: part name -> {channel: (H, W) array}: one scanline part per entry (how several renderers lay out AOVs)."""
    comp = {"none": _COMP_NONE, "rle": _COMP_RLE, "zips": _COMP_ZIPS, "zip": _COMP_ZIP, "piz": _COMP_PIZ, "pxr24": _COMP_PXR24}[compression]
    pt = _PT_HALF if half else _PT_FLOAT
    dt = _PT_DTYPE[pt]
    per_block = {_COMP_NONE: 1, _COMP_RLE: 1, _COMP_ZIPS: 1, _COMP_ZIP: 16, _COMP_PIZ: 32, _COMP_PXR24: 16}[comp]
    headers, chunk_lists = b"", []
    for k, (pname, channels) in enumerate(parts.items()):
        names = sorted(channels)
        planes = [np.ascontiguousarray(np.asarray(channels[n])[::-1].astype(dt)) for n in names]
        H, W = planes[0].shape
        chans = [(n, pt) for n in names]
        chunks = []
        for y0 in range(0, H, per_block):
            y1 = min(H, y0 + per_block)
            raw = _exr_pack_block(b"".join(tp[y].tobytes() for y in range(y0, y1) for tp in planes), comp, chans, W, y1 - y0)
            chunks.append(struct.pack("<iii", k, y0, len(raw)) + raw)
        chunk_lists.append(chunks)
        chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", pt, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
        box = struct.pack("<iiii", 0, 0, W - 1, H - 1)
        headers += (_exr_attr(b"channels", b"chlist", chlist) + _exr_attr(b"chunkCount", b"int", struct.pack("<i", len(chunks))) +
                    _exr_attr(b"compression", b"compression", bytes([comp])) + _exr_attr(b"dataWindow", b"box2i", box) +
                    _exr_attr(b"displayWindow", b"box2i", box) + _exr_attr(b"lineOrder", b"lineOrder", b"\0") +
                    _exr_attr(b"name", b"string", pname.encode()) + _exr_attr(b"pixelAspectRatio", b"float", struct.pack("<f", 1.0)) +
                    _exr_attr(b"screenWindowCenter", b"v2f", struct.pack("<ff", 0.0, 0.0)) +
                    _exr_attr(b"screenWindowWidth", b"float", struct.pack("<f", 1.0)) + _exr_attr(b"type", b"string", b"scanlineimage") + b"\0")
    head = b"\x76\x2f\x31\x01" + struct.pack("<i", 2 | 0x1000) + headers + b"\0"
    pos = len(head) + 8 * sum(len(c) for c in chunk_lists)
    table = b""
    for chunks in chunk_lists:
        for c in chunks:
            table += struct.pack("<Q", pos)
            pos += len(c)
    with open(path, "wb") as f:
        f.write(head + table)
        for chunks in chunk_lists:
            for c in chunks:
                f.write(c)


# ------------------------------------------------------------------------------------------------ AOV EXR -> dump planes
# layer.channel names a renderer's AOV EXR is expected to carry (rename on export, or pass `names=`): the attributes the G-buffer and
# velocity raster passes of the reference write (GBufferMaterial.js:56-91, VelocityDepthNormalMaterial.js:75-83,180-189)
AOV_LAYOUT = {
    "diffuse": ("diffuse.R", "diffuse.G", "diffuse.B", "diffuse.A"),   # base colour, alpha
    "normal": ("normal.X", "normal.Y", "normal.Z"),                    # WORLD-space normal
    "roughness": ("roughness.Y",),
    "metalness": ("metalness.Y",),
    "emissive": ("emissive.R", "emissive.G", "emissive.B"),
    "velocity": ("velocity.X", "velocity.Y"),                          # uv-space motion, current - previous
    "depth": ("depth.Z",),                                             # gl_FragCoord.z in [0, 1], 1 = not covered
    "direct": ("direct.R", "direct.G", "direct.B", "direct.A"),        # the composer's input buffer (direct lighting)
}


def exr_to_dump_planes(path: str, names: dict | None = None) -> dict:
    """One multi-layer AOV EXR -> the dump's unpacked attribute planes: {"aov": {diffuse, normal, roughness, metalness, emissive,
    velocity}, "depth": (H, W), "direct": (H, W, 4)} ready for Context.pack_gbuffer / pack_velocity (the device packs them into the
    reference's texel formats) or rfx_amd.dump.write_dump."""
    ch = read_exr(path)
    layout = dict(AOV_LAYOUT)
    layout.update(names or {})
    planes = {}
    for key, cn in layout.items():
        missing = [c for c in cn if c not in ch]
        if missing:
            if key == "diffuse" and missing == [cn[3]]:  # no alpha layer: opaque
                ch[cn[3]] = np.ones_like(ch[cn[0]])
            elif key == "direct" and missing == [cn[3]]:
                ch[cn[3]] = np.ones_like(ch[cn[0]])
            else:
                raise KeyError("%s: channel(s) %s of AOV %r missing (have: %s)" % (path, missing, key, sorted(ch)))
        a = np.stack([ch[c].astype(np.float32) for c in cn], axis=-1)
        planes[key] = a[..., 0] if len(cn) == 1 else a
    depth, direct = planes.pop("depth"), planes.pop("direct")
    return {"aov": planes, "depth": np.ascontiguousarray(depth), "direct": np.ascontiguousarray(direct)}
