"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

numpy restatement of what libjpeg (the decoder inside PIL, which ``myutils.load_image_in_PIL`` -- myutils/data.py:87-90 --
and ``Video_DS.__getitem__`` -- Water_DS.py:105-109 -- rely on) does after entropy decoding: dequantisation, the accurate
integer IDCT (jidctint.c ``jpeg_idct_islow``), "fancy" chroma upsampling (jdsample.c) and YCbCr -> RGB (jdcolor.c).
libjpeg is a third-party dependency of the reference (through Pillow), not part of /root/reference; its published
integer algorithm is restated here and PINNED against PIL itself: tests/test_jpeg.py compares this oracle (fed by the
product's host-side entropy decoder) with ``PIL.Image.open(...).convert('RGB')`` on JPEGs of every supported layout.
"""
import numpy as np


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct8(v, shift):
    """v: int64 [..., 8] -> [..., 8]; one pass of jpeg_idct_islow."""
    i0, i1, i2, i3, i4, i5, i6, i7 = (v[..., k] for k in range(8))
    z1 = (i2 + i6) * 4433
    tmp2 = z1 + i6 * (-15137)
    tmp3 = z1 + i2 * 6270
    tmp0 = (i0 + i4) << 13
    tmp1 = (i0 - i4) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = i7, i5, i3, i1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * 9633
    t0, t1, t2, t3 = t0 * 2446, t1 * 16819, t2 * 25172, t3 * 12299
    z1, z2, z3, z4 = z1 * -7373, z2 * -20995, z3 * -16069 + z5, z4 * -3196 + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    out = [tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3]
    return np.stack([_descale(o, shift) for o in out], -1)


def idct_plane(coef, qt, bpr, brows):
    """coef int16 [brows*bpr*64] natural order, qt [64] -> uint8 plane [brows*8, bpr*8]."""
    blk = coef.reshape(brows, bpr, 8, 8).astype(np.int64) * qt.reshape(8, 8).astype(np.int64)
    ws = np.swapaxes(_idct8(np.swapaxes(blk, -1, -2), 11), -1, -2)      # pass 1: columns
    px = _idct8(ws, 18) + 128                                           # pass 2: rows, range limit
    px = np.clip(px, 0, 255).astype(np.uint8)
    return px.transpose(0, 2, 1, 3).reshape(brows * 8, bpr * 8)


def _upsample(pl, cw, chh, hs, vs, W, H):
    pl = pl[:chh, :cw].astype(np.int32)
    if hs == 1 and vs == 1:
        return pl[:H, :W]
    ys, xs = np.arange(H), np.arange(W)
    if hs == 2 and cw <= 2:
        return pl[np.ix_(ys // vs, xs // 2)]
    if vs == 2:
        r = ys // 2
        rn = np.clip(np.where(ys & 1, r + 1, r - 1), 0, chh - 1)
        if hs == 1:
            bias = np.where(ys & 1, 2, 1)[:, None]
            return ((3 * pl[r] + pl[rn] + bias) >> 2)[:, :W]
        col = 3 * pl[r] + pl[rn]                                         # [H, cw] column sums
        i = xs // 2
        nb = np.clip(np.where(xs & 1, i + 1, i - 1), 0, cw - 1)
        bias = np.where(xs & 1, 7, 8)[None, :]
        return (3 * col[:, i] + col[:, nb] + bias) >> 4
    i = xs // 2                                                          # h2v1
    nb = np.clip(np.where(xs & 1, i + 1, i - 1), 0, cw - 1)
    bias = np.where(xs & 1, 2, 1)[None, :]
    return (3 * pl[:H][:, i] + pl[:H][:, nb] + bias) >> 2


def decode(coef, qt, info):
    """Outputs of the product's ``jpeg_device.entropy_decode`` -> RGB uint8 [H,W,3] as libjpeg produces it."""
    info = [int(x) for x in info]
    W, H, ncomp, hmax, vmax = info[:5]
    planes, off = [], 0
    for c in range(ncomp):
        bpr, brows = info[9 + 4 * c], info[10 + 4 * c]
        n = bpr * brows * 64
        planes.append(idct_plane(np.asarray(coef[off:off + n]), np.asarray(qt[c]), bpr, brows))
        off += n
    Y = planes[0][:H, :W].astype(np.int32)
    if ncomp == 1:
        return np.stack([Y, Y, Y], -1).astype(np.uint8)
    hs, vs = hmax // info[11], vmax // info[12]
    cw, chh = (W + hs - 1) // hs, (H + vs - 1) // vs
    cb = _upsample(planes[1], cw, chh, hs, vs, W, H) - 128
    cr = _upsample(planes[2], cw, chh, hs, vs, W, H) - 128
    R = Y + ((91881 * cr + 32768) >> 16)
    B = Y + ((116130 * cb + 32768) >> 16)
    G = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    return np.clip(np.stack([R, G, B], -1), 0, 255).astype(np.uint8)
