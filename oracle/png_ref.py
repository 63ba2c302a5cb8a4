"""TEST INFRASTRUCTURE (oracle): numpy restatement of PNG scanline reconstruction (PNG specification, section 9.2 "Filter
types for filter method 0") and of PIL's ``convert('RGB')`` for 8-bit colour types 0 / 2 / 3 / 4 / 6 -- what
``myutils.load_image_in_PIL`` (myutils/data.py:87-90) returns for a PNG frame.  Pinned against PIL itself in
tests/test_png_decode.py.  Only tests may import this module."""
import numpy as np


def unfilter(filtered, W, H, bpp):
    """filtered uint8[H*(1+W*bpp)] -> raw uint8[H, W*bpp]."""
    rb = W * bpp
    f = np.asarray(filtered, np.uint8).reshape(H, rb + 1)
    out = np.zeros((H, rb), np.uint8)
    prev = np.zeros(rb, np.int32)
    for r in range(H):
        ft = int(f[r, 0])
        x = f[r, 1:].astype(np.int32)
        cur = np.zeros(rb, np.int32)
        if ft == 0:
            cur = x.copy()
        elif ft == 2:
            cur = (x + prev) & 255
        else:
            for i in range(rb):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                elif ft == 4:
                    pp = a + b - c
                    pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError(f'filter type {ft}')
                cur[i] = (x[i] + p) & 255
        out[r] = cur
        prev = cur
    return out


def to_rgb(raw, W, H, ctype, palette):
    """raw uint8[H, W*bpp] -> RGB uint8[H, W, 3] as PIL's convert('RGB')."""
    bpp = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    px = raw.reshape(H, W, bpp)
    if ctype in (2, 6):
        return px[:, :, :3].copy()
    if ctype == 3:
        return np.asarray(palette, np.uint8).reshape(256, 3)[px[:, :, 0]]
    return np.repeat(px[:, :, :1], 3, axis=2)
