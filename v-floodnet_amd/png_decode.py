"""PNG frames without a host-side image decoder (``csrc/png_decode.hip``).

``Video_DS.__getitem__`` of the reference (``video_module/dataset/Water_DS.py:105-109``) opens every frame with PIL
(``myutils/data.py:87-90``: ``Image.open(path).convert('RGB')``) and ``ToTensor`` divides by 255 on the host.  Here a
DataLoader worker only walks the chunks and inflates the IDAT stream (``inflate``: zlib is a serial bit stream, it stays
on a host core) and the main process undoes the scanline filters and converts the pixels on the GPU (``to_tensor``): the
same tensor, bit for bit.  8-bit grey / RGB / palette / grey+alpha / RGBA, not interlaced, up to 4096 pixels wide; anything
else (16-bit, 1/2/4-bit, Adam7) raises ``RuntimeError('unsupported ...')`` -- decode those with PIL.
"""
import struct
import zlib

import numpy as np
import torch

from . import _lib
from ._lib import ptr, stream, check

_SIG = b'\x89PNG\r\n\x1a\n'
_BPP = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def inflate(data):
    """bytes of a PNG file -> (filtered uint8[H * (1 + W*bpp)], info int32[4] = (W, H, colour type, bpp),
    palette uint8[768]); host only (safe in DataLoader workers)."""
    if data[:8] != _SIG:
        raise RuntimeError('not a PNG file')
    pos, idat, pal, hdr = 8, [], np.zeros(768, np.uint8), None
    while pos + 8 <= len(data):
        n, typ = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif typ == b'PLTE':
            pal[:len(body)] = np.frombuffer(body, np.uint8)
        elif typ == b'IDAT':
            idat.append(body)
        elif typ == b'IEND':
            break
    if hdr is None or not idat:
        raise RuntimeError('truncated PNG file')
    W, H, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype not in _BPP or interlace != 0 or W > 4096:
        raise RuntimeError(f'unsupported PNG variant (bit depth {depth}, colour type {ctype}, interlace {interlace}, width {W}): '
                           'decode it with PIL')
    bpp = _BPP[ctype]
    raw = zlib.decompress(b''.join(idat))
    if len(raw) != H * (1 + W * bpp):
        raise RuntimeError('corrupt PNG data (inflated size does not match the header)')
    out = np.frombuffer(raw, np.uint8).copy()
    # the filter-type byte of every scanline, checked here -- at this frame, in the worker, before anything is segmented or
    # memorised (PIL raises at the frame too); the device kernel keeps its own flag (check_status) as a second line
    if int(out.reshape(H, 1 + W * bpp)[:, 0].max()) > 4:
        raise RuntimeError('corrupt PNG data (scanline filter type outside 0..4)')
    return out, np.array([W, H, ctype, bpp], np.int32), pal


_work_cache = {}


def to_tensor(filtered, info, palette, device, want_u8=False):
    """Inflated scanlines (host or device tensors / arrays from ``inflate``) -> float32 [3,H,W] in [0,1] on ``device``
    (and, with ``want_u8``, the RGB uint8 [H,W,3] image as well)."""
    L = _lib.lib()
    W, H, ctype, bpp = (int(v) for v in np.asarray(info).reshape(-1)[:4])
    pitch, wb = _lib.C.c_int(), _lib.C.c_longlong()
    check(L.vfn_png_unfilter_sizes(W, H, bpp, _lib.C.byref(pitch), _lib.C.byref(wb)), 'vfn_png_unfilter_sizes')
    f_d = torch.as_tensor(filtered).to(device=device, dtype=torch.uint8, non_blocking=True).reshape(-1)
    key = (str(device), W, H, bpp, torch.cuda.current_stream().cuda_stream)    # per stream: concurrent decodes do not share
    if key not in _work_cache:
        _work_cache[key] = (torch.empty(wb.value, dtype=torch.uint8, device=device),
                            torch.empty(H * pitch.value, dtype=torch.uint8, device=device),
                            torch.zeros(1, dtype=torch.int32, device=device))
    work, raw, status = _work_cache[key]
    check(L.vfn_png_unfilter_u8(ptr(f_d), W, H, bpp, ptr(work), ptr(raw), ptr(status), stream()), 'vfn_png_unfilter_u8')
    pal_d = torch.as_tensor(palette).to(device=device, dtype=torch.uint8, non_blocking=True).reshape(-1) if ctype == 3 else None
    out = torch.empty(3, H, W, dtype=torch.float32, device=device)
    u8 = torch.empty(H, W, 3, dtype=torch.uint8, device=device) if want_u8 else None
    check(L.vfn_png_to_tensor_f32(ptr(raw), pitch.value, W, H, ctype, ptr(pal_d), ptr(out), ptr(u8), stream()),
          'vfn_png_to_tensor_f32')
    return (out, u8) if want_u8 else out


def check_status(device):
    """True if no decode on ``device`` since the last call met a filter-type byte outside 0..4 (corrupt data); syncs."""
    ok = True
    for (dev, *_), (_, _, status) in _work_cache.items():
        if dev == str(device) and int(status.item()) != 0:
            status.zero_()
            ok = False
    return ok


def decode_file(path, device):
    """PNG file -> float32 [3,H,W] on the device (inflate on this thread)."""
    with open(path, 'rb') as f:
        filtered, info, pal = inflate(f.read())
    return to_tensor(filtered, info, pal, device)
