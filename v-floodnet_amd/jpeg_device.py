"""JPEG frames without a host-side image decoder (``csrc/jpeg.hip``).

``Video_DS.__getitem__`` of the reference (``video_module/dataset/Water_DS.py:105-109``) opens every frame with PIL
(``myutils/data.py:87-90``) and turns it into a float tensor on the host.  Here a DataLoader worker only undoes the
entropy coding (``entropy_decode``: Huffman decoding is a serial bit stream, it stays on a host core, in C++ inside
``libvfn_hip.so``) and the main process runs dequantisation, inverse DCT, chroma upsampling, colour conversion and
``ToTensor`` on the GPU (``to_tensor``) -- with libjpeg's integer arithmetic, so the tensor equals the one PIL +
``ToTensor`` would have produced.  Progressive / CMYK / 12-bit files are rejected with a clear error (decode those with
``decode='pil'``).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import ptr, stream, check

_ERR = {-1: 'not a JPEG file or truncated', -2: 'unsupported JPEG variant (progressive, arithmetic-coded, 12-bit, CMYK / '
        'Adobe RGB or non-interleaved scans): decode it with PIL', -3: 'coefficient buffer too small', -4: 'corrupt entropy-coded data'}


def entropy_decode(data):
    """bytes of a JPEG file -> (coef int16[total], qt uint16[3,64], info int32[24]); host only (safe in DataLoader workers)."""
    L = _lib.lib()
    buf = np.frombuffer(data, dtype=np.uint8)
    info = np.zeros(24, np.int32)
    qt = np.zeros((3, 64), np.uint16)
    rc = L.vfn_jpeg_entropy_decode(buf.ctypes.data_as(C.c_void_p), buf.size, None, 0, qt.ctypes.data_as(C.c_void_p),
                                   info.ctypes.data_as(C.c_void_p))
    if rc not in (0, -3):
        raise RuntimeError(f'vfn_jpeg_entropy_decode: {_ERR.get(rc, rc)}')
    coef = np.empty(int(info[23]), np.int16)
    rc = L.vfn_jpeg_entropy_decode(buf.ctypes.data_as(C.c_void_p), buf.size, coef.ctypes.data_as(C.c_void_p), coef.size,
                                   qt.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(f'vfn_jpeg_entropy_decode: {_ERR.get(rc, rc)}')
    return coef, qt, info


_plane_cache = {}


def to_tensor(coef, qt, info, device, out=None, want_u8=False):
    """Coefficients (host or device tensors / arrays from ``entropy_decode``) -> float32 [3,H,W] in [0,1] on ``device``
    (and, with ``want_u8``, the RGB uint8 [H,W,3] image as well)."""
    L = _lib.lib()
    info = np.asarray(info).astype(np.int64).reshape(-1)
    W, H, ncomp, hmax, vmax = (int(info[i]) for i in range(5))
    coef_d = torch.as_tensor(coef).to(device=device, dtype=torch.int16, non_blocking=True).reshape(-1)
    qt_d = torch.as_tensor(np.asarray(qt).astype(np.int16) if not torch.is_tensor(qt) else qt).to(device=device, dtype=torch.int16).reshape(-1)
    # (tensors that are already on the device pass through unchanged: video_seg.main uploads them asynchronously)
    planes, off = [], 0
    for c in range(ncomp):
        bpr, brows = int(info[9 + 4 * c]), int(info[10 + 4 * c])
        key = (str(device), c, bpr, brows, torch.cuda.current_stream().cuda_stream)   # per stream (see png_decode)
        if key not in _plane_cache:
            _plane_cache[key] = torch.empty(brows * 8, bpr * 8, dtype=torch.uint8, device=device)
        pl = _plane_cache[key]
        check(L.vfn_jpeg_idct_u8(C.c_void_p(coef_d.data_ptr() + 2 * off), C.c_void_p(qt_d.data_ptr() + 2 * 64 * c), ptr(pl),
                                 bpr, brows, bpr * 8, stream()), 'vfn_jpeg_idct_u8')
        planes.append(pl)
        off += bpr * brows * 64
    if out is None:
        out = torch.empty(3, H, W, dtype=torch.float32, device=device)
    u8 = torch.empty(H, W, 3, dtype=torch.uint8, device=device) if want_u8 else None
    hs = hmax // int(info[11]) if ncomp == 3 else 1
    vs = vmax // int(info[12]) if ncomp == 3 else 1
    check(L.vfn_jpeg_to_tensor_f32(ptr(planes[0]), ptr(planes[1]) if ncomp == 3 else None, ptr(planes[2]) if ncomp == 3 else None,
                                   planes[0].shape[1], planes[1].shape[1] if ncomp == 3 else 0, W, H, hs, vs, ncomp,
                                   ptr(out), ptr(u8), stream()), 'vfn_jpeg_to_tensor_f32')
    return (out, u8) if want_u8 else out


def decode_file(path, device):
    """JPEG file -> float32 [3,H,W] on the device (entropy decoding on this thread)."""
    with open(path, 'rb') as f:
        coef, qt, info = entropy_decode(f.read())
    return to_tensor(coef, qt, info, device)
