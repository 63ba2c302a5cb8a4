"""PNG files whose pixel data was compressed on the GPU (``csrc/png.hip``).

Replaces the compression inside the reference's ``save_seg_mask`` (PIL, ``myutils/data.py:49-53``) and
``save_overlay`` (``cv2.imwrite``, ``myutils/data.py:78-84``): the label map / overlay stays on the device, the
deflate stream (a few KB for a mask, about a third of the raw size for an overlay) is what crosses PCIe, and the host
only adds the chunk framing.  The consumers (``est_waterlevel.py:26-28``, ``estimation/reference_tracking.py:166``)
read the files through PIL / OpenCV: any valid PNG with the same pixels and palette is the same input to them.
"""
import ctypes as C
import struct
import threading
import zlib

import torch

from . import _lib
from ._lib import ptr, stream, check

PNG_SIGNATURE = b'\x89PNG\r\n\x1a\n'
_NOWRITE = __import__('os').environ.get('VFN_PNG_NOWRITE') == '1'       # (throughput diagnostics)


def _chunk(tag, payload):
    return struct.pack('>I', len(payload)) + tag + payload + struct.pack('>I', zlib.crc32(tag + payload) & 0xffffffff)


def frame_png(width, height, bpp, deflate, adler, palette=None):
    """signature | IHDR | (PLTE) | IDAT(zlib(deflate)) | IEND.  ``deflate``: bytes of one final deflate block over the
    filtered scanlines; ``adler``: their Adler-32."""
    color_type = 3 if bpp == 1 else 2                       # palette index / RGB, 8 bits per sample
    out = [PNG_SIGNATURE, _chunk(b'IHDR', struct.pack('>IIBBBBB', width, height, 8, color_type, 0, 0, 0))]
    if bpp == 1:
        pal = bytes(int(x) & 255 for x in (list(palette) + [0] * 768)[:768])
        out.append(_chunk(b'PLTE', pal))
    out.append(_chunk(b'IDAT', b'\x78\x01' + deflate + struct.pack('>I', adler & 0xffffffff)))
    out.append(_chunk(b'IEND', b''))
    return b''.join(out)


class PngEncoder:
    """Buffers for one image geometry; ``encode(img)`` enqueues the kernels and the D2H copy of the stream on the
    current HIP stream and returns a ticket; ``finish(ticket)`` (any thread, after the stream has been synchronised or
    via the ticket's event) returns the file bytes."""

    def __init__(self, H, W, bpp, device, slots=4):
        L = _lib.lib()
        wb, ob = C.c_longlong(), C.c_longlong()
        check(L.vfn_png_sizes(H, W, bpp, C.byref(wb), C.byref(ob)), 'vfn_png_sizes')
        self.H, self.W, self.bpp = H, W, bpp
        self.device = device
        self._work = torch.empty(wb.value, dtype=torch.uint8, device=device)
        self._slots = []
        for _ in range(slots):
            self._slots.append(dict(out=torch.empty(ob.value, dtype=torch.uint8, device=device),
                                    stats=torch.zeros(4, dtype=torch.int32, device=device),
                                    host=torch.empty(ob.value, dtype=torch.uint8).pin_memory(),
                                    hstats=torch.zeros(4, dtype=torch.int32).pin_memory(),
                                    event=torch.cuda.Event(), busy=False, free=threading.Event()))
            self._slots[-1]['free'].set()
        self._next = 0
        self._last_n = 0                  # size of the last finished stream: the next one's D2H prefix is sized by it

    def encode(self, img):
        """img: uint8 device tensor [H,W] (bpp 1) or [H,W,3] (bpp 3), contiguous."""
        assert img.dtype == torch.uint8 and img.is_cuda and img.is_contiguous()
        assert tuple(img.shape) == ((self.H, self.W) if self.bpp == 1 else (self.H, self.W, 3)), tuple(img.shape)
        sl = self._slots[self._next]
        if not sl['free'].wait(timeout=60.0):                # a writer thread still owns the oldest slot
            raise RuntimeError('PngEncoder: all slots in flight and none was finished within 60 s')
        sl['free'].clear()
        self._next = (self._next + 1) % len(self._slots)
        check(_lib.lib().vfn_png_deflate_u8(ptr(img), self.H, self.W, self.bpp, ptr(self._work), ptr(sl['out']),
                                            ptr(sl['stats']), stream()), 'vfn_png_deflate_u8')
        # the stream length is only known on the device: copy the statistics and a generous prefix now, the rest (rare:
        # incompressible images) in finish()
        sl['hstats'].copy_(sl['stats'], non_blocking=True)
        sl['prefix'] = min(sl['out'].numel(), max(65536, int(self._last_n * 1.15) + 4096))
        sl['host'][:sl['prefix']].copy_(sl['out'][:sl['prefix']], non_blocking=True)
        sl['event'].record()
        sl['busy'] = True
        return sl

    def finish(self, ticket, palette=None):
        ticket['event'].synchronize()
        n, adler = int(ticket['hstats'][0]), int(ticket['hstats'][1]) & 0xffffffff
        self._last_n = n
        if n > ticket['prefix']:
            ticket['host'][:n].copy_(ticket['out'][:n])
        data = ticket['host'][:n].numpy().tobytes()
        ticket['busy'] = False
        ticket['free'].set()
        return frame_png(self.W, self.H, self.bpp, data, adler, palette)


_encoders = {}


def encoder_for(H, W, bpp, device, slots=4):
    key = (H, W, bpp, str(device), slots)
    if key not in _encoders:
        _encoders[key] = PngEncoder(H, W, bpp, device, slots)
    return _encoders[key]


def png_bytes(img, palette=None):
    """One-shot: device image -> PNG file bytes (synchronises)."""
    bpp = 1 if img.dim() == 2 else 3
    enc = encoder_for(img.shape[0], img.shape[1], bpp, img.device)
    return enc.finish(enc.encode(img.contiguous()), palette)


def _finish_to_file(enc, ticket, palette, path):
    data = enc.finish(ticket, palette)
    if _NOWRITE:
        return
    with open(path, 'wb') as f:
        f.write(data)


class PngSink:
    """The output side of the loop (test_video_seg.py:117-121) without a host-side compressor: label maps and overlays
    are filtered + deflated on the GPU on a side stream (underneath the next frame's kernels), the few-KB / few-hundred-KB
    streams are copied to pinned memory, and writer threads add the chunk framing and write the files."""

    def __init__(self, device, writer, slots=8, beside=()):
        self.device = device
        self.writer = writer                      # data.AsyncWriter
        self.slots = slots
        # (a stream on a hardware queue of its own: _lib.independent_stream)
        from ._lib import independent_stream
        self.stream = independent_stream(device, beside=beside)

    def _submit(self, img, palette, path, ready):
        bpp = 1 if img.dim() == 2 else 3
        enc = encoder_for(img.shape[0], img.shape[1], bpp, img.device, self.slots)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            ticket = enc.encode(img)
        img.record_stream(self.stream)
        self.writer.submit(_finish_to_file, enc, ticket, palette, path)

    def save(self, label_dev, mask_path, palette, frame=None, overlay_path=None, alpha=0.4, cscale=1):
        """label_dev uint8 [H,W] and (optionally) frame f32 [3,H,W], both produced on the current stream.
        Returns the event after which ``label_dev`` is no longer read."""
        from . import ops
        ready = torch.cuda.Event()
        ready.record()
        self._submit(label_dev, palette, mask_path, ready)
        if frame is not None:
            with torch.cuda.stream(self.stream):
                ov = ops.overlay_device(frame.contiguous(), label_dev, palette, alpha, cscale,
                                        out=torch.empty(label_dev.shape[0], label_dev.shape[1], 3, dtype=torch.uint8,
                                                        device=label_dev.device))
                frame.record_stream(self.stream)
            self._submit(ov, None, overlay_path, ready)
        done = torch.cuda.Event()
        with torch.cuda.stream(self.stream):
            done.record()
        return done
