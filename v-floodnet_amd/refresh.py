"""Derived-parameter refresh: everything the engine and the backward pass derive from the model's parameters follows an
optimizer step (train_video_seg.py:76) in two kernel launches (``csrc/refresh.hip``) instead of being rebuilt with tensor
operators -- packed forward filters, data-gradient filters, stem taps, tap-form heads, folded frozen-BatchNorm scale / shift
(train_video_seg.py:103-106 freezes the statistics only).

The objects that own a derived tensor register it here when they build it (``engine.ConvLayer``, ``engine.Pred2Layer``, the stems,
``backward._ConvBwd`` / ``DecoderBackward``); ``Refresher.run()`` rewrites all of them in place, so every descriptor that points
at them stays valid.  The table of (source pointer, destination pointer, geometry) lives in device memory and is rebuilt only
when a tensor moved (``p.data`` re-seated, ``model.to()``) or an entry was added."""
import ctypes as C

import torch

from . import _lib
from ._lib import check, stream, RefreshFilter, RefreshEpilogue

FORWARD, DGRAD, STEM, TAPS, WINO, WINO_DGRAD = 0, 1, 2, 3, 4, 5


def _addr(t):
    return 0 if t is None else t.data_ptr()


class Refresher:
    def __init__(self, device):
        self.device = device
        self._filters = []        # dicts; 'src' a Parameter / tensor, 'bn' a BatchNorm module or None
        self._epilogues = []
        self._tables = None       # (filter table, n, blocks, epilogue table, n) on the device
        self._addresses = None
        self.runs = 0

    # ------------------------------------------------------------------ registration
    def add_filter(self, src, dst, kind, cin=None, cin_off=0, dst_ld=None, dst_row0=0, dst_col0=0, cout_ld=0, bn=None, eps=None):
        cout, cin_total, kh, kw = src.shape
        cin = cin_total if cin is None else cin
        self._filters.append(dict(src=src, dst=dst, kind=kind, cout=cout, cin=cin, cin_off=cin_off, cin_total=cin_total, kh=kh, kw=kw,
                                  dst_ld=dst.shape[-1] if dst_ld is None else dst_ld, dst_row0=dst_row0, dst_col0=dst_col0,
                                  cout_ld=cout_ld or cout, bn=bn, eps=eps))
        self._tables = None

    def add_epilogue(self, scale, shift, bn=None, bias=None, eps=None):
        """BatchNorm ``bn``: scale = weight / sqrt(running_var + eps), shift = bias - running_mean * scale; otherwise
        shift = ``bias`` (a Parameter)."""
        C_ = (scale if scale is not None else shift).numel()
        self._epilogues.append(dict(scale=scale, shift=shift, bn=bn, bias=bias, eps=eps, C=C_))
        self._tables = None

    # ------------------------------------------------------------------ tables
    @staticmethod
    def _usable(t):
        return t is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()

    def _collect(self):
        """Current device addresses of every tensor the tables name (sources first); raises if one is not a contiguous f32
        device tensor (the caller then rebuilds with tensor operators)."""
        out = []
        for e in self._filters:
            ts = [e['src'].data, e['dst']] + ([e['bn'].weight.data, e['bn'].running_var] if e['bn'] is not None else [])
            for t in ts:
                if not self._usable(t):
                    raise RuntimeError('refresh: a parameter is not a contiguous f32 device tensor')
                out.append(t.data_ptr())
        for e in self._epilogues:
            bn = e['bn']
            ts = [bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var] if bn is not None else ([e['bias'].data] if e['bias'] is not None else [])
            for t in ts:
                if not self._usable(t):
                    raise RuntimeError('refresh: a parameter is not a contiguous f32 device tensor')
                out.append(t.data_ptr())
            out += [_addr(e['scale']), _addr(e['shift'])]
        return out

    def _build(self):
        L = _lib.lib()
        per_block = L.vfn_refresh_elems_per_block()
        ft = (RefreshFilter * max(1, len(self._filters)))()
        block = 0
        for i, e in enumerate(self._filters):
            f = ft[i]
            f.src, f.dst = e['src'].data.data_ptr(), e['dst'].data_ptr()
            bn = e['bn']
            if bn is not None:
                f.gamma, f.var, f.eps = bn.weight.data.data_ptr(), bn.running_var.data_ptr(), float(e['eps'] if e['eps'] is not None else bn.eps)
            f.kind = e['kind']
            f.cout, f.cin, f.cin_off, f.cin_total, f.kh, f.kw = e['cout'], e['cin'], e['cin_off'], e['cin_total'], e['kh'], e['kw']
            f.dst_ld, f.dst_row0, f.dst_col0, f.cout_ld = e['dst_ld'], e['dst_row0'], e['dst_col0'], e['cout_ld']
            f.block0 = block
            elems = e['cout'] * e['cin'] * (1 if e['kind'] >= WINO else e['kh'] * e['kw'])
            block += (elems + per_block - 1) // per_block
        et = (RefreshEpilogue * max(1, len(self._epilogues)))()
        for i, e in enumerate(self._epilogues):
            q = et[i]
            bn = e['bn']
            if bn is not None:
                q.gamma, q.beta = bn.weight.data.data_ptr(), bn.bias.data.data_ptr()
                q.mean, q.var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                q.eps = float(e['eps'] if e['eps'] is not None else bn.eps)
            elif e['bias'] is not None:
                q.beta = e['bias'].data.data_ptr()
            q.scale, q.shift, q.C = _addr(e['scale']) or None, _addr(e['shift']) or None, e['C']

        def upload(arr):
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            return host.to(self.device)
        self._tables = (upload(ft), len(self._filters), block, upload(et), len(self._epilogues))

    # ------------------------------------------------------------------ run
    def run(self, verify=True):
        """Rewrite every registered derived tensor from the current parameter values (two launches on the current stream).
        ``verify=False``: the caller vouches that no registered tensor moved since the last run (train.train_step: the parameters
        are views of its own optimizer's flat buffer) -- the ~2 000 address look-ups of the verification (1 ms of host time at the
        end of every step) are skipped while the tables exist."""
        if verify or self._tables is None:
            addresses = self._collect()
            if self._tables is None or addresses != self._addresses:
                self._build()
                self._addresses = addresses
        ft, nf, blocks, et, ne = self._tables
        L = _lib.lib()
        if nf:
            check(L.vfn_refresh_filters_f32(C.c_void_p(ft.data_ptr()), nf, blocks, stream()), 'vfn_refresh_filters_f32')
        if ne:
            check(L.vfn_refresh_epilogues_f32(C.c_void_p(et.data_ptr()), ne, stream()), 'vfn_refresh_epilogues_f32')
        self.runs += 1
