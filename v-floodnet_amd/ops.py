"""Thin host wrappers: torch tensors in, one C-ABI kernel launch each.

Used by the engine (``engine.py``), the feature bank and the parity tests.  Every
function enqueues on the current HIP stream and returns without synchronising.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ptr, stream, check, ConvDesc


def conv_cfg_tiles():
    L = _lib.lib()
    out = []
    for c in range(L.vfn_conv_cfg_count()):
        bm, bn = C.c_int(), C.c_int()
        L.vfn_conv_cfg_tile(c, C.byref(bm), C.byref(bn))
        out.append((bm.value, bn.value))
    return out


def pad_rows(wp, mult=256):
    """Pad packed weights [Cout,K] with zero rows to a multiple of ``mult`` filters."""
    cout = wp.shape[0]
    cp = (cout + mult - 1) // mult * mult
    if cp == cout:
        return wp.contiguous()
    out = torch.zeros(cp, wp.shape[1], dtype=wp.dtype, device=wp.device)
    out[:cout] = wp
    return out


def make_conv_desc(x, wp, cout, kh, kw, stride, pad, out, scale=None, shift=None, res=None,
                   relu_in=False, relu_out=False, cin=None, in_ld=None, out_ld=None, res_ld=None,
                   N=None, H=None, W=None):
    """x: NHWC [N,H,W,in_ld]; wp: [cout_pad, kh*kw*cin]; out: [N,Ho,Wo,out_ld] (or 2-D [M,out_ld])."""
    if N is None:
        N, H, W = x.shape[0], x.shape[1], x.shape[2]
    in_ld = in_ld if in_ld is not None else x.shape[-1]
    cin = cin if cin is not None else in_ld
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    out_ld = out_ld if out_ld is not None else out.shape[-1]
    d = ConvDesc()
    d.inp, d.w, d.scale, d.shift, d.res, d.out = ptr(x), ptr(wp), ptr(scale), ptr(shift), ptr(res), ptr(out)
    d.N, d.H, d.W, d.Cin, d.in_ld = N, H, W, cin, in_ld
    d.Ho, d.Wo, d.Cout, d.cout_pad = Ho, Wo, cout, wp.shape[0]
    d.out_ld = out_ld
    d.res_ld = res_ld if res_ld is not None else (res.shape[-1] if res is not None else 0)
    d.KH, d.KW, d.stride, d.pad = kh, kw, stride, pad
    d.relu_in, d.relu_out = int(relu_in), int(relu_out)
    d.M = N * Ho * Wo
    assert wp.shape[1] == kh * kw * cin
    return d


def conv2d_launch(desc, cfg):
    check(_lib.lib().vfn_conv2d_nhwc_f32(C.byref(desc), int(cfg), stream()), 'vfn_conv2d_nhwc_f32')


def conv2d_nhwc(x, wp, cout, kh, kw, stride, pad, scale=None, shift=None, res=None,
                relu_in=False, relu_out=False, cfg=3, out=None):
    """Convenience form for tests: allocates the output."""
    N, H, W, _ = x.shape
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty(N, Ho, Wo, cout, device=x.device, dtype=torch.float32)
    d = make_conv_desc(x, wp, cout, kh, kw, stride, pad, out, scale, shift, res, relu_in, relu_out)
    conv2d_launch(d, cfg)
    return out
