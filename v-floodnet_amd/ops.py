"""Thin host wrappers: torch tensors in, one C-ABI kernel launch each.

Used by the engine (``engine.py``), the feature bank and the parity tests.  Every
function enqueues on the current HIP stream and returns without synchronising.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ptr, stream, check, ConvDesc


_cfg_tiles = None


def conv_cfg_tiles():
    """(BM, BN) of every tile configuration of the library (a constant of the build: asked once)."""
    global _cfg_tiles
    if _cfg_tiles is None:
        L = _lib.lib()
        out = []
        for c in range(L.vfn_conv_cfg_count()):
            bm, bn = C.c_int(), C.c_int()
            L.vfn_conv_cfg_tile(c, C.byref(bm), C.byref(bn))
            out.append((bm.value, bn.value))
        _cfg_tiles = tuple(out)
    return _cfg_tiles


def conv_cfg_wk(cfg):
    """K groups per workgroup of a tile configuration (vfn_conv_cfg_wk): > 1 = split-K inside the workgroup."""
    return _lib.lib().vfn_conv_cfg_wk(int(cfg))


def conv_cfg_tpb(cfg):
    """K tiles per workgroup barrier of a tile configuration (vfn_conv_cfg_tpb)."""
    return _lib.lib().vfn_conv_cfg_tpb(int(cfg))


def conv_cfg_kind(cfg):
    """0 = LDS-tiled, 1 = wave-autonomous (conv_direct_kernel), 2 = stream-K (conv_streamk_kernel) -- vfn_conv_cfg_kind."""
    return _lib.lib().vfn_conv_cfg_kind(int(cfg))


SK_WS_FLOATS = 16 * 1024 * 1024       # include/vfn_hip.h VFN_CONV_SK_WS_FLOATS / VFN_CONV_SK_MAX_TILES
SK_MAX_TILES = 16384
_sk_scratch = {}


def streamk_scratch(device):
    """(workspace, counters) for stream-K launches made through the convenience wrappers (one pair per device; launches on
    one stream run one after the other, so they may share it)."""
    key = str(device)
    if key not in _sk_scratch:
        _sk_scratch[key] = (torch.empty(SK_WS_FLOATS, dtype=torch.float32, device=device),
                            torch.zeros(SK_MAX_TILES, dtype=torch.int32, device=device))
    return _sk_scratch[key]


def set_streamk(desc, workspace, counters):
    """A stream-K configuration (conv_cfg_kind == 2) needs its partial-tile workspace and zeroed tile counters."""
    assert workspace.numel() >= SK_WS_FLOATS and counters.numel() >= SK_MAX_TILES
    desc.ksplit, desc.split_from = 1, 0
    desc.partial, desc.tile_counters = ptr(workspace), ptr(counters)
    return desc


def conv_cfg_names(mode=0):
    """Kernel instantiation behind every tile configuration, as rocprofv3 prints it."""
    L = _lib.lib()
    out = []
    for c in range(L.vfn_conv_cfg_count()):
        v = [C.c_int() for _ in range(5)]
        L.vfn_conv_cfg_info(c, *[C.byref(x) for x in v])
        bm, bn, wm, wn, dma = [x.value for x in v]
        wk, tpb = L.vfn_conv_cfg_wk(c), L.vfn_conv_cfg_tpb(c)
        if L.vfn_conv_cfg_kind(c) > 0:
            buf = C.create_string_buffer(96)
            L.vfn_conv_cfg_name(c, buf, 96)
            out.append(buf.value.decode())
            continue
        out.append(f'conv_igemm_dma_kernel<{bm}, {bn}, {wm}, {wn}, {dma}>' if dma else
                   f'conv_igemm_wk_kernel<{bm}, {bn}, {wm}, {wn}, {wk}, {4 if tpb > 1 else 3}, {tpb}, {int(mode)}>' if (wk > 1 or tpb > 1) else
                   f'conv_igemm_kernel<{bm}, {bn}, {wm}, {wn}, {int(mode)}>')
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


def pack_weights_lp(wp, mode):
    """f32 packed filters [cout_pad, K] -> the operand image the bf16 / bf16x3 conv kernels stage without conversion
    (vfn_conv_desc.w_packed = 1): mode 1: bf16 [cout_pad, K]; mode 2: per 32-channel K tile 32 hi then 32 lo bf16,
    x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (round to nearest even, as the kernels' own conversion)."""
    if mode == 1:
        return wp.to(torch.bfloat16).contiguous()
    assert mode == 2 and wp.shape[1] % 32 == 0
    hi = wp.to(torch.bfloat16)
    lo = (wp - hi.float()).to(torch.bfloat16)
    cp, K = wp.shape
    return torch.cat([hi.view(cp, K // 32, 32), lo.view(cp, K // 32, 32)], dim=2).contiguous()


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
    out_ld = out_ld if out_ld is not None else out.shape[-1]       # (out may be None with an image-only producer: pass out_ld)
    d = ConvDesc()
    d.inp, d.w, d.scale, d.shift, d.res, d.out = ptr(x), ptr(wp), ptr(scale), ptr(shift), ptr(res), ptr(out)
    d.N, d.H, d.W, d.Cin, d.in_ld = N, H, W, cin, in_ld
    d.Ho, d.Wo, d.Cout, d.cout_pad = Ho, Wo, cout, wp.shape[0]
    d.out_ld = out_ld
    d.res_ld = res_ld if res_ld is not None else (res.shape[-1] if res is not None else 0)
    d.KH, d.KW, d.stride, d.pad = kh, kw, stride, pad
    d.relu_in, d.relu_out = int(relu_in), int(relu_out)
    d.M = N * Ho * Wo
    d.ksplit, d.split_from, d.partial = 1, 0, None
    d.tile_counters = None
    d.res_mod = 0
    d.w_packed = 0
    d.in_lp, d.out_lp_relu, d.out_lp = 0, 0, None
    d.mask, d.mask_ld, d.mask_after = None, 0, 0
    d.w_batch_rows = 0
    assert wp.shape[1] == kh * kw * cin
    return d


def use_packed_weights(desc, w_lp):
    """Point the descriptor at filters from ``pack_weights_lp`` (keep ``w_lp`` alive: raw pointer)."""
    desc.w = ptr(w_lp)
    desc.w_packed = 1
    return desc


def set_splitk(desc, ksplit, workspace, split_from=0, rows=None, counters=None):
    """Cut the tiles from index ``split_from`` on along K into ``ksplit`` slices.  ``rows`` = output rows
    those tiles cover (default: all M); ``workspace``: float32 tensor with >= ksplit*rows*Cout elements;
    ``counters``: zeroed int32 tensor (one per split tile) -> the tile is reduced inside the same launch."""
    if ksplit > 1:
        rows = desc.M if rows is None else rows
        assert workspace is not None and workspace.numel() >= ksplit * rows * desc.Cout
        desc.ksplit, desc.split_from, desc.partial = int(ksplit), int(split_from), ptr(workspace)
        desc.tile_counters = ptr(counters) if counters is not None else None
    else:
        desc.ksplit, desc.split_from, desc.partial = 1, 0, None
        desc.tile_counters = None
    return desc


def tail_split_options(desc, bm, bn, max_split=8, mode=0):
    """(split_from, ksplit, rows) candidates that cut only the last partial round of tiles."""
    m_tiles = (desc.M + bm - 1) // bm
    n_tiles = (desc.Cout + bn - 1) // bn
    tiles = m_tiles * n_tiles
    out = []
    for unit in (256, 512):
        full = (tiles // unit) * unit
        full -= full % n_tiles
        if full <= 0 or full >= tiles:
            continue
        rows = desc.M - (full // n_tiles) * bm
        for ks in valid_splits(desc, max_split, mode)[1:]:
            out.append((full, ks, rows))
    return out


def valid_splits(desc, max_split=16, mode=0):
    """Split factors for which every K slice is non-empty and the epilogue can run vectorised."""
    nk = desc.KH * desc.KW * (desc.Cin // (64 if mode == 1 else 32))
    out = [1]
    if desc.Cout % 4 or desc.out_ld % 4 or (desc.res and desc.res_ld % 4):
        return out
    for s_ in range(2, max_split + 1):
        per = (nk + s_ - 1) // s_
        if per * (s_ - 1) < nk and per >= 4:
            out.append(s_)
    return out


# arithmetic of the matrix kernels: 0 exact f32, 1 bf16 operands, 2 bf16x3 (hi/lo split operands, 3 MFMAs per product)
MODES = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}
_CONV_FN = ('vfn_conv2d_nhwc_f32', 'vfn_conv2d_nhwc_bf16', 'vfn_conv2d_nhwc_bf16x3')


# Configuration ids from WINO_GEMM_CFG0 on: a Winograd-domain GEMM descriptor (make_winograd_gemm_desc) launched as the PERSISTENT
# kernel (vfn_winograd_gemm_f32) instead of one workgroup per output tile: id = WINO_GEMM_CFG0 + tile cfg (0..7) + 8 * (workgroups / 128)
WINO_GEMM_CFG0 = 1000
WINO_GEMM_TILES = ((128, 128, 4, 2), (64, 128, 2, 4), (128, 64, 4, 2), (64, 64, 2, 2))


def wino_gemm_cfg(tile_cfg, wgs=512):
    return WINO_GEMM_CFG0 + int(tile_cfg) + 8 * (int(wgs) // 128)


def wino_gemm_cfg_options(rows_pad, cout):
    """Every persistent configuration that fits a GEMM with ``rows_pad`` rows per component."""
    out = []
    for tc in range(8):
        bm, bn = WINO_GEMM_TILES[tc & 3][:2]
        if rows_pad % bm or (bn > 64 and cout < 128):
            continue
        out += [wino_gemm_cfg(tc, w) for w in (256, 512, 768)]
    return out


# ... and from PCONV_CFG0 on: a 1x1 / stride-1 convolution descriptor through the same persistent kernel with the layer's epilogue
# (vfn_conv1x1_persistent_f32), same encoding of tile / request depth / workgroups
PCONV_CFG0 = 2000


def pconv_cfg(tile_cfg, wgs=512):
    return PCONV_CFG0 + int(tile_cfg) + 8 * (int(wgs) // 128)


def pconv_eligible(desc, mode=0):
    """Does vfn_conv1x1_persistent_f32 take this descriptor?  Mirrors every VFN_ERR_ARG condition of that entry point (csrc/
    conv_winograd.hip), so that a descriptor it refuses falls back to the tiled kernels instead of raising at launch."""
    lim = 0x7fffff00
    return (mode == 0 and desc.KH == 1 and desc.KW == 1 and desc.stride == 1 and desc.pad == 0 and not desc.mask and not desc.in_lp and
            not desc.out_lp and not desc.w_packed and not desc.w_batch_rows and desc.Cin % 32 == 0 and desc.in_ld % 4 == 0 and
            desc.M >= 1 and desc.M == desc.N * desc.H * desc.W and desc.M * desc.in_ld * 4 < lim and desc.M * desc.out_ld * 4 < lim and
            desc.cout_pad * desc.Cin * 4 < lim and
            (not desc.res or (desc.res_mod if desc.res_mod > 0 else desc.M) * desc.res_ld * 4 < lim))


def pconv_cfg_options(cout):
    out = []
    for tc in range(8):
        if WINO_GEMM_TILES[tc & 3][1] > 64 and cout < 128:
            continue
        out += [pconv_cfg(tc, w) for w in (256, 512, 768)]
    return out


def conv2d_launch(desc, cfg, mode=0):
    cfg = int(cfg)
    if cfg >= PCONV_CFG0:
        c = cfg - PCONV_CFG0
        if mode != 0:
            raise RuntimeError('the persistent 1x1 convolution is an f32 kernel')
        check(_lib.lib().vfn_conv1x1_persistent_f32(C.byref(desc), c & 7, (c >> 3) * 128, stream()), 'vfn_conv1x1_persistent_f32')
        return
    if cfg >= WINO_GEMM_CFG0:
        c, rows = cfg - WINO_GEMM_CFG0, desc.w_batch_rows
        if mode == 1 and rows > 0:           # plain bf16: the descriptor's `inp` / `w` point at bf16 V / bf16 U (make_winograd_gemm_desc on them)
            check(_lib.lib().vfn_winograd_gemm_bf16(desc.inp, desc.w, desc.out, desc.M // rows, rows, desc.Cin, desc.Cout, desc.cout_pad, c & 7,
                                                    (c >> 3) * 128, stream()), 'vfn_winograd_gemm_bf16')
            return
        if mode != 0 or rows <= 0:
            raise RuntimeError('the persistent GEMM takes an f32 (or plain-bf16) Winograd-domain descriptor (w_batch_rows)')
        check(_lib.lib().vfn_winograd_gemm_f32(desc.inp, desc.w, desc.out, desc.M // rows, rows, desc.Cin, desc.Cout, desc.cout_pad, c & 7,
                                               (c >> 3) * 128, stream()), 'vfn_winograd_gemm_f32')
        return
    name = _CONV_FN[int(mode)]
    check(getattr(_lib.lib(), name)(C.byref(desc), cfg, stream()), name)


def conv_cfg_name(cfg, mode=0, _cache={}):
    """Kernel instantiation behind configuration id ``cfg`` as rocprofv3 prints it (conv_cfg_names + the persistent GEMM ids)."""
    cfg = int(cfg)
    if cfg >= WINO_GEMM_CFG0:
        c = (cfg - (PCONV_CFG0 if cfg >= PCONV_CFG0 else WINO_GEMM_CFG0)) & 7
        bm, bn, wm, wn = WINO_GEMM_TILES[c & 3]
        return f'wino_gemm_kernel<{bm}, {bn}, {wm}, {wn}, {1 + (c >> 2)}, {"true" if cfg >= PCONV_CFG0 else "false"}, {"true" if mode == 1 else "false"}>'
    if mode not in _cache:
        _cache[mode] = conv_cfg_names(mode)
    return _cache[mode][cfg]


BF16_CFGS = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 17, 19, 22, 23) + tuple(range(26, 38))     # (no LDS-DMA variants: the DMA cannot convert)


def conv2d_nhwc(x, wp, cout, kh, kw, stride, pad, scale=None, shift=None, res=None,
                relu_in=False, relu_out=False, cfg=3, out=None, mode=0):
    """Convenience form for tests: allocates the output."""
    N, H, W, _ = x.shape
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty(N, Ho, Wo, cout, device=x.device, dtype=torch.float32)
    d = make_conv_desc(x, wp, cout, kh, kw, stride, pad, out, scale, shift, res, relu_in, relu_out)
    if mode == 0 and conv_cfg_kind(cfg) == 2:
        set_streamk(d, *streamk_scratch(x.device))
    conv2d_launch(d, cfg, mode)
    return out


# --------------------------------------------------------------------------- Winograd F(4x4, 3x3)
_WINO_G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                       dtype=torch.float64)
WINO_ROW_MULT = 256                   # rows per transform component are padded to a multiple of the tallest tile
_WINO_KROT = int(os.environ.get('VFN_KROT', '0'))      # vfn_conv_desc.k_rot for the transform-domain GEMMs (0: every workgroup starts at K tile 0)


def pack_winograd_weight(w):
    """[Cout, Cin, 3, 3] conv filters -> U [36 * cout_pad, Cin] f32: U[6i+j] = (G g G^T)[i][j] for every filter pair, computed
    in float64 (the 36 filter banks of the transform-domain GEMMs; cout_pad = Cout rounded up to 256 like pad_rows)."""
    cout, cin = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3)
    g = w.detach().double().cpu()
    U = torch.einsum('ia,ocab,jb->ijoc', _WINO_G, g, _WINO_G)                 # [6,6,Cout,Cin]
    cp = (cout + 255) // 256 * 256
    out = torch.zeros(36, cp, cin, dtype=torch.float32)
    out[:, :cout] = U.reshape(36, cout, cin).float()
    return out.reshape(36 * cp, cin).contiguous()


def winograd_rows(N, H, W):
    t = N * ((H + 3) // 4) * ((W + 3) // 4)
    return (t + WINO_ROW_MULT - 1) // WINO_ROW_MULT * WINO_ROW_MULT


def pack_winograd_weight_bf16(w):
    """The 36 transform-domain filter banks (pack_winograd_weight: float64 transform) rounded once to bf16: [36 * cout_pad, Cin] bf16."""
    return pack_winograd_weight(w).to(torch.bfloat16).contiguous()


def winograd_input(x, V, rows_pad, relu, N=None, H=None, W=None, cin=None, ld_x=None):
    if N is None:
        N, H, W = x.shape[0], x.shape[1], x.shape[2]
    cin = cin if cin is not None else x.shape[-1]
    ld_x = ld_x if ld_x is not None else x.shape[-1]
    if V.dtype == torch.bfloat16:            # the plain-bf16 mode: V written as bf16
        check(_lib.lib().vfn_winograd_input_bf16(ptr(x), N, H, W, cin, ld_x, int(relu), ptr(V), rows_pad, stream()), 'vfn_winograd_input_bf16')
        return
    check(_lib.lib().vfn_winograd_input_f32(ptr(x), N, H, W, cin, ld_x, int(relu), ptr(V), rows_pad, stream()), 'vfn_winograd_input_f32')


def winograd_output(Mb, rows_pad, out, N, H, W, cout, scale, shift, res, res_ld, res_mod, relu_out, out_ld=None):
    check(_lib.lib().vfn_winograd_output_f32(ptr(Mb), rows_pad, N, H, W, cout, ptr(scale), ptr(shift), ptr(res), int(res_ld), int(res_mod),
                                             int(relu_out), ptr(out), out_ld if out_ld is not None else out.shape[-1], stream()),
          'vfn_winograd_output_f32')


def make_winograd_gemm_desc(V, U, Mb, rows_pad, cin, cout):
    """The 36 transform-domain GEMMs as one batched-filter launch of the convolution kernels."""
    d = make_conv_desc(V, U, cout, 1, 1, 1, 0, Mb, None, None, None, False, False, cin=cin, in_ld=cin, out_ld=cout, N=1, H=1, W=36 * rows_pad)
    d.cout_pad = U.shape[0] // 36
    d.w_batch_rows = rows_pad
    d.k_rot = _WINO_KROT
    return d


def winograd_gemm(V, U, Mb, rows_pad, cin, cout, cfg=0, wgs=0, comps=36):
    """The transform-domain GEMMs as one persistent launch (vfn_winograd_gemm_f32): V [comps * rows_pad, cin], U [comps * cout_pad, cin],
    Mb [comps * rows_pad, cout]."""
    check(_lib.lib().vfn_winograd_gemm_f32(ptr(V), ptr(U), ptr(Mb), comps, rows_pad, cin, cout, U.shape[0] // comps, int(cfg), int(wgs), stream()),
          'vfn_winograd_gemm_f32')


def conv2d_winograd(x, w, scale=None, shift=None, res=None, relu_in=False, relu_out=False, cfg=2, res_mod=0):
    """Convenience form for tests: x NHWC, w [Cout,Cin,3,3] (torch layout) -> NHWC output of the 3x3 / stride-1 / pad-1 convolution."""
    N, H, W, cin = x.shape
    cout = w.shape[0]
    rows = winograd_rows(N, H, W)
    U = pack_winograd_weight(w).to(x.device)
    V = torch.empty(36 * rows, cin, device=x.device, dtype=torch.float32)
    Mb = torch.empty(36 * rows, cout, device=x.device, dtype=torch.float32)
    out = torch.empty(N, H, W, cout, device=x.device, dtype=torch.float32)
    if rows > vfn_winograd_tiles(N, H, W):
        V.view(36, rows, cin)[:, vfn_winograd_tiles(N, H, W):].zero_()
    winograd_input(x, V, rows, relu_in)
    d = make_winograd_gemm_desc(V, U, Mb, rows, cin, cout)
    if conv_cfg_kind(cfg) == 2:
        set_streamk(d, *streamk_scratch(x.device))
    conv2d_launch(d, cfg, 0)
    winograd_output(Mb, rows, out, N, H, W, cout, scale, shift, res, res.shape[-1] if res is not None else 0, res_mod, relu_out)
    return out


def vfn_winograd_tiles(N, H, W):
    return N * ((H + 3) // 4) * ((W + 3) // 4)


# --------------------------------------------------------------------------- stems / pooling
def make_stem_desc(frame, mask, w, scale, shift, out, mean, std, N, H0, W0, pad, Hp, Wp):
    """frame [3,H0,W0] planar; mask [N,H0,W0] or None; out NHWC [N,Hp/2,Wp/2,64]; pad=(lw,uw,lh,uh)."""
    from ._lib import StemDesc
    d = StemDesc()
    d.frame, d.mask, d.w, d.scale, d.shift, d.out = ptr(frame), ptr(mask), ptr(w), ptr(scale), ptr(shift), ptr(out)
    for i in range(3):
        d.mean[i] = float(mean[i])
        d.std[i] = float(std[i])
    d.N, d.cin = N, (5 if mask is not None else 3)
    d.H0, d.W0 = H0, W0
    d.pad_top, d.pad_left = pad[2], pad[0]
    d.Hp, d.Wp, d.Ho, d.Wo = Hp, Wp, Hp // 2, Wp // 2
    return d


def stem_launch(desc):
    check(_lib.lib().vfn_stem_conv7x7_f32(C.byref(desc), stream()), 'vfn_stem_conv7x7_f32')


def pack_stem_weight(ws):
    """list of [64,c_i,7,7] conv weights -> [sum(c_i)*7*8, 64], k=(plane*7+kh)*8+kw, 8th tap zero."""
    w = torch.cat([x.detach().float() for x in ws], dim=1)          # [64, P, 7, 7]
    P = w.shape[1]
    out = torch.zeros(P, 7, 8, 64, dtype=torch.float32, device=w.device)
    out[:, :, :7, :] = w.permute(1, 2, 3, 0)
    return out.reshape(P * 56, 64).contiguous()


def maxpool3x3s2(x, out):
    N, H, W, Cc = x.shape
    check(_lib.lib().vfn_maxpool3x3s2_nhwc_f32(ptr(x), ptr(out), N, H, W, Cc, stream()), 'vfn_maxpool3x3s2_nhwc_f32')
    return out


# --------------------------------------------------------------------------- decoder pointwise
def upsample2x_add(s, pm, out, s_bcast, out_lp=None, lp_relu=False):
    """out = s + bilinear_x2(pm); ``out_lp``: also the split-bf16 image of (relu of) the result (a bf16x3 conv consumes it)."""
    N, h, w, Cc = out.shape
    if out_lp is not None:
        check(_lib.lib().vfn_upsample2x_add_lp_nhwc_f32(ptr(s), ptr(pm), ptr(out), ptr(out_lp), int(lp_relu), N, h, w, Cc,
                                                        int(s_bcast), stream()), 'vfn_upsample2x_add_lp_nhwc_f32')
        return out
    check(_lib.lib().vfn_upsample2x_add_nhwc_f32(ptr(s), ptr(pm), ptr(out), N, h, w, Cc, int(s_bcast), stream()),
          'vfn_upsample2x_add_nhwc_f32')
    return out


def rough_uncertainty(p, p_up, rough, unc):
    obj_n, h, w, _ = p.shape
    check(_lib.lib().vfn_rough_uncertainty_f32(ptr(p), ptr(p_up), ptr(rough), ptr(unc), obj_n, h, w, stream()),
          'vfn_rough_uncertainty_f32')


def local_stats(r1, rough, hs, hr, hm, lm, conf):
    """AFB_URR.py:226-229.  One fused pass (no scratch) for C = 64 and up to 4 objects; otherwise the two-pass pair, which
    needs the scratch buffers hs [obj,h,w,C], hr / hm [obj,h,w]."""
    obj_n, h, w = rough.shape
    Cc = r1.shape[-1]
    L = _lib.lib()
    if Cc == 64 and obj_n <= 4:
        check(L.vfn_local_stats_f32(ptr(r1), ptr(rough), ptr(lm), ptr(conf), obj_n, h, w, Cc, stream()), 'vfn_local_stats_f32')
        return
    check(L.vfn_local_hpass_f32(ptr(r1), ptr(rough), ptr(hs), ptr(hr), ptr(hm), obj_n, h, w, Cc, stream()),
          'vfn_local_hpass_f32')
    check(L.vfn_local_vpass_f32(ptr(hs), ptr(hr), ptr(hm), ptr(lm), ptr(conf), obj_n, h, w, Cc, stream()),
          'vfn_local_vpass_f32')


def local_stats_two_pass(r1, rough, hs, hr, hm, lm, conf):
    """The unfused pair (tests compare the fused kernel with it)."""
    obj_n, h, w = rough.shape
    Cc = r1.shape[-1]
    L = _lib.lib()
    check(L.vfn_local_hpass_f32(ptr(r1), ptr(rough), ptr(hs), ptr(hr), ptr(hm), obj_n, h, w, Cc, stream()), 'vfn_local_hpass_f32')
    check(L.vfn_local_vpass_f32(ptr(hs), ptr(hr), ptr(hm), ptr(lm), ptr(conf), obj_n, h, w, Cc, stream()), 'vfn_local_vpass_f32')


def pred2_gather(z, bias, out):
    """z [N,h,w,ldz] (tap GEMM output) -> out [N,h,w,2] = bias + sum of the nine shifted taps."""
    N, h, w, ldz = z.shape
    check(_lib.lib().vfn_pred2_gather_f32(ptr(z), ptr(bias), ptr(out), N, h, w, ldz, stream()), 'vfn_pred2_gather_f32')
    return out


def final_logits(p_up, unc, conf, q, score, pad, H0, W0):
    obj_n, h, w, _ = p_up.shape
    check(_lib.lib().vfn_final_logits_f32(ptr(p_up), ptr(unc), ptr(conf), ptr(q), ptr(score), obj_n, h, w,
                                          pad[2], pad[0], H0, W0, stream()), 'vfn_final_logits_f32')


def segment_uncertainty(score):
    """score: logits [bs,obj,H,W] from segment -> 0-dim tensor (AFB_URR.py:302-305)."""
    bs, obj_n, H, W = score.shape
    assert score.is_contiguous()
    partial = torch.empty(bs * 64, device=score.device, dtype=torch.float32)
    out = torch.empty(1, device=score.device, dtype=torch.float32)
    check(_lib.lib().vfn_segment_uncertainty_f32(ptr(score), bs, obj_n, H * W, ptr(partial), ptr(out), stream()),
          'vfn_segment_uncertainty_f32')
    return out[0]


def segment_loss(score, label, lu, want_grad=True):
    """train_video_seg.py:72-74 on the logits ``segment`` returns: score f32 [bs,obj,H,W], label int [bs,H,W] ->
    (stats f32[3+bs] = loss, cross entropy, uncertainty, ||u|| per sample; dloss/dscore [bs,obj,H,W] or None)."""
    bs, obj_n, H, W = score.shape
    assert score.is_contiguous()
    lab = label.to(device=score.device, dtype=torch.int32).contiguous()
    partial = torch.empty(2 * bs * 64, device=score.device, dtype=torch.float32)
    stats = torch.empty(3 + bs, device=score.device, dtype=torch.float32)
    grad = torch.empty_like(score) if want_grad else None
    check(_lib.lib().vfn_segment_loss_f32(ptr(score), ptr(lab), bs, obj_n, H * W, float(lu), ptr(partial), ptr(stats), ptr(grad),
                                          stream()), 'vfn_segment_loss_f32')
    return stats, grad


_wgrad_ws = {}
# 1: the pixel slices of a weight gradient are finished inside the launch (vfn_wgrad_desc.tile_counters).  Measured SLOWER on the training
# step (49.6 against 43.1 ms: the wave that arrives last walks ksplit x 64 dwords alone, where the reduce launch spreads them over
# the chip), so it is off; tests/test_kernels_gpu.py keeps the path correct.
_WGRAD_INLAUNCH = os.environ.get('VFN_WGRAD_INLAUNCH', '0') == '1'
_STEM_WGRAD = os.environ.get('VFN_STEM_WGRAD', '1') == '1'        # the 7x7 stems' weight gradient through vfn_stem_wgrad_f32 (0: the generic kernel)


def conv_wgrad(x, gy, k, stride, pad, cin=None, cout=None, ld_x=None, relu=False, rowscale=None, out=None, accumulate=False,
               N=None, H=None, W=None, ksplit=None, batch=1, x_bstride=0, g_bstride=0):
    """dL/dW of y = conv_{k x k, stride, pad}(act(x)) as an implicit GEMM over the pixels (vfn_conv_wgrad_f32): x NHWC
    [N,H,W,ld_x] (``cin`` channels used), gy [N,Ho,Wo,ld_g] (``cout`` used) -> out [cout, k*k*cin] in the packed filter layout
    (kh, kw, cin); ``rowscale`` [cout]: the frozen BatchNorm scale; ``accumulate``: add to ``out``."""
    from ._lib import WgradDesc
    if N is None:
        N, H, W = x.shape[0], x.shape[1], x.shape[2]
    Ho, Wo = gy.shape[1], gy.shape[2]
    ld_x = ld_x if ld_x is not None else (x.stride(-2) if x.dim() >= 2 else x.shape[-1])     # (channel-sliced views keep their pixel stride)
    cin = cin if cin is not None else x.shape[-1]
    cout = cout if cout is not None else gy.shape[-1]
    ld_g = gy.stride(-2)
    assert x.stride(-1) == 1 and gy.stride(-1) == 1
    Kc = k * k * cin
    if out is None:
        assert not accumulate
        out = torch.empty(cout, Kc, device=x.device, dtype=torch.float32)
    assert out.is_contiguous() and out.numel() == batch * cout * Kc
    M = N * Ho * Wo
    if (_STEM_WGRAD and k == 7 and stride == 2 and pad == 3 and cout == 64 and cin in (3, 5) and ld_x == cin and ld_g == 64 and batch == 1
            and not relu and ksplit is None and Ho == (H - 1) // 2 + 1 and Wo == (W - 1) // 2 + 1):
        # the encoders' stems: the operand columns are (kw, c) of one filter row, one pixel walk for all 49 taps (vfn_stem_wgrad_f32)
        L = _lib.lib()
        key = (str(x.device), torch.cuda.current_stream(x.device).cuda_stream, 'stem')
        need = L.vfn_stem_wgrad_scratch_floats(cin)
        part = _wgrad_ws.get(key)
        if part is None or part.numel() < need:
            part = _wgrad_ws[key] = torch.empty(need, device=x.device, dtype=torch.float32)
        check(L.vfn_stem_wgrad_f32(ptr(x), ptr(gy), ptr(rowscale), ptr(out), ptr(part), part.numel(), N, H, W, cin, Ho, Wo, int(accumulate),
                                   stream()), 'vfn_stem_wgrad_f32')
        return out
    tiles = ((cout + 63) // 64 if cout > 32 else 1) * k * k * ((cin + (31 if cin <= 32 else 63)) // (32 if cin <= 32 else 64))
    tiles *= batch                               # (``batch`` independent problems of this shape in one launch: vfn_wgrad_desc.batch)
    if ksplit is None:
        # ~2 000 workgroups, at least 300 pixels per slice (swept on the training step's 51 shapes, scripts/bench_wgrad_shapes.py:
        # 20.0 -> 17.5 ms per step against the first rule, 17.3 with the best split of every shape)
        ksplit = max(1, min(64, 2048 // tiles, M // 300))
        if batch == 1 and cin > 32:
            # round 5 (the step's launches are five times fewer and larger since the decoder is batched over a sample's frames): the
            # 1x1 / 3x3 layers with 3 000 ... 50 000 pixels are fastest with at least 256 workgroups and about 800 pixels per slice --
            # 12 500 x 256 x 512: 71 -> 56 us, 3 125 x 1024 x 256: 40 -> 35 (profiles/r05_wgrad_shapes.txt); the 7x7 stems and the
            # Winograd-domain batches keep the rule above
            ksplit = max(1, min(64, M // 300, max((256 + tiles - 1) // tiles, M // 800)))
    d = WgradDesc()
    d.x, d.gy, d.rowscale, d.dw = ptr(x), ptr(gy), ptr(rowscale), ptr(out)
    d.N, d.H, d.W, d.Cin, d.ld_x = N, H, W, cin, ld_x
    d.Ho, d.Wo, d.Cout, d.ld_g = Ho, Wo, cout, ld_g
    d.k, d.stride, d.pad, d.relu, d.accumulate, d.ksplit = k, stride, pad, int(relu), int(accumulate), int(ksplit)
    d.batch, d.x_bstride, d.g_bstride = int(batch), int(x_bstride), int(g_bstride)
    part = None
    if ksplit > 1:
        key = (str(x.device), torch.cuda.current_stream(x.device).cuda_stream)     # (a workspace per stream: the backward pass
        need = batch * ksplit * cout * Kc                                               #  runs weight gradients beside the data-gradient chain)
        hit = _wgrad_ws.get(key)
        if hit is None or hit[0].numel() < need:
            # (+ the tiles' arrival counters of the in-launch finish, zero at rest; VFN_WGRAD_INLAUNCH=0: the separate reduce launch)
            hit = (torch.empty(max(need, 8 * 1024 * 1024), device=x.device, dtype=torch.float32),
                   hit[1] if hit is not None else torch.zeros(16384, dtype=torch.int32, device=x.device))
            _wgrad_ws[key] = hit
        part = hit[0]
        if _WGRAD_INLAUNCH and tiles <= hit[1].numel() and need * 4 < 0x7fffff00:
            d.tile_counters = ptr(hit[1])
    d.partial = ptr(part)
    check(_lib.lib().vfn_conv_wgrad_f32(C.byref(d), stream()), 'vfn_conv_wgrad_f32')
    return out


def winograd_gy(gy, Z, rows_pad, N, H, W, cout, ld_g):
    check(_lib.lib().vfn_winograd_gy_f32(ptr(gy), N, H, W, cout, ld_g, ptr(Z), rows_pad, stream()), 'vfn_winograd_gy_f32')


_wino_wgrad_ws = {}


def conv_wgrad_winograd(x, gy, cin=None, cout=None, ld_x=None, relu=False, rowscale=None, out=None, accumulate=False, N=None, H=None, W=None):
    """dL/dW of a 3x3 / stride-1 / pad-1 convolution in the Winograd domain (csrc/conv_winograd.hip: dW = G^T [sum_tiles (B^T d B)
    (.) (A dY A^T)] G, a quarter of the direct form's multiplies): input transform of act(x), transform of the gradient tiles,
    the 36 component sums as ONE batched launch of the weight-gradient kernel, back-transform with the row scale / accumulation.
    Same arguments and packed result [cout, 9 * cin] as ``conv_wgrad``; scratch per stream."""
    if N is None:
        N, H, W = x.shape[0], x.shape[1], x.shape[2]
    ld_x = ld_x if ld_x is not None else x.stride(-2)
    cin = cin if cin is not None else x.shape[-1]
    cout = cout if cout is not None else gy.shape[-1]
    ld_g = gy.stride(-2)
    assert cin % 4 == 0 and cout % 4 == 0 and x.stride(-1) == 1 and gy.stride(-1) == 1
    if out is None:
        assert not accumulate
        out = torch.empty(cout, 9 * cin, device=x.device, dtype=torch.float32)
    tiles = vfn_winograd_tiles(N, H, W)
    rows = winograd_rows(N, H, W)
    key = (str(x.device), torch.cuda.current_stream(x.device).cuda_stream)
    need = (36 * rows * cin, 36 * rows * cout, 36 * cout * cin)
    ws = _wino_wgrad_ws.get(key)
    if ws is None or any(t.numel() < n_ for t, n_ in zip(ws, need)):
        old = ws or (None, None, None)
        ws = tuple(t if (t is not None and t.numel() >= n_) else torch.empty(n_, device=x.device, dtype=torch.float32) for t, n_ in zip(old, need))
        _wino_wgrad_ws[key] = ws
    V, Z, dU = ws[0][:need[0]], ws[1][:need[1]], ws[2][:need[2]]
    winograd_input(x, V, rows, relu, N, H, W, cin, ld_x)
    winograd_gy(gy, Z, rows, N, H, W, cout, ld_g)
    # component 0's rows as a one-row image of ``tiles`` pixels; the other 35 components follow at the batch strides
    conv_wgrad(V[:tiles * cin].view(1, 1, tiles, cin), Z[:tiles * cout].view(1, 1, tiles, cout), 1, 1, 0, cin=cin, cout=cout, ld_x=cin,
               out=dU.view(36 * cout, cin), N=1, H=1, W=tiles, batch=36, x_bstride=rows * cin, g_bstride=rows * cout)
    check(_lib.lib().vfn_winograd_dw_f32(ptr(dU), cout, cin, ptr(rowscale), ptr(out), int(accumulate), stream()), 'vfn_winograd_dw_f32')
    return out


def segment_uncertainty_backward(score, g_unc, g_score=None):
    """dL/dscores [bs,obj,H,W] = g_score (optional, from the caller's criterion) + g_unc * d uncertainty / d scores, where
    ``g_unc`` is dL/duncertainty as a 0-dim DEVICE tensor (autograd's hand-over; no host synchronisation)."""
    bs, obj_n, H, W = score.shape
    assert score.is_contiguous() and g_unc.is_cuda and g_unc.numel() == 1 and g_unc.dtype == torch.float32
    if g_score is not None:
        g_score = g_score.to(torch.float32).contiguous()
        assert g_score.shape == score.shape
    partial = torch.empty(2 * bs * 64, device=score.device, dtype=torch.float32)
    stats = torch.empty(3 + bs, device=score.device, dtype=torch.float32)
    grad = torch.empty_like(score)
    check(_lib.lib().vfn_segment_uncertainty_backward_f32(ptr(score), bs, obj_n, H * W, ptr(g_unc), ptr(g_score), ptr(partial),
                                                          ptr(stats), ptr(grad), stream()), 'vfn_segment_uncertainty_backward_f32')
    return grad


# --------------------------------------------------------------------------- loop operators
def resize_bicubic(x, Ho, Wo, out=None):
    """x [C,Hi,Wi] (or [1,C,Hi,Wi]) float32 -> [.., Ho, Wo]."""
    shp = x.shape
    Cc, Hi, Wi = shp[-3], shp[-2], shp[-1]
    assert x.is_contiguous() and x.numel() == Cc * Hi * Wi
    if out is None:
        out = torch.empty(*shp[:-2], Ho, Wo, device=x.device, dtype=torch.float32)
    check(_lib.lib().vfn_resize_bicubic_f32(ptr(x), ptr(out), Cc, Hi, Wi, Ho, Wo, stream()), 'vfn_resize_bicubic_f32')
    return out


def resize_nearest(x, Ho, Wo, out=None):
    shp = x.shape
    Cc, Hi, Wi = shp[-3], shp[-2], shp[-1]
    assert x.is_contiguous() and x.numel() == Cc * Hi * Wi
    if out is None:
        out = torch.empty(*shp[:-2], Ho, Wo, device=x.device, dtype=torch.float32)
    check(_lib.lib().vfn_resize_nearest_f32(ptr(x), ptr(out), Cc, Hi, Wi, Ho, Wo, stream()), 'vfn_resize_nearest_f32')
    return out


def softmax_objects(score, out=None):
    """score [1,obj,H,W] -> softmax over dim 1."""
    obj_n = score.shape[-3]
    n = score.shape[-2] * score.shape[-1]
    assert score.is_contiguous()
    if out is None:
        out = torch.empty_like(score)
    check(_lib.lib().vfn_softmax_objects_f32(ptr(score), ptr(out), obj_n, n, stream()), 'vfn_softmax_objects_f32')
    return out


def resize_argmax(prob, Ho, Wo, out=None):
    """prob [1,obj,Hi,Wi] -> uint8 labels [Ho,Wo] = argmax over objects of the bicubic-resized maps."""
    obj_n, Hi, Wi = prob.shape[-3], prob.shape[-2], prob.shape[-1]
    assert prob.is_contiguous()
    if out is None:
        out = torch.empty(Ho, Wo, device=prob.device, dtype=torch.uint8)
    check(_lib.lib().vfn_resize_argmax_u8(ptr(prob), ptr(out), obj_n, Hi, Wi, Ho, Wo, stream()), 'vfn_resize_argmax_u8')
    return out


def postprocess_pred(pred_np):
    """Host uint8 [H,W] -> uint8 [H,W] (largest 8-connected water blob; myutils/data.py:17-37)."""
    import numpy as np
    pred_np = np.ascontiguousarray(pred_np, dtype=np.uint8)
    out = np.empty_like(pred_np)
    H, W = pred_np.shape
    check(_lib.lib().vfn_postprocess_pred_u8(pred_np.ctypes.data_as(C.c_void_p), H, W,
                                             out.ctypes.data_as(C.c_void_p)), 'vfn_postprocess_pred_u8')
    return out


def postprocess_pred_device(label, out=None, scratch=None):
    """Device uint8 [H,W] -> uint8 [H,W]: postprocessing_pred on the GPU (union-find CCL, largest blob)."""
    H, W = label.shape
    if out is None:
        out = torch.empty_like(label)
    if scratch is None:
        scratch = torch.empty(2 * H * W + 8, dtype=torch.int32, device=label.device)
    check(_lib.lib().vfn_postprocess_pred_device_u8(ptr(label), ptr(out), ptr(scratch), H, W, stream()),
          'vfn_postprocess_pred_device_u8')
    return out


def to_tensor_device(img_u8, out=None):
    """torchvision ToTensor on the device: uint8 [H,W,3] (device) -> float32 [3,H,W] = x / 255."""
    H, W, c = img_u8.shape
    assert c == 3 and img_u8.dtype == torch.uint8 and img_u8.is_contiguous()
    if out is None:
        out = torch.empty(3, H, W, device=img_u8.device, dtype=torch.float32)
    check(_lib.lib().vfn_to_tensor_u8(ptr(img_u8), ptr(out), H, W, stream()), 'vfn_to_tensor_u8')
    return out


_palette_cache = {}


def overlay_device(frame, label, palette, alpha=0.4, cscale=1, out=None):
    """add_overlay + save_overlay's uint8 conversion on the device (myutils/data.py:56-84).
    frame f32 [3,H,W] in [0,1], label uint8 [H,W] (both on the GPU); returns RGB uint8 [H,W,3] on the GPU."""
    _, H, W = frame.shape
    key = (tuple(palette), frame.device)
    if key not in _palette_cache:
        pal = (list(palette) + [0] * 768)[:768]
        _palette_cache[key] = (torch.tensor(pal, dtype=torch.uint8, device=frame.device),
                               torch.empty(1, dtype=torch.int32, device=frame.device))
    pal, scratch = _palette_cache[key]
    if out is None:
        out = torch.empty(H, W, 3, dtype=torch.uint8, device=frame.device)
    check(_lib.lib().vfn_overlay_u8(ptr(frame), ptr(label), ptr(pal), ptr(scratch), ptr(out), H, W,
                                    float(alpha), float(cscale), stream()), 'vfn_overlay_u8')
    return out


# --------------------------------------------------------------------------- bank
def row_norms(x, stride_obj, ld, dim, len_dev, rows, obj_n, nrm, inv, stride_n):
    check(_lib.lib().vfn_row_norms(ptr(x), int(stride_obj), ld, dim, ptr(len_dev), rows, obj_n, ptr(nrm), ptr(inv),
                                   int(stride_n), stream()), 'vfn_row_norms')


def scatter_mean_checked_launch(src, index, index_s0, out, status):
    """src [D,S] (any strides), index int64 (row 0 contiguous; ``index_s0`` = row stride of a materialised [D,S] index, 0 for
    a broadcast / 1-D one), out [D,B] (any strides), status int32[1] on the device (sticky: bit 0 = a target outside
    [0, B), bit 1 = index rows differ).  No host synchronisation."""
    D, S = src.shape
    check(_lib.lib().vfn_scatter_mean_checked_f32(ptr(src), src.stride(0), src.stride(1), ptr(index), int(index_s0), S, ptr(out),
                                                  out.stride(0), out.stride(1), D, out.shape[1], ptr(status), stream()),
          'vfn_scatter_mean_checked_f32')


def scatter_mean_launch(src, index_row, out):
    """src [D,S] (any strides), index_row int64 [S] contiguous, out [D,B] (any strides)."""
    D, S = src.shape
    check(_lib.lib().vfn_scatter_mean_f32(ptr(src), src.stride(0), src.stride(1), ptr(index_row), S, ptr(out),
                                          out.stride(0), out.stride(1), D, stream()), 'vfn_scatter_mean_f32')
