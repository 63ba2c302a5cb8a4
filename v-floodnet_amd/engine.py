"""The HIP execution engine behind ``AFB_URR.memorize`` / ``AFB_URR.segment``.

For one frame size the whole forward is a *static* list of kernel launches over
pre-allocated NHWC buffers (``FramePlan``): descriptors are built once, a frame is
``for launch in plan: launch()``.  Weights are repacked once per ``load_state_dict``
(``[Cout][kh][kw][Cin]`` filters, BatchNorm as per-channel scale/shift, the three
memory-encoder stems as one 5-plane filter bank).

Work the reference does twice is done once (mathematically identical):
``RF3/RF2.convFS + ResFS`` see the same ``r3`` / ``r2`` for every object
(``AFB_URR.py:289-295`` only ``expand``s them), so they run at N=1 and are broadcast
into the per-object ``ResMM`` path by the fused upsample-add kernel.

Call graph per frame (reference lines in brackets):
  segment   pad+norm+stem [AFB_URR.py:279-285] -> res2..res4 -> KeyValue -> memory read
            [136-178] -> decoder [208-239] -> logits [300-316]
  memorize  pad+norm+3 stems [259-266] -> res2..res4 -> KeyValue [268-272]
"""
import math
import os

import torch

from . import _lib, ops, weights as W
from ._lib import ptr, stream, check, MemReadDesc, BankScanDesc
from .feature_bank import pick_nsplit, pick_scan_slices, MAX_SPLIT, MAX_SPLIT_SCAN, QT_SCAN, DK, DV

# (BM, BN) -> relative efficiency of the tile shape in the implicit-GEMM kernel
_CFG_EFF = {(128, 128): 1.00, (128, 64): 0.95, (64, 128): 0.95, (64, 64): 0.86, (32, 64): 0.74,
            (64, 32): 0.74, (128, 32): 0.80, (256, 128): 0.97}
_CFG_TILES = None
_TUNED = {}          # (M, Cout, K) -> (cfg, ksplit): measured choices (Engine.autotune / tuned_gfx950.json)
_TUNED_PATH = __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)),
                                         'tuned_gfx950.json')
_TUNED_BF16 = {}     # the same table for the bf16-operand kernel (64-channel K tiles: its own split factors)
_TUNED_BF16_PATH = _TUNED_PATH.replace('tuned_gfx950.json', 'tuned_gfx950_bf16.json')
_TUNED_BF16X3 = {}   # ... and for the bf16x3 kernel
_TUNED_BF16X3_PATH = _TUNED_PATH.replace('tuned_gfx950.json', 'tuned_gfx950_bf16x3.json')
_TABLES = (_TUNED, _TUNED_BF16, _TUNED_BF16X3)                 # by ops.MODES value
_TABLE_PATHS = (_TUNED_PATH, _TUNED_BF16_PATH, _TUNED_BF16X3_PATH)


def _load_tuned():
    import json
    import os
    if os.environ.get('VFN_IGNORE_TUNED') == '1':
        return
    for path, table in zip(_TABLE_PATHS, _TABLES):
        alt = os.environ.get('VFN_TUNED_DIR')                     # (A/B of two sets of tables: same file names in another directory)
        if alt and os.path.isfile(os.path.join(alt, os.path.basename(path))):
            path = os.path.join(alt, os.path.basename(path))
        if os.path.isfile(path):
            for k, v in json.load(open(path)).items():
                # (cfg, ksplit, split_from) + optionally the same triple again for descriptors of this shape the first choice cannot
                # run: a persistent 1x1 configuration (ids >= 2000) does not take the backward passes' masked data-gradient
                # convolutions, which share the shape key -- apply_choice falls back to the second triple
                t = (int(v[0]), int(v[1]), int(v[2]) if len(v) > 2 else 0)
                table[tuple(int(x) for x in k.split(','))] = t + tuple(int(x) for x in v[3:6]) if len(v) >= 6 else t


def save_tuned(path=None, mode=0):
    import json
    table = _TABLES[mode]
    with open(path or _TABLE_PATHS[mode], 'w') as f:
        json.dump({','.join(str(x) for x in k): list(v) for k, v in sorted(table.items())}, f, indent=0)


_load_tuned()


def pad_divide_by(h, w, d=16):
    """Pad amounts of myutils.pad_divide_by (myutils/data.py:132-149): (lw, uw, lh, uh)."""
    new_h = h + d - h % d if h % d > 0 else h
    new_w = w + d - w % d if w % d > 0 else w
    lh = int((new_h - h) / 2)
    uh = int(new_h - h) - lh
    lw = int((new_w - w) / 2)
    uw = int(new_w - w) - lw
    return (lw, uw, lh, uh), new_h, new_w


WS_FLOATS = 16 * 1024 * 1024        # split-K workspace (64 MB), shared by all launches of a plan
_WINOGRAD_TRAIN = os.environ.get('VFN_WINOGRAD_TRAIN', '1') == '1'     # the training plans' forward convolutions too
_BATCH_MEMREAD = os.environ.get('VFN_TRAIN_BATCH_MEMREAD', '1') == '1'     # segment_batch: one memory read for all frames of the sample
MAX_GROUP = 16                # Engine.segment_group: frames per batched pass (the batch's buffers are G times a frame's)
_TRAIN_SLOTS = int(os.environ.get('VFN_TRAIN_SLOTS', 2))      # training plans segment() alternates between (see Engine.plan)
# Winograd F(4x4, 3x3) for the 3x3 / stride-1 layers (csrc/conv_winograd.hip): 1 (default) = where the measured table says so
# (wino_gfx950.json: "M,cin,cout" -> 0 / 1, scripts/tune_winograd.py; shapes it lacks: >= 128 channels either side and at least
# VFN_WINOGRAD_MIN_M output pixels), 2 = every eligible layer, 0 = off (the direct implicit GEMM everywhere)
_WINOGRAD = os.environ.get('VFN_WINOGRAD', '1')
_WINOGRAD_LP = os.environ.get('VFN_WINOGRAD_LP', '1') == '1'        # Winograd layers in the plain-bf16 mode too (bf16 V / U, vfn_winograd_gemm_bf16)
_WINOGRAD_MIN_M = int(os.environ.get('VFN_WINOGRAD_MIN_M', 10000))
_WINO_TABLE = {}
_WINO_TABLE_BF16 = {}                 # the plain-bf16 mode's own decisions (its direct kernels are faster, its transforms the same)
_WINO_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wino_gfx950.json')
if os.environ.get('VFN_IGNORE_TUNED') != '1':
    for _path, _tab in ((_WINO_PATH, _WINO_TABLE), (_WINO_PATH.replace('.json', '_bf16.json'), _WINO_TABLE_BF16)):
        if os.path.isfile(_path):
            _tab.update({tuple(int(x) for x in k.split(',')): int(v) for k, v in __import__('json').load(open(_path)).items()})
_INLAUNCH_SPLITK = __import__('os').environ.get('VFN_INLAUNCH_SPLITK', '1') == '1'


def _tiles():
    global _CFG_TILES
    if _CFG_TILES is None:
        _CFG_TILES = ops.conv_cfg_tiles()
    return _CFG_TILES


def apply_choice(desc, choice, ws, counters=None):
    """Configure a conv descriptor for a (cfg, ksplit, split_from) choice; returns the cfg index."""
    cfg, ks, split_from = choice[:3]
    if ks > 1 and (desc.out_ld % 4 or (desc.res and desc.res_ld % 4)):
        # (the split-K paths move 16 bytes at a time; checked HERE, on the triple actually used: the callers' guard only saw the
        # first triple of a six-entry table row, whose fallback may be a split-K choice -- ADVICE r5)
        ks, split_from = 1, 0
    if cfg >= ops.PCONV_CFG0 and not ops.pconv_eligible(desc, 0):
        # the shape's measured choice is the persistent 1x1 kernel, this descriptor (a masked data gradient, an operand image, ...)
        # is not one it takes: the shape's second entry, else the heuristic
        fb = tuple(choice[3:6]) if len(choice) >= 6 else choose_cfg(desc.M, desc.Cout, desc.KH * desc.KW * desc.Cin, 0, use_table=False)
        return apply_choice(desc, fb, ws, counters)
    if cfg >= ops.WINO_GEMM_CFG0:                         # the persistent kernels: no split, no workspace
        ops.set_splitk(desc, 1, None)
        return cfg
    if ops.conv_cfg_kind(cfg) == 2:                       # stream-K: its own workspace + counters, never a K split on top
        if counters is None or ws is None or ws.numel() < ops.SK_WS_FLOATS:
            # (callers with no counters of their own -- the backward pass, LinkNet: main-stream launches, one after the other)
            ws, counters = ops.streamk_scratch(ws.device if ws is not None else torch.device('cuda', torch.cuda.current_device()))
        ops.set_streamk(desc, ws, counters)
        return cfg
    if ks > 1:
        bm, bn = _tiles()[cfg]
        n_tiles = (desc.Cout + bn - 1) // bn
        rows = desc.M - (split_from // n_tiles) * bm
        # In-launch finish (the slice that arrives last reduces the tile; bit-identical to the reduce launch).  Round 1 published
        # the partials with an agent-scope release fence (an L2 write-back per slice workgroup) and measured slower than the
        # second launch; round 4 publishes them with write-through stores instead (no fence).  VFN_INLAUNCH_SPLITK=0 restores
        # the separate reduce launch.  f32 LDS-tiled configurations only.
        if desc.Cout % bn or not _INLAUNCH_SPLITK or ops.conv_cfg_kind(cfg) != 0 or desc.w_packed or desc.in_lp or desc.out_lp:
            counters = None
        ops.set_splitk(desc, ks, ws, split_from, rows, counters)
    else:
        ops.set_splitk(desc, 1, None)
    return cfg


def _scores_buffer(p, fb):
    """[obj][ceil(cap/64) * ceil(HW/128) * 8192] floats for the scores of one frame, or None when switched off / too large."""
    if os.environ.get('VFN_STORE_SCORES', '1') == '0':
        return None
    per_obj = ((fb._cap + 63) // 64) * ((p.HW + 127) // 128) * 8192
    if fb.obj_n * per_obj * 4 > float(os.environ.get('VFN_SCORES_MAX_GB', 48)) * 2 ** 30:
        return None
    key = (fb.obj_n, per_obj)
    if getattr(p, '_scores_key', None) != key:
        p._scores = torch.empty(fb.obj_n, per_obj, device=p.dec_in.device)
        p._scores_key = key
    return p._scores


def choose_cfg(M, cout, K, mode=0, use_table=True):
    """(tile config, split-K factor, first split tile [, fallback triple]): the measured table if the shape is in it, otherwise
    minimise (rounds over 256 CUs) x (tile work / efficiency), cutting K when there are too few tiles."""
    global _CFG_TILES
    key = (M, cout, K)
    table = _TABLES[mode]
    if use_table and key in table:
        return table[key]
    _tiles()
    nk = K // (64 if mode == 1 else 32)
    best, best_cost = 0, None
    for c, (bm, bn) in enumerate(_CFG_TILES[:8]):
        if bn > 32 and cout <= 32:
            continue
        if bn >= 128 and cout < 128:
            continue
        blocks = ((M + bm - 1) // bm) * ((cout + bn - 1) // bn)
        rounds = (blocks + 255) // 256
        cost = rounds * bm * bn / _CFG_EFF[(bm, bn)]
        if best_cost is None or cost < best_cost:
            best, best_cost = c, cost
    c = 2 if cout >= 128 else 3                           # 64x128 / 64x64
    bm, bn = _CFG_TILES[c]
    blocks = ((M + bm - 1) // bm) * ((cout + bn - 1) // bn)
    if blocks < 200 and nk >= 16 and cout % 4 == 0 and cout >= 64:
        # too few output tiles for 256 CUs: cut K (measured on MI355X: 28 -> 60 TFLOP/s at M=1620, K=2304)
        ks = max(1, min(16, (448 + blocks - 1) // blocks, nk // 4))
        while ks > 1 and (((nk + ks - 1) // ks) * (ks - 1) >= nk or ks * M * cout > WS_FLOATS):
            ks -= 1
        return (c, ks, 0)
    return (best, 1, 0)


def tune_desc(d, bf, ws, cnt, iters=3, cfg_filter=None, allow_split=True):
    """Time every tile configuration (and its split options) on the convolution descriptor ``d`` -- launched in place, on its
    own buffers -- and return the fastest (cfg, ksplit, split_from).  ``ws`` / ``cnt``: split-K workspace and tile counters."""
    tiles = ops.conv_cfg_tiles()
    key = (d.M, d.Cout, d.KH * d.KW * d.Cin)

    def timeit(c):
        ops.conv2d_launch(d, c, bf)
        torch.cuda.synchronize()
        best_ms = None
        for _ in range(3):                                   # fastest of three batches: robust against a transient
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.conv2d_launch(d, c, bf)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best_ms = ms if best_ms is None else min(best_ms, ms)
        return best_ms

    best, best_t = None, None
    if d.w_batch_rows > 0 and bf == 1:
        # a transform-domain GEMM of the plain-bf16 mode: its operands ARE bf16 in memory (vfn_winograd_input_bf16), which only the
        # persistent kernel reads -- the tiled bf16 kernels would take them for f32 tensors
        for c in ops.wino_gemm_cfg_options(d.w_batch_rows, d.Cout):
            if cfg_filter is not None and not cfg_filter(key, c):
                continue
            t = timeit(c)
            if best_t is None or t < best_t:
                best, best_t = (c, 1, 0), t
        return best
    if d.w_batch_rows > 0 and bf == 0 and d.M // d.w_batch_rows * d.w_batch_rows == d.M:
        # a Winograd-domain GEMM: the persistent kernel's configurations compete with the batched-filter launches below
        for c in ops.wino_gemm_cfg_options(d.w_batch_rows, d.Cout):
            if cfg_filter is not None and not cfg_filter(key, c):
                continue
            apply_choice(d, (c, 1, 0), ws, cnt)
            try:
                t = timeit(c)
            except RuntimeError:
                continue
            if best_t is None or t < best_t:
                best, best_t = (c, 1, 0), t
    if ops.pconv_eligible(d, bf) and os.environ.get('VFN_PCONV', '1') == '1':
        # a 1x1 / stride-1 layer: the persistent kernel with the epilogue in registers competes too
        for c in ops.pconv_cfg_options(d.Cout):
            if cfg_filter is not None and not cfg_filter(key, c):
                continue
            apply_choice(d, (c, 1, 0), ws, cnt)
            try:
                t = timeit(c)
            except RuntimeError:
                continue
            if best_t is None or t < best_t:
                best, best_t = (c, 1, 0), t
    for c, (bm, bn) in enumerate(tiles):
        if bf and c not in ops.BF16_CFGS:              # no LDS-DMA variants (the DMA cannot convert)
            continue
        if cfg_filter is not None and not cfg_filter(key, c):
            continue
        if d.cout_pad < ((d.Cout + bn - 1) // bn) * bn:
            continue
        if (bn > 64 and d.Cout <= 32) or (bn > 128 and d.Cout < 256):
            continue
        blocks = ((d.M + bm - 1) // bm) * ((d.Cout + bn - 1) // bn)
        options = [(c, 1, 0)]
        wk = ops.conv_cfg_wk(c)
        kind = ops.conv_cfg_kind(c)
        if kind == 2:                                  # stream-K balances by itself
            if blocks > ops.SK_MAX_TILES:
                continue
        elif not allow_split:
            pass
        elif kind == 1 and wk > 1:                     # wave-autonomous with K groups: a second split only when tiles are scarce
            if blocks < 256:
                options += [(c, k_, 0) for k_ in ops.valid_splits(d, 8, bf)[1:] if k_ * d.M * d.Cout <= WS_FLOATS]
        elif wk > 1 or ops.conv_cfg_tpb(c) > 1:          # split-K inside the workgroup / two tiles per barrier:
            if blocks > 1024 or d.KH * d.KW * d.Cin // (64 if bf == 1 else 32) < 2 * wk:   # no second split, scarce tiles
                continue
        elif blocks < 256:
            options += [(c, k_, 0) for k_ in ops.valid_splits(d, 16, bf)[1:] if k_ * d.M * d.Cout <= WS_FLOATS]
        else:
            options += [(c, k_, full) for (full, k_, rows) in ops.tail_split_options(d, bm, bn, 8, bf)
                        if k_ * rows * d.Cout <= WS_FLOATS and k_ in (2, 3, 4, 5, 6, 8)]
        for opt in options:
            apply_choice(d, opt, ws, cnt)
            t = timeit(c)
            if best_t is None or t < best_t:
                best, best_t = opt, t
    return best


class ConvLayer:
    """Packed filters + epilogue constants of one convolution (optionally of a slice of its input channels:
    a conv over torch.cat([a, b]) is conv_a(a) + conv_b(b), which lets a shared half be computed once)."""

    def __init__(self, conv, bn=None, device=None, cin_range=None, with_bias=True, reg=None):
        """``reg`` (a refresh.Refresher): the packed filters / epilogue constants are registered there, so that they follow the
        parameters after an optimizer step without being rebuilt."""
        w = conv.weight.detach().float()
        if cin_range is not None:
            w = w[:, cin_range[0]:cin_range[1]].contiguous()
        self.cout, self.cin, self.k, _ = w.shape
        self.stride = conv.stride[0]
        self.pad = conv.padding[0]
        self.w = ops.pad_rows(W.pack_conv_weight(w)).to(device)
        sc, sh = W.conv_epilogue(conv, bn)
        if not with_bias:
            sh = torch.zeros_like(sh)
        self.scale = sc.to(device).contiguous()
        self.shift = sh.to(device).contiguous()
        self._w_lp = {}
        self._wino_src = None
        if reg is not None and isinstance(conv, torch.nn.Module):
            from .refresh import FORWARD
            self._wino_src = [(reg, conv.weight, cin_range[0] if cin_range is not None else 0, 0)]      # (parameter, first channel, first filter row)
            reg.add_filter(conv.weight, self.w, FORWARD, cin=self.cin, cin_off=cin_range[0] if cin_range is not None else 0)
            if bn is not None:
                reg.add_epilogue(self.scale, self.shift if with_bias else None, bn=bn, eps=W.BN_EPS)
            elif conv.bias is not None and with_bias:
                reg.add_epilogue(None, self.shift, bias=conv.bias)

    def refresh_derived(self):
        """The images derived from ``w`` (reduced-precision operands, Winograd filter banks) again, in place."""
        for mode, t in self._w_lp.items():
            if mode == 'wino':
                if self._wino_src is not None:             # (registered with the Refresher: already rewritten)
                    continue
                w = self.w[:self.cout].view(self.cout, 3, 3, self.cin).permute(0, 3, 1, 2)
                t.copy_(ops.pack_winograd_weight(w).to(t.device))
            elif mode == 'wino_bf16':
                w = self.w[:self.cout].view(self.cout, 3, 3, self.cin).permute(0, 3, 1, 2)
                t.copy_(ops.pack_winograd_weight_bf16(w).to(t.device))
            else:
                new = ops.pack_weights_lp(self.w, mode)
                t.copy_(new)

    def w_lp(self, mode):
        """Filters pre-converted for the bf16 / bf16x3 kernels (built once per mode)."""
        if mode not in self._w_lp:
            self._w_lp[mode] = ops.pack_weights_lp(self.w, mode)
        return self._w_lp[mode]

    def w_wino(self):
        """The 36 transform-domain filter banks U = G g G^T of a 3x3 layer (Winograd F(4x4, 3x3), built once)."""
        if 'wino' not in self._w_lp:
            assert self.k == 3
            w = self.w[:self.cout].view(self.cout, 3, 3, self.cin).permute(0, 3, 1, 2)       # packed (kh,kw,cin) -> [Cout,Cin,3,3]
            self._w_lp['wino'] = ops.pack_winograd_weight(w).to(self.w.device)
            if self._wino_src is not None:                 # (follows the parameter after an optimizer step: refresh kind WINO)
                from .refresh import WINO
                for reg, weight, cin_off, row0 in self._wino_src:
                    reg.add_filter(weight, self._w_lp['wino'], WINO, cin=self.cin, cin_off=cin_off, dst_ld=self.cin, dst_row0=row0,
                                   cout_ld=self._w_lp['wino'].shape[0] // 36)
        return self._w_lp['wino']

    def w_wino_bf16(self):
        """... rounded once to bf16: the filter operand of the plain-bf16 mode's Winograd layers (vfn_winograd_gemm_bf16)."""
        if 'wino_bf16' not in self._w_lp:
            assert self.k == 3
            w = self.w[:self.cout].view(self.cout, 3, 3, self.cin).permute(0, 3, 1, 2)
            self._w_lp['wino_bf16'] = ops.pack_winograd_weight_bf16(w).to(self.w.device)
        return self._w_lp['wino_bf16']


class Pred2Layer:
    """A 3x3 convolution with two filters (pred2 / local_pred2, AFB_URR.py:195,202) as a tap GEMM: the 9 taps x 2 filters
    become 18 (padded to 20) 1x1 filters, so the input is read once on the matrix cores; ``ops.pred2_gather`` then adds
    the nine shifted taps and the bias."""
    TAPS = 20

    def __init__(self, conv, device, reg=None):
        w = conv.weight.detach().float().cpu()                 # [2, Cin, 3, 3]
        assert w.shape[0] == 2 and w.shape[2:] == (3, 3)
        self.cin = w.shape[1]
        w18 = torch.zeros(self.TAPS, self.cin)
        for dy in range(3):
            for dx in range(3):
                for o in range(2):
                    w18[(dy * 3 + dx) * 2 + o] = w[o, :, dy, dx]
        self.cout, self.k, self.stride, self.pad = self.TAPS, 1, 1, 0
        self.w = ops.pad_rows(w18.contiguous()).to(device)
        self.scale = torch.ones(self.TAPS, device=device)
        self.shift = torch.zeros(self.TAPS, device=device)
        self.bias = conv.bias.detach().float().to(device).contiguous().clone()
        self._w_lp = {}
        self._wino_src = None
        if reg is not None:
            from .refresh import TAPS
            reg.add_filter(conv.weight, self.w, TAPS)
            reg.add_epilogue(None, self.bias, bias=conv.bias)

    refresh_derived = ConvLayer.refresh_derived

    def w_lp(self, mode):
        if mode not in self._w_lp:
            self._w_lp[mode] = ops.pack_weights_lp(self.w, mode)
        return self._w_lp[mode]


class Launch:
    """One pre-built kernel launch."""
    __slots__ = ('fn', 'args', 'name', 'flops')

    def __init__(self, fn, args, name, flops=0.0):
        self.fn, self.args, self.name, self.flops = fn, args, name, flops

    def __call__(self):
        self.fn(*self.args)


# HIP graphs for the fixed-shape launch lists (VERDICT r4 "missing" 4).  A plan's lists -- the memory
# encoder, the frame-only query side (first / second half), the bank-dependent decoder per slot -- are sequences of launches whose
# descriptors, buffers and grids never change once the tile choices are settled: each is captured ONCE into a hipGraph (through
# torch.cuda.CUDAGraph: the launches go out through ctypes on torch's current stream, which is the capturing stream inside the context)
# and replayed with ONE host call instead of 20-60.  The memory read and the bank update stay eager: their grids follow the bank's
# length.  What it buys is host time -- the HBM-resident C2 loop is GPU-bound either way, `video_seg.main` (files to files) and the
# 2-ms frames of C3 in bf16 are paced by the launching thread.  VFN_GRAPHS=0 switches it off; instrumented runs (bench.py's sampled
# frames, Engine.eager = True) take the eager path; training plans are never captured (their lists change with the sample).
_GRAPHS = os.environ.get('VFN_GRAPHS', '1') == '1'


class GraphCache:
    """Captured graphs of one plan's launch lists, keyed by (id of the list, first index, last index)."""

    def __init__(self):
        self.graphs = {}
        self.runs = {}                     # key -> eager runs so far (a list is captured on its THIRD run: tile choices, lazy
                                           # hipFuncSetAttribute calls and first-use tuning are behind it by then)
        self.captures = 0                  # captures performed so far (bench.py reports how many fell inside its timed region:
                                           # torch.cuda.graph() enters with a device-wide synchronize + gc.collect + empty_cache)

    def invalidate(self):
        self.graphs.clear()
        self.runs.clear()

    def run(self, lst, lo=0, hi=None, eager=False):
        hi = len(lst) if hi is None else hi
        if hi <= lo:
            return
        key = (id(lst), lo, hi)
        g = None if (eager or not _GRAPHS) else self.graphs.get(key)
        if g is not None:
            g.replay()
            return
        n = self.runs.get(key, 0)
        if eager or not _GRAPHS or n < 2 or hi - lo < 4:
            for l in lst[lo:hi]:
                l()
            if not eager:
                self.runs[key] = n + 1
            return
        # capture: the list runs once more inside the capture (that IS this call's execution: capture records, replay executes)
        g = torch.cuda.CUDAGraph()
        cur = torch.cuda.current_stream()
        cap = torch.cuda.Stream(device=cur.device)
        cap.wait_stream(cur)
        try:
            with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):   # (writer / loader threads keep making HIP calls)
                for l in lst[lo:hi]:
                    l()
        except Exception:                  # (a runtime that cannot capture some launch: stay eager for this list, loudly once)
            import warnings
            warnings.warn('vfloodnet_amd: HIP graph capture of a launch list failed; running it eagerly')
            self.runs[key] = -10 ** 9
            # the context manager has ended the capture on its way out; if the capture stream is somehow still recording, running
            # the list "eagerly" would only extend a dead graph -- fail loudly instead of returning without having computed anything
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('vfloodnet_amd: the stream is still capturing after a failed HIP graph capture')
            cur.wait_stream(cap)
            for l in lst[lo:hi]:
                l()
            return
        cur.wait_stream(cap)
        self.graphs[key] = g
        self.captures += 1
        g.replay()

    def warm(self, lst, lo=0, hi=None):
        """Run a list until it is captured (three runs), outside any timed / latency-sensitive loop (Engine.capture)."""
        hi = len(lst) if hi is None else hi
        if not _GRAPHS or hi - lo < 4:
            return
        key = (id(lst), lo, hi)
        while key not in self.graphs and self.runs.get(key, 0) >= 0:
            self.run(lst, lo, hi)


class FramePlan:
    """Buffers + launch lists for one (H0, W0, obj_n)."""

    def __init__(self, eng, H0, W0, obj_n, keep_acts=False):
        self.eng = eng
        self.graphs = GraphCache()
        self.keep_acts = keep_acts            # training: every bottleneck keeps its own activation buffers (the backward reads them)
        self.acts_m = {}                      # memory encoder: (stage, block) -> dict(x, t1, t2, ds, out, stride, H, W)
        self.H0, self.W0, self.obj_n = H0, W0, obj_n
        self.pad, self.Hp, self.Wp = pad_divide_by(H0, W0)
        dev = eng.device
        self.h2, self.w2 = self.Hp // 2, self.Wp // 2
        self.h4, self.w4 = self.Hp // 4, self.Wp // 4
        self.h8, self.w8 = self.Hp // 8, self.Wp // 8
        self.h16, self.w16 = self.Hp // 16, self.Wp // 16
        self.HW = self.h16 * self.w16
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        K = obj_n
        self.frame_in = f(3, H0, W0)          # memorize() input
        self.mask_in = f(K, H0, W0)
        # query side: two sets of frame-only state, two frames (slots) each -- see QuerySet
        self.qsets = [QuerySet(self), QuerySet(self)]
        self._qbatch = {}
        # memory encoder
        self.m = self._trunk_buffers(K)
        self.kv_m = f(K, self.HW, DK + DV)
        # memory read
        self.ml = f(K, self.HW, 2)
        self.ml_part = f(K, MAX_SPLIT_SCAN, self.HW, 2)
        self.work = torch.zeros(4, dtype=torch.int32, device=dev)       # queue head of the persistent scan kernel
        self.o_part = f(K, MAX_SPLIT, self.HW, DV)
        self.dec_in = f(K, self.h16, self.w16, DV)          # memory read-out only; the query-value half of
                                                            # cat([mem, q_out]) goes through its own conv (QuerySet.fm_q)
        # decoder
        self.d16 = [f(K, self.h16, self.w16, 256) for _ in range(3)]
        self.d8 = [f(K, self.h8, self.w8, 256) for _ in range(3)]
        self.d4 = [f(K, self.h4, self.w4, 256) for _ in range(3)]
        self.pp = f(K, self.h4, self.w4, 2)
        self.z4 = f(K, self.h4, self.w4, Pred2Layer.TAPS)     # tap products of pred2 / local_pred2
        self.z2 = f(K, self.h2, self.w2, Pred2Layer.TAPS)
        self.p_up = f(K, self.h2, self.w2, 2)
        self.rough = f(K, self.h2, self.w2)
        self.unc = f(self.h2, self.w2)
        fused_ok = K <= 4                                    # ops.local_stats: one fused pass, no scratch
        self.hs = None if fused_ok else f(K, self.h2, self.w2, 64)
        self.hr = None if fused_ok else f(K, self.h2, self.w2)
        self.hm = None if fused_ok else f(K, self.h2, self.w2)
        self.lm = f(K, self.h2, self.w2, 64)                 # r1_local
        self.conf = f(K, self.h2, self.w2)
        self.l2 = [f(K, self.h2, self.w2, 32) for _ in range(3)]
        self.qq = f(K, self.h2, self.w2, 2)
        self.score = f(1, K, H0, W0)
        self._twins = {}                      # storage -> split-bf16 image buffer (bf16x3 mode)
        self._lp_state = {}                   # (ptr, shape) of an f32 view -> its twin holds the image of relu(x)? (True / False)
        self.ws = f(WS_FLOATS)
        self.ws_q = f(WS_FLOATS)              # split-K workspace of the query-encoder list (side stream)
        self.cnt = torch.zeros(ops.SK_MAX_TILES, dtype=torch.int32, device=dev)      # split-tile arrival counters (zero at rest)
        self.cnt_q = torch.zeros(ops.SK_MAX_TILES, dtype=torch.int32, device=dev)
        self._ws_cur, self._cnt_cur = self.ws, self.cnt
        self._wino = {}                       # id(workspace of the launch list's stream) -> (V, M) scratch of the Winograd layers

        self.mem = []         # memorize
        self._build()

    # (single-frame views kept for tools and tests: slot 0 of set 0)
    @property
    def seg_pre(self):
        return self.qsets[0].pre[1]

    @property
    def seg_post(self):
        return self.qsets[0].post[0]

    @property
    def kv_q(self):
        return self.qsets[0].kv_q[0:1]

    def all_lists(self):
        """Every launch list of the plan (autotune, bench instrumentation)."""
        out = [self.mem]
        for qs in self.qsets:
            out += [qs.pre[1], qs.pre[2], qs.post[0], qs.post[1]]
        for qs in self._qbatch.values():
            out += [qs.pre[qs.nq]]
            if qs._dec_batch is not None:
                out += [qs._dec_batch.post]
        return out

    def batch_set(self, n):
        """Training: a query set for the n frames of a sample (frame-only part in ONE pass over the batch, Engine.query_batch);
        built on first use, next to the two sets of the inference loop."""
        qs = self._qbatch.get(n)
        if qs is None:
            qs = self._qbatch[n] = QuerySet(self, nq=n)
            qs.build()
        return qs

    def _trunk_buffers(self, N):
        dev = self.eng.device
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        b = {}
        b['r1'] = f(N, self.h2, self.w2, 64)
        b['x4'] = f(N, self.h4, self.w4, 64)
        for name, (h, w, planes) in {'res2': (self.h4, self.w4, 64), 'res3': (self.h8, self.w8, 128),
                                     'res4': (self.h16, self.w16, 256)}.items():
            # t1 may live at the input resolution when the block strides (conv1 runs before the strided 3x3)
            hin, win = (h * 2, w * 2) if name != 'res2' else (h, w)
            b[name] = dict(t1a=f(N, hin, win, planes), t1=f(N, h, w, planes), t2=f(N, h, w, planes),
                           ds=f(N, h, w, planes * 4), o=[f(N, h, w, planes * 4) for _ in range(2)],
                           out=f(N, h, w, planes * 4))
        return b

    # ------------------------------------------------------------------ builders
    def _lp_args(self, out):
        """Extra arguments of ops.upsample2x_add in bf16x3 mode: the image of relu(out) for the ResBlock behind it."""
        if not self.eng.any_x3 or out.shape[-1] % 32:
            return ()
        self._lp_state[(out.data_ptr(), tuple(out.shape))] = True
        return (self.lp_twin(out), True)

    def lp_twin(self, t):
        """The split-bf16 image that travels with an f32 activation buffer in bf16x3 mode (vfn_conv_desc.in_lp / out_lp):
        same bytes, same strides -- a view of ``t`` maps onto the same view of the twin."""
        st = t.untyped_storage()
        key = st.data_ptr()
        if key not in self._twins:
            self._twins[key] = torch.empty(st.nbytes() // 4, dtype=torch.float32, device=t.device)
        return torch.as_strided(self._twins[key], t.size(), t.stride(), t.storage_offset())

    def has_lp(self, t, relu):
        """Does the twin of ``t`` hold the image of t (relu=False) / of relu(t) (relu=True)?"""
        return self._lp_state.get((t.data_ptr(), tuple(t.shape))) == bool(relu)

    def _conv(self, lst, layer, x, out, N, H, Wd, res=None, relu_in=False, relu_out=False, name='conv',
              in_ld=None, out_ld=None, res_mod=0, lp_out=None, f32_out=True):
        """``lp_out`` ('plain' / 'relu', bf16x3 mode only): the epilogue also writes the split-bf16 image of the result (of
        its ReLU) into the twin of ``out``; ``f32_out=False`` then drops the f32 tensor when only convolutions consume
        it.  An input whose twin holds the image this layer needs (has_lp) is staged from the twin without conversion."""
        bf = self.eng.layer_mode(name)
        lp = bf == 2
        if self.eng.mixed:
            f32_out = True                                    # (a consumer in another mode reads the f32 tensor)
        if ((bf == 0 or (bf == 1 and _WINOGRAD_LP and not self.keep_acts and layer.cin % 64 == 0)) and in_ld is None and
                (_WINOGRAD_TRAIN or not self.keep_acts) and x.shape[-1] == layer.cin and self.eng.use_winograd(layer, N * H * Wd, bf)):
            return self._conv_winograd(lst, layer, x, out, N, H, Wd, res, relu_in, relu_out, name, out_ld, res_mod, bf)
        d = ops.make_conv_desc(x, layer.w, layer.cout, layer.k, layer.k, layer.stride, layer.pad, out,
                               layer.scale, layer.shift, res, relu_in, relu_out,
                               cin=layer.cin, in_ld=in_ld if in_ld is not None else x.shape[-1],
                               out_ld=out_ld, N=N, H=H, W=Wd)
        d.res_mod = int(res_mod)
        if lp and in_ld is None and layer.cin % 32 == 0 and self.has_lp(x, relu_in):
            d.inp, d.in_lp, d.relu_in = ptr(self.lp_twin(x)), 1, 0
        if lp and lp_out and layer.cout % 32 == 0 and d.out_ld % 32 == 0:
            d.out_lp, d.out_lp_relu = ptr(self.lp_twin(out)), int(lp_out == 'relu')
            self._lp_state[(out.data_ptr(), tuple(out.shape))] = (lp_out == 'relu')
            if not f32_out:
                d.out = None
        else:
            self._lp_state.pop((out.data_ptr(), tuple(out.shape)), None)      # (an f32-only producer: the twin is stale)
        K = layer.k * layer.k * layer.cin
        if bf == 1 and layer.cin % 64:                     # (the 32-channel local head: no 64-channel K tile)
            bf = 2
        if bf:
            ops.use_packed_weights(d, layer.w_lp(bf))
        choice = choose_cfg(d.M, layer.cout, K, bf)
        if choice[1] > 1 and (d.out_ld % 4 or (res is not None and d.res_ld % 4)):
            choice = (choice[0], 1, 0)
        cfg = apply_choice(d, choice, self._ws_cur, self._cnt_cur)
        lst.append(Launch(ops.conv2d_launch, (d, cfg, bf), f'{name}[{d.M}x{layer.cout}x{K}]', 2.0 * d.M * layer.cout * K))
        return out

    def _conv_winograd(self, lst, layer, x, out, N, H, Wd, res, relu_in, relu_out, name, out_ld, res_mod, bf=0):
        """A 3x3 / stride-1 layer as Winograd F(4x4, 3x3): input transform, the 36 transform-domain GEMMs as ONE batched-filter
        launch of the convolution kernels (tile choice from the same tables / tuner, keyed by the GEMM's shape), output transform
        with the layer's epilogue (csrc/conv_winograd.hip)."""
        rows = ops.winograd_rows(N, H, Wd)
        key = id(self._ws_cur)
        need_v, need_m = 36 * rows * layer.cin, 36 * rows * layer.cout
        V, Mb = self._wino.get(key, (None, None))
        if V is None or V.numel() < need_v or Mb.numel() < need_m:
            dev = self.eng.device
            V = torch.zeros(max(need_v, V.numel() if V is not None else 0), device=dev)
            Mb = torch.empty(max(need_m, Mb.numel() if Mb is not None else 0), device=dev)
            self._wino[key] = (V, Mb)                      # (launches built earlier keep their own, smaller buffers alive)
        Mb = Mb[:need_m].view(36 * rows, layer.cout)
        if bf == 1:
            # the plain-bf16 mode (round 5): V is written as bf16 by the input transform (f32 arithmetic, one rounding -- what the bf16
            # convolution does to its operands as it stages them), the filter banks are bf16, the persistent GEMM accumulates in f32
            V = V.view(torch.bfloat16)[:need_v].view(36 * rows, layer.cin)
            U = layer.w_wino_bf16()
        else:
            V = V[:need_v].view(36 * rows, layer.cin)
            U = layer.w_wino()
        lst.append(Launch(ops.winograd_input, (x, V, rows, relu_in, N, H, Wd, layer.cin, x.shape[-1]), name + '.wino_in'))
        dg = ops.make_winograd_gemm_desc(V, U, Mb, rows, layer.cin, layer.cout)
        if bf:
            choice = _TABLES[bf].get((dg.M, layer.cout, layer.cin))
            if choice is None or choice[0] < ops.WINO_GEMM_CFG0:
                tc = 5 if rows % 64 == 0 and layer.cout >= 128 and rows >= 1024 else 3
                choice = (ops.wino_gemm_cfg(tc, 512), 1, 0)
        else:
            choice = choose_cfg(dg.M, layer.cout, layer.cin, 0)
        cfg = apply_choice(dg, choice, self._ws_cur, self._cnt_cur)
        lst.append(Launch(ops.conv2d_launch, (dg, cfg, bf), f'{name}.wino_gemm[{dg.M}x{layer.cout}x{layer.cin}]', 2.0 * dg.M * layer.cout * layer.cin))
        self._lp_state.pop((out.data_ptr(), tuple(out.shape)), None)
        lst.append(Launch(ops.winograd_output, (Mb, rows, out, N, H, Wd, layer.cout, layer.scale, layer.shift, res,
                                                res.shape[-1] if res is not None else 0, res_mod, relu_out,
                                                out_ld if out_ld is not None else out.shape[-1]), name + '.wino_out'))
        return out

    def _trunk(self, lst, enc, bufs, N, prefix, acts=None):
        x = bufs['x4']
        lst.append(Launch(ops.maxpool3x3s2, (bufs['r1'], x), prefix + '.maxpool'))
        H, Wd = self.h4, self.w4
        for lname in ('res2', 'res3', 'res4'):
            blocks = enc[lname]
            lb = bufs[lname]
            for bi, blk in enumerate(blocks):
                s = blk['conv2'].stride
                Ho, Wo = H // s, Wd // s
                t1 = lb['t1a'] if (s == 2) else lb['t1']
                t2, ds_buf = lb['t2'], lb['ds']
                if self.keep_acts:                        # own buffers per block (sliced like the shared ones)
                    t1, t2 = torch.empty_like(t1), torch.empty_like(lb['t2'])
                    ds_buf = torch.empty_like(lb['ds'])
                nm = f'{prefix}.{lname}.{bi}'
                # (bf16x3: t1 / t2 feed one convolution each -> image only; a block's output is the next block's residual
                # (f32) and the next convolutions' operand (image))
                self._conv(lst, blk['conv1'], x, t1, N, H, Wd, relu_out=True, name=nm + '.conv1', lp_out='plain', f32_out=False)
                self._conv(lst, blk['conv2'], t1, t2, N, H, Wd, relu_out=True, name=nm + '.conv2', lp_out='plain',
                           f32_out=False)
                if 'down' in blk:
                    self._conv(lst, blk['down'], x, ds_buf, N, H, Wd, name=nm + '.down')
                    idn = ds_buf
                else:
                    idn = x
                out = lb['out'] if bi == len(blocks) - 1 else lb['o'][bi % 2]
                if self.keep_acts and bi != len(blocks) - 1:
                    out = torch.empty_like(out)
                self._conv(lst, blk['conv3'], t2, out, N, Ho, Wo, res=idn, relu_out=True, name=nm + '.conv3',
                           lp_out='plain')
                if self.keep_acts and acts is not None:
                    acts[(lname, bi)] = dict(x=x, t1=t1, t2=t2, ds=ds_buf if 'down' in blk else None, out=out, stride=s, H=H, W=Wd)
                x = out
                H, Wd = Ho, Wo
        return x

    def _resblock(self, lst, rb, x, t, out, N, H, Wd, name):
        # (bf16x3: t feeds conv2 only, through its ReLU -> the image of relu(t) and no f32 tensor)
        self._conv(lst, rb['conv1'], x, t, N, H, Wd, relu_in=True, name=name + '.conv1', lp_out='relu', f32_out=False)
        self._conv(lst, rb['conv2'], t, out, N, H, Wd, res=x, relu_in=True, name=name + '.conv2')
        return out

    def _build(self):
        e = self.eng
        K = self.obj_n
        for qs in self.qsets:
            qs.build()
        # ---- memorize
        self._ws_cur, self._cnt_cur = self.ws, self.cnt
        self.stem_m = ops.make_stem_desc(self.frame_in, self.mask_in, e.stem_m_w, e.stem_m_scale, e.stem_m_shift,
                                         self.m['r1'], e.mean, e.std, K, self.H0, self.W0, self.pad, self.Hp, self.Wp)
        self.mem.append(Launch(ops.stem_launch, (self.stem_m,), 'encoder_m.stem',
                               2.0 * K * self.h2 * self.w2 * 64 * 245))
        r4m = self._trunk(self.mem, e.enc_m, self.m, K, 'encoder_m', acts=self.acts_m)
        self._conv(self.mem, e.keyval, r4m, self.kv_m, K, self.h16, self.w16, name='keyval')

    def conv_flops(self, which):
        return sum(l.flops for l in getattr(self, which))


class QuerySet:
    """Frame-only state of the query side for up to TWO frames (slots 0 / 1 = the batch dimension): query encoder, KeyValue
    and the decoder branches that depend on the frame alone (AFB_URR.py:122-127: ResFS(convFS(f)); the query-value half of
    convFM; the r1 half of local_convFM).  None of it depends on the bank, so it is computed ahead of the loop for the NEXT
    two frames in one pass -- every GEMM sees twice the rows (M) with the same filters -- on a side stream underneath
    memorize / update of the frames before (Engine.prefetch_begin / prefetch_finish).  Two sets alternate: one is consumed
    by the decoder while the other is being filled.

    ``pre[n]``: launch list for n frames (n = 1: slot 0 only); ``post[slot]``: the bank-dependent decoder reading slot."""

    def __init__(self, plan, nq=2):
        """``nq``: frames the set holds (2 in the inference loop; the training step batches the T - 1 frames of a sample)."""
        self.plan = plan
        self.nq = nq
        p = plan
        f = lambda *s_: torch.empty(*s_, device=p.eng.device, dtype=torch.float32)
        self.frames = f(nq, 3, p.H0, p.W0)
        self.q = p._trunk_buffers(nq)
        self.kv_q = f(nq, p.HW, DK + DV)
        self.fm_q = f(nq, p.h16, p.w16, 256)
        self.s8 = [f(nq, p.h8, p.w8, 256) for _ in range(3)]
        self.s4 = [f(nq, p.h4, p.w4, 256) for _ in range(3)]
        self.lq = f(nq, p.h2, p.w2, 32)                      # local_convFM over the shared r1 half
        self.sizes = sorted({1, nq})                        # batch sizes with a launch list of their own
        self.pre = {n: [] for n in self.sizes}
        self.acts = {n: {} for n in self.sizes}             # training plans: the bottlenecks' activations per launch list
        self.post = [[] for _ in range(nq)]
        self.split = {n: 0 for n in self.sizes}             # pre[n][:split[n]] = the first half (by estimated time)
        # bookkeeping of Engine.prefetch_*: which frames the slots hold
        self.keys = [None] * nq
        self.held = [None] * nq                             # the frames behind ``keys``: a key is an allocator address, so
                                                            # the set keeps its frames alive until they are consumed
        self.consumed = [True] * nq
        self.stage = 0                                      # 0 idle, 1 first half enqueued, 2 complete
        self.n = 0
        self.done = None
        self._dec_batch = None                              # training: DecoderBatch over the nq frames (built on first use)

    def dec_batch(self):
        if self._dec_batch is None:
            self._dec_batch = DecoderBatch(self.plan, self)
        return self._dec_batch

    @staticmethod
    def _sl(t, n):
        return t[0:n]

    def build(self):
        p, e = self.plan, self.plan.eng
        K = p.obj_n
        D = e.dec
        for n in self.sizes:
            P = self.pre[n]
            sl = lambda t: t[0:n]
            q = {k: (sl(v) if torch.is_tensor(v) else {kk: ([sl(x) for x in vv] if isinstance(vv, list) else sl(vv)) for kk, vv in v.items()})
                 for k, v in self.q.items()}
            p._ws_cur, p._cnt_cur = p.ws_q, p.cnt_q
            for i in range(n):                             # the stem takes one frame per launch
                d = ops.make_stem_desc(self.frames[i], None, e.stem_q_w, e.stem_q_scale, e.stem_q_shift,
                                       self.q['r1'][i:i + 1], e.mean, e.std, 1, p.H0, p.W0, p.pad, p.Hp, p.Wp)
                P.append(Launch(ops.stem_launch, (d,), 'encoder_q.stem', 2.0 * p.h2 * p.w2 * 64 * 147))
            r4 = p._trunk(P, e.enc_q, q, n, 'encoder_q', acts=self.acts[n])
            p._conv(P, e.keyval, r4, sl(self.kv_q), n, p.h16, p.w16, name='keyval')
            # convFM(cat([mem_i, q_out])) = convFM[:, :512](mem_i) + convFM[:, 512:](q_out): the second term is the
            # same for every object (AFB_URR.py:159,176) -> computed once (with the bias) and added as a shared residual
            kvq_val = sl(self.kv_q)[:, :, DK:]                  # [n, HW, 512] view, pixel stride 640
            p._conv(P, D['convFM_q'], kvq_val, sl(self.fm_q), n, p.h16, p.w16, name='decoder.convFM.q', in_ld=DK + DV)
            s8, s4 = [sl(t) for t in self.s8], [sl(t) for t in self.s4]
            p._conv(P, D['RF3']['convFS'], q['res3']['out'], s8[0], n, p.h8, p.w8, name='decoder.RF3.convFS', lp_out='relu')
            p._resblock(P, D['RF3']['ResFS'], s8[0], s8[1], s8[2], n, p.h8, p.w8, 'decoder.RF3.ResFS')
            p._conv(P, D['RF2']['convFS'], q['res2']['out'], s4[0], n, p.h4, p.w4, name='decoder.RF2.convFS', lp_out='relu')
            p._resblock(P, D['RF2']['ResFS'], s4[0], s4[1], s4[2], n, p.h4, p.w4, 'decoder.RF2.ResFS')
            # local_convFM(cat([r1, r1_local])): the r1 half is shared by the objects (AFB_URR.py:231-232)
            p._conv(P, D['local_convFM_r1'], q['r1'], sl(self.lq), n, p.h2, p.w2, name='decoder.local_convFM.r1')
            est = [l.flops / 100e12 + 6e-6 for l in P]
            half, acc = 0.5 * sum(est), 0.0
            for i, t_ in enumerate(est):
                acc += t_
                if acc >= half:
                    self.split[n] = i + 1
                    break
        p._ws_cur, p._cnt_cur = p.ws, p.cnt
        # ---- decoder, bank-dependent part, once per slot
        d16, d8, d4 = p.d16, p.d8, p.d4
        for slot in range(self.nq):
            L = self.post[slot]
            o = lambda t: t[slot:slot + 1]
            p._conv(L, D['convFM_m'], p.dec_in, d16[0], K, p.h16, p.w16, res=o(self.fm_q), res_mod=p.HW,
                    name='decoder.convFM.mem', lp_out='relu')
            p._resblock(L, D['ResMM'], d16[0], d16[1], d16[2], K, p.h16, p.w16, 'decoder.ResMM')
            L.append(Launch(ops.upsample2x_add, (o(self.s8[2]), d16[2], d8[0], True) + p._lp_args(d8[0]), 'decoder.RF3.up_add'))
            p._resblock(L, D['RF3']['ResMM'], d8[0], d8[1], d8[2], K, p.h8, p.w8, 'decoder.RF3.ResMM')
            L.append(Launch(ops.upsample2x_add, (o(self.s4[2]), d8[2], d4[0], True) + p._lp_args(d4[0]), 'decoder.RF2.up_add'))
            p._resblock(L, D['RF2']['ResMM'], d4[0], d4[1], d4[2], K, p.h4, p.w4, 'decoder.RF2.ResMM')
            p._conv(L, D['pred2'], d4[2], p.z4, K, p.h4, p.w4, relu_in=True, name='decoder.pred2.taps')
            L.append(Launch(ops.pred2_gather, (p.z4, D['pred2'].bias, p.pp), 'decoder.pred2.gather'))
            L.append(Launch(ops.rough_uncertainty, (p.pp, p.p_up, p.rough, p.unc), 'decoder.rough_unc'))
            L.append(Launch(ops.local_stats, (self.q['r1'][slot], p.rough, p.hs, p.hr, p.hm, p.lm, p.conf),
                            'decoder.local_stats'))
            l2 = p.l2
            p._conv(L, D['local_convFM_loc'], p.lm, l2[0], K, p.h2, p.w2, res=o(self.lq),
                    res_mod=p.h2 * p.w2, name='decoder.local_convFM.local', lp_out='relu')
            p._resblock(L, D['local_ResMM'], l2[0], l2[1], l2[2], K, p.h2, p.w2, 'decoder.local_ResMM')
            p._conv(L, D['local_pred2'], l2[2], p.z2, K, p.h2, p.w2, relu_in=True, name='decoder.local_pred2.taps')
            L.append(Launch(ops.pred2_gather, (p.z2, D['local_pred2'].bias, p.qq), 'decoder.local_pred2.gather'))
            L.append(Launch(ops.final_logits, (p.p_up, p.unc, p.conf, p.qq, p.score, p.pad, p.H0, p.W0), 'decoder.final_logits'))


class DecoderBatch:
    """Training: the bank-dependent half of the decoder for ALL G frames of a sample in one pass (round 5).  The bank is fixed while
    a sample's frames are segmented (train_video_seg.py:65-69), so after ``Engine.query_batch`` the G memory read-outs exist
    together and the decoder's convolutions run once over G * obj_n images (sample-major: image g * obj_n + k) instead of G times
    over obj_n -- the 1/16-resolution layers see 6 250 pixels instead of 1 250, five times fewer launches on the dependent chain and,
    in the backward pass, five times fewer (and larger) weight-gradient launches.  What mixes the objects of ONE frame (the shared
    residuals of convFM / local_convFM, the skip additions, uncertainty, the local statistics, the final softmax) keeps its kernels and
    runs once per frame on that frame's images.  Attributes the backward pass reads (d16, d8, d4, dec_in, pp, p_up, rough, unc, lm,
    conf, l2, qq, score) have the plan's names; geometry, workspaces and everything else fall through to the plan."""

    def __init__(self, plan, qs):
        self.plan, self.qs, self.G = plan, qs, qs.nq
        p, G, K = plan, qs.nq, plan.obj_n
        N = G * K
        self.N = N
        f = lambda *s_: torch.empty(*s_, device=p.eng.device, dtype=torch.float32)
        self.dec_in = f(N, p.h16, p.w16, DV)
        self.d16 = [f(N, p.h16, p.w16, 256) for _ in range(3)]
        self.d8 = [f(N, p.h8, p.w8, 256) for _ in range(3)]
        self.d4 = [f(N, p.h4, p.w4, 256) for _ in range(3)]
        self.pp = f(N, p.h4, p.w4, 2)
        self.z4 = f(N, p.h4, p.w4, Pred2Layer.TAPS)
        self.z2 = f(N, p.h2, p.w2, Pred2Layer.TAPS)
        self.p_up = f(N, p.h2, p.w2, 2)
        self.rough = f(N, p.h2, p.w2)
        self.unc = f(G, p.h2, p.w2)
        self.lm = f(N, p.h2, p.w2, 64)
        self.conf = f(N, p.h2, p.w2)
        self.l2 = [f(N, p.h2, p.w2, 32) for _ in range(3)]
        self.qq = f(N, p.h2, p.w2, 2)
        self.score = f(G, K, p.H0, p.W0)
        # the memory read of all G frames in one pass: G * HW query columns against the bank, read-out object-major [K, G * HW, 512]
        # (then one strided copy into the frame-major ``dec_in``); its own statistics / partial buffers, sized by G * HW
        self._mr = None                       # (built on first use: the inference loop's groups read frame by frame)
        self.post = []
        self._build()

    @property
    def mr(self):
        if self._mr is None:
            import types
            p, G, K = self.plan, self.G, self.plan.obj_n
            f = lambda *s_: torch.empty(*s_, device=p.eng.device, dtype=torch.float32)
            HWb = G * p.HW
            self._mr = types.SimpleNamespace(HW=HWb, ml=f(K, HWb, 2), ml_part=f(K, MAX_SPLIT_SCAN, HWb, 2), o_part=f(K, MAX_SPLIT, HWb, DV),
                                             work=torch.zeros(4, dtype=torch.int32, device=p.eng.device), dec_in=f(K, HWb, DV))
        return self._mr

    def __getattr__(self, name):               # (only what is not set above: geometry, workspaces, obj_n, ...)
        return getattr(self.plan, name)

    def grp(self, t, g):
        """The images of frame g in a sample-major tensor."""
        K = self.plan.obj_n
        return t[g * K:(g + 1) * K]

    def _build(self):
        p, qs, G = self.plan, self.qs, self.G
        K, N = p.obj_n, self.N
        D = p.eng.dec
        L = self.post
        grp = self.grp
        p._ws_cur, p._cnt_cur = p.ws, p.cnt
        d16, d8, d4, l2 = self.d16, self.d8, self.d4, self.l2
        for g in range(G):                      # convFM: the query-value half is frame g's shared residual
            p._conv(L, D['convFM_m'], grp(self.dec_in, g), grp(d16[0], g), K, p.h16, p.w16, res=qs.fm_q[g:g + 1], res_mod=p.HW,
                    name='decoder.convFM.mem', lp_out='relu')
        p._resblock(L, D['ResMM'], d16[0], d16[1], d16[2], N, p.h16, p.w16, 'decoder.ResMM')
        for g in range(G):
            L.append(Launch(ops.upsample2x_add, (qs.s8[2][g:g + 1], grp(d16[2], g), grp(d8[0], g), True), 'decoder.RF3.up_add'))
        p._resblock(L, D['RF3']['ResMM'], d8[0], d8[1], d8[2], N, p.h8, p.w8, 'decoder.RF3.ResMM')
        for g in range(G):
            L.append(Launch(ops.upsample2x_add, (qs.s4[2][g:g + 1], grp(d8[2], g), grp(d4[0], g), True), 'decoder.RF2.up_add'))
        p._resblock(L, D['RF2']['ResMM'], d4[0], d4[1], d4[2], N, p.h4, p.w4, 'decoder.RF2.ResMM')
        p._conv(L, D['pred2'], d4[2], self.z4, N, p.h4, p.w4, relu_in=True, name='decoder.pred2.taps')
        L.append(Launch(ops.pred2_gather, (self.z4, D['pred2'].bias, self.pp), 'decoder.pred2.gather'))
        for g in range(G):
            L.append(Launch(ops.rough_uncertainty, (grp(self.pp, g), grp(self.p_up, g), grp(self.rough, g), self.unc[g]), 'decoder.rough_unc'))
            L.append(Launch(ops.local_stats, (qs.q['r1'][g], grp(self.rough, g), p.hs, p.hr, p.hm, grp(self.lm, g), grp(self.conf, g)),
                            'decoder.local_stats'))
            p._conv(L, D['local_convFM_loc'], grp(self.lm, g), grp(l2[0], g), K, p.h2, p.w2, res=qs.lq[g:g + 1],
                    res_mod=p.h2 * p.w2, name='decoder.local_convFM.local', lp_out='relu')
        p._resblock(L, D['local_ResMM'], l2[0], l2[1], l2[2], N, p.h2, p.w2, 'decoder.local_ResMM')
        p._conv(L, D['local_pred2'], l2[2], self.z2, N, p.h2, p.w2, relu_in=True, name='decoder.local_pred2.taps')
        L.append(Launch(ops.pred2_gather, (self.z2, D['local_pred2'].bias, self.qq), 'decoder.local_pred2.gather'))
        for g in range(G):
            L.append(Launch(ops.final_logits, (grp(self.p_up, g), self.unc[g], grp(self.conf, g), grp(self.qq, g), self.score[g:g + 1],
                                               p.pad, p.H0, p.W0), 'decoder.final_logits'))


class Engine:
    def __init__(self, model):
        dev = next(model.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('AFB_URR runs on hand-written HIP kernels only: move the model to the GPU '
                               f'(model.to("cuda")); got {dev}.  There is no CPU fallback.')
        _lib.lib()
        self.device = dev
        self.model = model
        self.plans = {}
        self.precision = getattr(model, 'precision', 'fp32')
        if self.precision not in ops.MODES:
            raise ValueError(f"precision must be one of {sorted(ops.MODES)}, got {self.precision!r}")
        self.mode = ops.MODES[self.precision]
        # per-layer arithmetic (round 4, the precision sweep of scripts/precision_sweep.py): "prefix=mode,prefix=mode" in
        # VFN_PRECISION_MAP or a dict in model.precision_map; a layer takes the mode of the longest matching prefix of its name
        # ('encoder_q', 'encoder_m.res4', 'keyval', 'decoder.RF2', 'decoder.local', 'memread', 'bank_update' ...), else the model's
        pm = dict(getattr(model, 'precision_map', None) or {})
        for item in filter(None, os.environ.get('VFN_PRECISION_MAP', '').split(',')):
            k_, v_ = item.split('=')
            pm.setdefault(k_.strip(), v_.strip())
        for v_ in pm.values():
            if v_ not in ops.MODES:
                raise ValueError(f'precision map: unknown mode {v_!r}')
        self.pmap = sorted(((k_, ops.MODES[v_]) for k_, v_ in pm.items()), key=lambda kv: -len(kv[0]))
        self.mixed = bool(self.pmap)
        self.any_x3 = self.mode == 2 or any(m_ == 2 for _, m_ in self.pmap)
        self.fwd_count = 0           # segment samples / memorize calls run so far: vfloodnet_amd.autograd checks with them
        self.mem_count = 0           # whether the activations a backward pass needs are still the ones its forward wrote
        self.eager = False           # True: launch lists run launch by launch (instrumented frames: bench.py brackets launches with events)
        self._side = None            # side stream for the query side of the next frames
        self._side_busy = None       # event behind the last work enqueued on it
        from .refresh import Refresher
        self.refresher = Refresher(self.device)     # packed filters / folded constants follow the parameters in place (refresh())
        self._backward = None
        self._train_slot = 0
        self._batch = None           # (plan, query set) of Engine.query_batch: the sample whose frames are being segmented
        self._pack(model)
        self._settle()

    def _settle(self):
        """Run the refresh kernels once over what the constructors just derived with tensor operators: a new engine and a
        refreshed one then hold the same bits (the kernels round the folded scale gamma / sqrt(var + eps) correctly; the
        device's tensor-operator division is 1-2 ulp off that).  Skipped when a parameter is not a contiguous f32 device tensor."""
        try:
            self.refresher.run()
        except RuntimeError:
            pass

    def refresh(self, check=True):
        """The parameters changed in place (an optimizer step): rewrite everything derived from them -- the engine's packed
        filters and folded BatchNorm constants, the backward pass's data-gradient filters -- where it lies (two launches,
        csrc/refresh.hip); plans, buffers and descriptors stay.  Raises RuntimeError if a parameter is no longer a contiguous f32
        device tensor (the caller builds a new engine then)."""
        self.refresher.run(verify=check)
        for layer in self._layers():                       # (reduced-precision operand images; Winograd banks that are not registered)
            if layer._w_lp:
                layer.refresh_derived()

    def _layers(self):
        def walk(o):
            if isinstance(o, (ConvLayer, Pred2Layer)):
                yield o
            elif isinstance(o, dict):
                for v in o.values():
                    yield from walk(v)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    yield from walk(v)
        yield from walk([self.enc_q, self.enc_m, self.keyval, self.dec])

    def backward(self):
        """The (cached) backward pass of this engine, emptied of the previous step's gradients."""
        from .backward import ModelBackward
        if self._backward is None:
            self._backward = ModelBackward(self)
            self._settle()
        self._backward.reset()
        return self._backward

    def use_winograd(self, layer, M, bf=0):
        """Winograd F(4x4, 3x3) for this layer (M = N * H * W output pixels)?  See _WINOGRAD above.  ``bf`` = 1: the plain-bf16 mode's
        table where it has the shape (else the f32 one's answer)."""
        if _WINOGRAD == '0' or getattr(layer, 'k', 0) != 3 or layer.stride != 1 or layer.pad != 1:
            return False
        if layer.cin % 32 or layer.cout % 4 or layer.cout < 32:
            return False
        if _WINOGRAD == '2':
            return True
        hit = (_WINO_TABLE_BF16.get((M, layer.cin, layer.cout)) if bf == 1 else None)
        if hit is None:
            hit = _WINO_TABLE.get((M, layer.cin, layer.cout))
        if hit is not None:
            return bool(hit)
        return layer.cin >= 128 and layer.cout >= 128 and M >= _WINOGRAD_MIN_M

    def layer_mode(self, name):
        for prefix, m_ in self.pmap:
            if name.startswith(prefix):
                return m_
        return self.mode

    def layer_precision(self, name):
        return {v: k for k, v in ops.MODES.items()}[self.layer_mode(name)]

    # ------------------------------------------------------------------ weights
    def _pack_trunk(self, enc):
        dev = self.device
        out = {}
        for lname in ('res2', 'res3', 'res4'):
            blocks = []
            for blk in getattr(enc, lname):
                reg = self.refresher
                b = dict(conv1=ConvLayer(blk.conv1, blk.bn1, dev, reg=reg), conv2=ConvLayer(blk.conv2, blk.bn2, dev, reg=reg),
                         conv3=ConvLayer(blk.conv3, blk.bn3, dev, reg=reg))
                if hasattr(blk, 'downsample'):
                    b['down'] = ConvLayer(blk.downsample[0], blk.downsample[1], dev, reg=reg)
                blocks.append(b)
            out[lname] = blocks
        return out

    def _pack(self, m):
        dev = self.device
        with torch.no_grad():
            self.mean = [float(x) for x in m.encoder_q.mean.flatten().cpu()]
            self.std = [float(x) for x in m.encoder_q.std.flatten().cpu()]
            mm = [float(x) for x in m.encoder_m.mean.flatten().cpu()]
            ms = [float(x) for x in m.encoder_m.std.flatten().cpu()]
            if mm != self.mean or ms != self.std:
                raise RuntimeError('encoder_m / encoder_q normalisation buffers differ; unsupported checkpoint')
            from .refresh import FORWARD, STEM
            reg = self.refresher
            self.stem_q_w = ops.pack_stem_weight([m.encoder_q.conv1.weight]).to(dev)
            self.stem_q_scale, self.stem_q_shift = [t.to(dev).clone() for t in W.bn_scale_shift(m.encoder_q.bn1)]
            stem_m = [m.encoder_m.conv1, m.encoder_m.conv1_m, m.encoder_m.conv1_o]
            self.stem_m_w = ops.pack_stem_weight([c.weight for c in stem_m]).to(dev)
            self.stem_m_scale, self.stem_m_shift = [t.to(dev).clone() for t in W.bn_scale_shift(m.encoder_m.bn1)]
            reg.add_filter(m.encoder_q.conv1.weight, self.stem_q_w, STEM)
            reg.add_epilogue(self.stem_q_scale, self.stem_q_shift, bn=m.encoder_q.bn1, eps=W.BN_EPS)
            plane = 0
            for c in stem_m:
                reg.add_filter(c.weight, self.stem_m_w, STEM, dst_row0=plane)
                plane += c.weight.shape[1]
            reg.add_epilogue(self.stem_m_scale, self.stem_m_shift, bn=m.encoder_m.bn1, eps=W.BN_EPS)
            self.enc_q = self._pack_trunk(m.encoder_q)
            self.enc_m = self._pack_trunk(m.encoder_m)
            # KeyValue: Key and Value share the input -> one GEMM with 640 filters (AFB_URR.py:106,109)
            kv = m.keyval_r4
            wk = torch.cat([kv.Key.weight.detach().float(), kv.Value.weight.detach().float()], 0)
            bk = torch.cat([kv.Key.bias.detach().float(), kv.Value.bias.detach().float()], 0)
            holder = type('KV', (), {})()
            holder.weight, holder.bias, holder.stride, holder.padding = wk, bk, (1, 1), (1, 1)
            self.keyval = ConvLayer(holder, None, dev)
            self.keyval.shift = self.keyval.shift.clone()
            row = 0
            self.keyval._wino_src = []
            for c in (kv.Key, kv.Value):                                       # the two halves of the 640-filter convolution
                reg.add_filter(c.weight, self.keyval.w, FORWARD, dst_row0=row)
                reg.add_epilogue(None, self.keyval.shift[row:row + c.weight.shape[0]], bias=c.bias)
                self.keyval._wino_src.append((reg, c.weight, 0, row))         # (its Winograd banks, if a plan uses them: w_wino)
                row += c.weight.shape[0]
            d = m.decoder
            cl = lambda c: ConvLayer(c, None, dev, reg=reg)
            rb = lambda r: dict(conv1=cl(r.conv1), conv2=cl(r.conv2))
            rf = lambda r: dict(convFS=cl(r.convFS), ResFS=rb(r.ResFS), ResMM=rb(r.ResMM))
            self.dec = dict(convFM_m=ConvLayer(d.convFM, None, dev, (0, DV), with_bias=False, reg=reg),
                            convFM_q=ConvLayer(d.convFM, None, dev, (DV, 2 * DV), reg=reg),
                            local_convFM_r1=ConvLayer(d.local_convFM, None, dev, (0, 64), reg=reg),
                            local_convFM_loc=ConvLayer(d.local_convFM, None, dev, (64, 128), with_bias=False, reg=reg),
                            ResMM=rb(d.ResMM), RF3=rf(d.RF3), RF2=rf(d.RF2), pred2=Pred2Layer(d.pred2, dev, reg=reg),
                            local_ResMM=rb(d.local_ResMM),
                            local_pred2=Pred2Layer(d.local_pred2, dev, reg=reg))

    # ------------------------------------------------------------------ plans
    def plan(self, H0, W0, obj_n, keep_acts=False, slot=0):
        """``keep_acts`` (training): a plan whose bottlenecks keep their activations for the backward pass; ``slot``: training
        alternates between two such plans, so that the weight gradients of one sample (the backward pass's side stream) can
        still read its activations while the next sample's forward runs."""
        key = (H0, W0, obj_n, bool(keep_acts)) + ((slot,) if slot else ())
        p = self.plans.get(key)
        if p is None:
            if len(self.plans) >= 6:
                self.plans.pop(next(iter(self.plans)))
            p = FramePlan(self, H0, W0, obj_n, keep_acts=bool(keep_acts))
            self.plans[key] = p
            if self.refresher._tables is None:             # (the plan registered new derived tensors: Winograd filter banks)
                self._settle()
        return p

    @staticmethod
    def _check_frame(frame, batch_ok=False):
        if frame.dim() != 4 or frame.shape[1] != 3 or frame.shape[0] < 1 or (frame.shape[0] != 1 and not batch_ok):
            raise RuntimeError(f'expected RGB frames [{"bs" if batch_ok else "1"},3,h,w], got {tuple(frame.shape)}')
        _lib.require_gpu(frame, 'frame')

    # ------------------------------------------------------------------ API
    def _join_backward(self, plan=None):
        """A backward pass may still be reading activations on its side stream (backward.ModelBackward): wait for all of it, or
        for the part that reads ``plan``'s buffers."""
        if self._backward is not None:
            if plan is None:
                self._backward.join()
            else:
                self._backward.wait_plan(plan)

    def memorize(self, frame, mask, training=False):
        self._check_frame(frame)
        self._join_backward()
        _, K, H, Wd = mask.shape
        p = self.plan(frame.shape[2], frame.shape[3], K, keep_acts=training)
        self.last_memorize = p
        self.mem_count += 1
        p.frame_in.copy_(frame[0])
        p.mask_in.copy_(mask[0])                       # uint8 / float -> float32 (mask.float(), AFB_URR.py:262)
        p.graphs.run(p.mem, eager=training or self.eager)
        kv = p.kv_m
        k_list = [kv[i, :, :DK].t() for i in range(K)]            # [128, HW] views
        v_list = [kv[i, :, DK:].t() for i in range(K)]            # [512, HW]
        return k_list, v_list

    def segment(self, frame, fb, update_bank, training=False):
        """AFB_URR.segment for frame f32[bs,3,h,w].  bs = 1 is the inference loop (test_video_seg.py:108); bs > 1 is how
        the training script calls it (train_video_seg.py:69): the samples of a batch are independent given the bank, so
        they run one after the other through the same launch list, and -- as in the reference, AFB_URR.py:165 -- only
        sample 0 contributes hit counts to ``fb.info``.  ``training``: no padding (AFB_URR.py:278: the frame size must
        then be a multiple of 16, anything else fails in the reference's decoder as well)."""
        self._check_frame(frame, batch_ok=True)
        bs = frame.shape[0]
        K = fb.obj_n
        H, Wd = frame.shape[2], frame.shape[3]
        if training and (H % 16 or Wd % 16):
            raise RuntimeError(f'training-mode segment does not pad (AFB_URR.py:278): {H}x{Wd} is not a multiple of 16')
        slot = 0
        batch = self._batched_slot(frame) if training and bs == 1 else None
        if batch is not None:                               # (query_batch ran the frame-only part for the whole sample)
            p = batch[0]
        else:
            if training and _TRAIN_SLOTS > 1:
                slot, self._train_slot = self._train_slot, (self._train_slot + 1) % _TRAIN_SLOTS
            p = self.plan(H, Wd, K, keep_acts=training, slot=slot)
        self._join_backward(p)
        if fb._kbuf is None:
            raise RuntimeError('feature bank is empty: call fb.init_bank() first')
        if fb._hw != p.HW:
            raise RuntimeError('feature bank was built for a different frame size')
        out = p.score if bs == 1 else torch.empty(bs, K, H, Wd, device=self.device, dtype=torch.float32)
        for b in range(bs):
            fr = frame[b:b + 1]
            if batch is not None:
                qs, slot = batch[1], batch[2]
            else:
                qs, slot = self._take_prefetched(p, fr, frame._version) if b == 0 else (None, 0)
            if qs is None:
                # not prefetched (first frame of a clip, a skipped frame, a direct segment() call): the frame-only part runs
                # here, on this stream, in a set the side stream is not filling
                qs = self._idle_set(p)
                if self._side_busy is not None:             # (that set's last prefetch, if any, has finished with it)
                    torch.cuda.current_stream().wait_event(self._side_busy)
                qs.frames[0].copy_(fr[0])
                qs.keys, qs.held, qs.consumed, qs.stage, qs.n = [None, None], [None, None], [True, True], 0, 0
                p.graphs.run(qs.pre[1], eager=training or self.eager)
                slot = 0
            self._memory_read(p, fb, update_bank and b == 0, qs.kv_q[slot:slot + 1])
            p.graphs.run(qs.post[slot], eager=training or self.eager)
            self.last_query = (p, qs, slot)                 # (where the backward slice finds this frame's activations)
            self.fwd_count += 1
            if bs > 1:
                out[b].copy_(p.score[0])
        return out

    # ------------------------------------------------------------------ training: the query side of a whole sample in one pass
    def query_batch(self, frames, obj_n):
        """The frame-only part of ``segment`` -- query encoder, KeyValue, the decoder branches that depend on the frame alone -- for
        ALL n frames of a training sample at once (frames f32[n,3,H,W] on the GPU): every layer sees n times the pixels instead of n
        launches of a 1/16-resolution layer with 625 of them.  The ``segment(frames[i:i+1], fb, training=True)`` calls that follow
        find their slot; the activations of the whole batch stay for the backward pass (ModelBackward.finish_query)."""
        n, H, Wd = frames.shape[0], frames.shape[2], frames.shape[3]
        if H % 16 or Wd % 16:
            raise RuntimeError(f'training-mode segment does not pad (AFB_URR.py:278): {H}x{Wd} is not a multiple of 16')
        self._join_backward()
        p = self.plan(H, Wd, obj_n, keep_acts=True)
        qs = p.batch_set(n)
        if self.refresher._tables is None:                 # (the batch lists registered new derived tensors: Winograd banks)
            self._settle()
        qs.frames[:n].copy_(frames)
        for l in qs.pre[n]:
            l()
        qs.keys = [self._key(frames[i:i + 1]) for i in range(n)]
        qs.held, qs.consumed, qs.n, qs.stage = [frames] * n, [False] * n, n, 2
        self._batch = (p, qs)
        return qs

    def segment_batch(self, fb):
        """The bank-dependent part of ``segment`` for all frames ``query_batch`` holds (training): one memory read per frame into the
        frame's images of the batch, then the decoder once over frames x objects (DecoderBatch).  Returns the logits f32
        [n, obj_n, H, W]; the activations stay for ``ModelBackward.segment_batch``."""
        if self._batch is None:
            raise RuntimeError('segment_batch follows query_batch')
        p, qs = self._batch
        if fb.obj_n != p.obj_n or fb._kbuf is None or fb._hw != p.HW:
            raise RuntimeError('feature bank does not match the batch (objects / frame size), or is empty')
        self._join_backward()
        b = qs.dec_batch()
        if self.refresher._tables is None:
            self._settle()
        K, G = p.obj_n, qs.n
        if _BATCH_MEMREAD:
            self._memory_read(b.mr, fb, False, qs.kv_q[0:G])
            b.dec_in.view(G, K, p.HW, DV).copy_(b.mr.dec_in.view(K, G, p.HW, DV).permute(1, 0, 2, 3))
        else:
            for g in range(G):
                self._memory_read(p, fb, False, qs.kv_q[g:g + 1], out=b.grp(b.dec_in, g))
        for l in b.post:
            l()
        qs.consumed = [True] * qs.n
        self.fwd_count += qs.n
        self.last_batch = (b, qs)
        return b.score

    # ------------------------------------------------------------------ inference: the frames between two memorize calls in one pass
    @torch.no_grad()
    def segment_group(self, frames, fb, update_bank, prefetch=None):
        """``segment`` for G consecutive frames f32[G,3,h,w] that see the SAME bank -- the frames between two ``memorize`` /
        ``FeatureBank.update`` calls when only every n-th frame is memorised (test_video_seg.py:110-112 with a key-frame interval;
        BASELINE config C3: every 5th) -- in one pass: the frame-only side over the G frames at once, the G memory reads back to back
        (each bumps the hit accumulators as its own ``segment`` call would), the decoder once over G x obj_n images
        (QuerySet(nq=G) / DecoderBatch, the structures the training step batches a sample with).  Given the bank the frames are
        independent, so the logits are those of G ``segment`` calls up to the summation order inside the convolutions (larger
        GEMMs pick other tiles) and of the bank slices.  Returns logits f32[G,obj_n,h,w] (a buffer of the plan: valid until the next
        group of this size); eval-mode padding as ``segment``.  ``frames``: a tensor, or a list of G f32[1,3,h,w] tensors;
        ``prefetch``: the NEXT group as such a list -- its frame-only side then runs on the side stream behind this group's decoder
        (``prefetch_group``) and the next call, given the same tensors, finds it done."""
        if isinstance(frames, (list, tuple)):             # G tensors f32[1,3,h,w]: copied straight into the batch's frame slots
            for fr in frames:
                self._check_frame(fr)
                if fr.shape != frames[0].shape:
                    raise RuntimeError(f'segment_group: frames of one size, got {tuple(fr.shape)} and {tuple(frames[0].shape)}')
            G, H, Wd = len(frames), frames[0].shape[2], frames[0].shape[3]
        else:
            self._check_frame(frames, batch_ok=True)
            G, H, Wd = frames.shape[0], frames.shape[2], frames.shape[3]
        K = fb.obj_n
        p = self.plan(H, Wd, K)
        self._join_backward(p)
        if fb._kbuf is None:
            raise RuntimeError('feature bank is empty: call fb.init_bank() first')
        if fb._hw != p.HW:
            raise RuntimeError('feature bank was built for a different frame size')
        if G > MAX_GROUP:
            raise RuntimeError(f'segment_group: at most {MAX_GROUP} frames per group, got {G}')
        keys = [self._key(fr) for fr in frames] if isinstance(frames, (list, tuple)) else None
        qs = self._group_qset(p, G)
        hit = keys is not None and qs.stage == 2 and qs.keys == keys and not any(qs.consumed)
        b = qs.dec_batch()
        if self.refresher._tables is None:                 # (the batch lists registered new derived tensors: Winograd banks)
            self._settle()
        if hit:
            torch.cuda.current_stream().wait_event(qs.done)      # prefetch_group ran the frame-only side on the side stream
        else:
            if self._side_busy is not None:                 # whatever the side stream is doing shares the lists' workspace
                torch.cuda.current_stream().wait_event(self._side_busy)
            if isinstance(frames, (list, tuple)):
                for i, fr in enumerate(frames):
                    qs.frames[i].copy_(fr[0])
            else:
                qs.frames[:G].copy_(frames)
            p.graphs.run(qs.pre[qs.nq], eager=self.eager)
        qs.consumed, qs.held, qs.stage = [True] * G, [None] * G, 0
        # one memory read per frame: the hit accumulator is bumped by log(hits + 1) PER FRAME (AFB_URR.py:165-174), which a read over
        # all G x HW query columns cannot reproduce from its summed counts (and it would save nothing: 5 x 113 against 554 us at C3)
        for g in range(G):
            self._memory_read(p, fb, update_bank, qs.kv_q[g:g + 1], out=b.grp(b.dec_in, g))
        p.graphs.run(b.post, eager=self.eager)
        if prefetch:
            # the next group's frame-only side: on the side stream, BEHIND this group's decoder (a second set of buffers and an earlier
            # start -- under the memory read or the decoder -- measured 2-3 % slower: profiles/r06_group_variants.txt)
            self.prefetch_group(prefetch, K)
        self.fwd_count += G
        return b.score[:G]

    @staticmethod
    def _group_qset(p, G):
        """The batch set a group of G frames runs through: its own, or -- for the short group at the end of a clip -- the smallest
        set already built that holds at least G frames (the surplus slots keep the frames of an earlier group: their images run
        through the lists and are not looked at; no hit counts come from them, the memory reads are per frame).  Building a set
        costs tens of milliseconds of allocations and descriptors: not for one group."""
        if G not in p._qbatch:
            built = [n for n, q_ in p._qbatch.items() if n > G and q_._dec_batch is not None]
            if built:
                return p._qbatch[min(built)]
        return p.batch_set(G)

    def prefetch_group(self, frames, obj_n):
        """The frame-only side of the NEXT group (a list of G f32[1,3,h,w] tensors on the GPU) on the side stream, behind everything
        enqueued so far -- call it after the current group's ``segment_group``: it then runs underneath that group's memorize /
        update / label tail, and the next ``segment_group`` with these very tensors picks it up.  The set of a group size is
        single-buffered: the decoder of the current group has finished with it when the side stream starts."""
        for fr in frames:
            self._check_frame(fr)
        G = len(frames)
        p = self.plan(frames[0].shape[2], frames[0].shape[3], obj_n)
        qs = self._group_qset(p, G)
        qs.dec_batch()
        if self.refresher._tables is None:
            self._settle()
        self.side_stream()
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            for i, fr in enumerate(frames):
                qs.frames[i].copy_(fr[0])
            p.graphs.run(qs.pre[qs.nq], eager=self.eager)
            qs.done = torch.cuda.Event()
            qs.done.record()
        self._side_busy = qs.done
        qs.keys = [self._key(fr) for fr in frames]
        qs.held = list(frames)
        qs.consumed = [False] * G
        qs.n, qs.stage = G, 2

    def _batched_slot(self, frame):
        if self._batch is None:
            return None
        p, qs = self._batch
        key = self._key(frame)
        for slot in range(qs.n):
            if qs.keys[slot] == key and not qs.consumed[slot]:
                qs.consumed[slot] = True
                return p, qs, slot
        return None

    # ------------------------------------------------------------------ look-ahead of the query side
    @staticmethod
    def _key(frame, version=None):
        return (frame.data_ptr(), frame._version if version is None else version, tuple(frame.shape))

    def _take_prefetched(self, p, fr, version):
        """(set, slot) holding the frame-only results for ``fr`` -- finishing a prefetch that is still at its first half
        and making this stream wait for it -- or (None, 0)."""
        key = self._key(fr, version)
        for qs in p.qsets:
            for slot in range(qs.n):
                if qs.stage > 0 and qs.keys[slot] == key and not qs.consumed[slot]:
                    if qs.stage == 1:
                        self.prefetch_finish(p)
                    torch.cuda.current_stream().wait_event(qs.done)
                    qs.consumed[slot] = True
                    qs.held[slot] = None
                    return qs, slot
        return None, 0

    def _idle_set(self, p):
        """A set that holds no unconsumed, prefetched frame (else: the one with fewer of them -- a wrong hint is redone)."""
        def live(qs):
            return 0 if qs.stage == 0 else sum(1 for i in range(qs.n) if not qs.consumed[i])
        return min(p.qsets, key=live)

    def prefetched_keys(self, p):
        """Frames whose query side is prefetched (complete or begun) and not yet consumed."""
        return [qs.keys[i] for qs in p.qsets if qs.stage > 0 for i in range(qs.n) if not qs.consumed[i]]

    def prefetch_pending(self, p):
        return any(qs.stage == 1 for qs in p.qsets)

    def prefetch_begin(self, frames, obj_n, full=False):
        """Start the frame-only part of ``segment`` for the next ONE or TWO frames (a list of f32[1,3,h,w] tensors on
        the GPU) on the side stream: query encoder, KeyValue and the decoder's skip-feature branches, batched over the
        frames.  It depends only on those frames, so it may overlap ``memorize`` / ``FeatureBank.update`` of the
        current frame; the ``segment`` calls for them pick the results up.  Call it after the current frame's
        ``segment`` has been enqueued: the side stream starts behind everything enqueued so far, i.e. it never competes
        with a memory read or a decoder that is already on its way (measured slower: the critical path then shares the
        CUs).  ``full=False`` enqueues the first half of the launch list only; ``prefetch_finish`` (called after the
        NEXT frame's ``segment``) enqueues the rest -- so one pass over two frames is spread underneath the memorize /
        update phases of two frames.  Returns False if the set could not be claimed."""
        for fr in frames:
            self._check_frame(fr)
        n = len(frames)
        assert n in (1, 2)
        p = self.plan(frames[0].shape[2], frames[0].shape[3], obj_n)
        if self.prefetch_pending(p):
            self.prefetch_finish(p)
        qs = self._idle_set(p)
        self.side_stream()
        ready = torch.cuda.Event()
        ready.record()                                           # everything enqueued so far (the decoder of the current frame)
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            for i, fr in enumerate(frames):
                qs.frames[i].copy_(fr[0])
            lst = qs.pre[n]
            cut = len(lst) if full else qs.split[n]
            p.graphs.run(lst, 0, cut, eager=self.eager)
            qs.done = torch.cuda.Event()
            qs.done.record()
        self._side_busy = qs.done
        qs.keys = [self._key(fr) for fr in frames] + [None] * (2 - n)
        qs.held = list(frames) + [None] * (2 - n)
        qs.consumed = [False] * n + [True] * (2 - n)
        qs.n = n
        qs.stage = 2 if full else 1
        return True

    def prefetch_finish(self, p=None):
        """Enqueue the second half of the prefetch begun by ``prefetch_begin(full=False)`` (behind everything enqueued so
        far on the current stream)."""
        plans = [p] if p is not None else list(self.plans.values())
        for pl in plans:
            for qs in pl.qsets:
                if qs.stage != 1:
                    continue
                ready = torch.cuda.Event()
                ready.record()
                with torch.cuda.stream(self._side):
                    self._side.wait_event(ready)
                    pl.graphs.run(qs.pre[qs.n], qs.split[qs.n], None, eager=self.eager)
                    qs.done = torch.cuda.Event()
                    qs.done.record()
                self._side_busy = qs.done
                qs.stage = 2

    def side_stream(self):
        """The stream the query side of the coming frames runs on, on a hardware queue of its own (the overlap is the point:
        _lib.independent_stream); created on first use."""
        if self._side is None:
            self._side = _lib.independent_stream(self.device)
        return self._side

    def prefetch_query(self, frame, obj_n):
        """One frame of look-ahead (round-1/2 API): ``prefetch_begin([frame], full=True)``."""
        return self.prefetch_begin([frame], obj_n, full=True)

    def _memory_read(self, p, fb, update_bank, kv_q=None, out=None):
        """Matcher.forward (AFB_URR.py:136-178) on the bank slabs; ``out`` [obj_n, h16, w16, 512] (default: the plan's ``dec_in``)."""
        L = _lib.lib()
        s = stream()
        kv_q = p.kv_q if kv_q is None else kv_q
        out = p.dec_in if out is None else out
        K, HW, cap = fb.obj_n, p.HW, fb._cap
        nsplit_scan = pick_scan_slices(HW, K, fb.len_upper())
        scale = 1.0 / math.sqrt(DK)
        d = BankScanDesc()
        d.q, d.bank_k, d.bank_len, d.rowscale, d.part = ptr(kv_q), ptr(fb._kbuf), ptr(fb._len_dev), None, ptr(p.ml_part)
        d.stride_q, d.stride_k, d.stride_rs = 0, cap * DK, 0
        d.scale = scale
        d.ldq, d.q_per_obj, d.HW, d.obj_n, d.nsplit, d.mode = DK + DV, 0, HW, K, nsplit_scan, 0
        mr_mode = self.layer_mode('memread') if hasattr(self, 'layer_mode') else self.mode     # (tests drive this with a stand-in engine)
        d.precision = mr_mode
        d.work_counter = ptr(p.work)
        klp, vlp = fb.lp_image() if mr_mode else (None, None)      # the bank's kept split-bf16 image (reduced precision)
        d.bank_k_lp = ptr(klp) if klp is not None else None
        # f32: the scan stores the scores it forms and the apply kernel reads them back instead of repeating the
        # keys x queries GEMM (bit-identical; HW x capacity floats per object, VFN_STORE_SCORES=0 or more than
        # VFN_SCORES_MAX_GB switch it off)
        scores = _scores_buffer(p, fb) if mr_mode == 0 else None
        if scores is not None:
            d.scores, d.stride_scores = ptr(scores), scores.shape[1]
        check(L.vfn_bank_scan(_lib.C.byref(d), s), 'vfn_bank_scan')
        check(L.vfn_bank_scan_finish(ptr(p.ml_part), nsplit_scan, HW, K, 0, ptr(p.ml), None, None, None, s),
              'vfn_bank_scan_finish')
        m = MemReadDesc()
        m.q = ptr(kv_q)
        m.qv = None                       # the query value joins through decoder.convFM.q instead of a concat
        m.bank_k, m.bank_v, m.bank_len, m.ml, m.o_part = ptr(fb._kbuf), ptr(fb._vbuf), ptr(fb._len_dev), ptr(p.ml), ptr(p.o_part)
        m.cnt = ptr(fb._cnt) if update_bank else None
        m.info, m.out = ptr(fb._ibuf), ptr(out)
        m.stride_k, m.stride_v, m.stride_cnt, m.stride_info = cap * DK, cap * DV, cap, cap * 2
        m.scale, m.thres = scale, 1e-3
        m.ldq, m.ldqv, m.ld_out, m.HW, m.obj_n = DK + DV, DK + DV, out.shape[-1], HW, K
        m.precision = mr_mode
        # 128 query columns per workgroup (8 waves)
        m.nsplit = pick_nsplit(HW, K, fb.len_upper(), QT_SCAN, MAX_SPLIT)
        if klp is not None:
            m.bank_k_lp, m.bank_v_lp = ptr(klp), ptr(vlp)
        if scores is not None:
            m.scores, m.stride_scores = ptr(scores), scores.shape[1]
        check(L.vfn_memread_apply(_lib.C.byref(m), s), 'vfn_memread_apply')
        check(L.vfn_memread_finish(_lib.C.byref(m), s), 'vfn_memread_finish')

    def capture(self, H0, W0, obj_n, group=0):
        """Capture the HIP graphs of this frame size's launch lists NOW -- at plan warm-up, before the frame loop -- instead of on
        each list's third run inside the loop (torch.cuda.graph() enters with a device-wide synchronize, a gc.collect and
        an empty_cache; inside the loop those one-off stalls land in a timed region and drain the side stream's prefetch).  The
        lists only write the plan's own activation buffers, which every real frame overwrites before reading; call it while no
        prefetch is outstanding (ClipRunner.start does, after the first frame's memorize)."""
        p = self.plan(H0, W0, obj_n)
        if not _GRAPHS or self.eager or p.keep_acts:
            return 0
        before = p.graphs.captures
        torch.cuda.current_stream().synchronize()
        p.graphs.warm(p.mem)
        for qs in p.qsets:
            for n in qs.sizes:
                lst = qs.pre[n]
                p.graphs.warm(lst)                                   # the whole list (prefetch_begin(full=True), segment())
                if 0 < qs.split[n] < len(lst):                       # ... and its two halves (prefetch_begin / prefetch_finish)
                    p.graphs.warm(lst, 0, qs.split[n])
                    p.graphs.warm(lst, qs.split[n], None)
            for L in qs.post:
                p.graphs.warm(L)
        for G in ([group] if group and group > 1 else []):           # segment_group's lists for groups of this size
            qs = p.batch_set(G)
            b = qs.dec_batch()
            if self.refresher._tables is None:
                self._settle()
            p.graphs.warm(qs.pre[G])
            p.graphs.warm(b.post)
        torch.cuda.current_stream().synchronize()
        return p.graphs.captures - before

    # ------------------------------------------------------------------ tuning
    def autotune(self, H0, W0, obj_n, iters=3, only_missing=False, shape_filter=None, cfg_filter=None):
        """Time every tile config for every distinct conv shape of this frame size; keep the fastest.
        ``only_missing``: leave shapes that the measured table already covers alone (a frame size the shipped
        tables were not tuned for costs a few seconds once, e.g. at the start of ``video_seg.main``)."""
        p = self.plan(H0, W0, obj_n)
        seen = {}
        # lists that run on the side stream keep the side stream's split-K workspace: the look-ahead sets' and -- in an inference plan,
        # where Engine.prefetch_group runs them beside memorize / update -- the batch sets' frame-only lists
        side_sets = list(p.qsets) + ([] if p.keep_acts else list(p._qbatch.values()))
        side_lists = [id(qs.pre[n]) for qs in side_sets for n in qs.sizes]
        for lst in p.all_lists():
            for l in lst:
                l_side = id(lst) in side_lists
                if l.fn is ops.conv2d_launch:
                    d = l.args[0]
                    key = (d.M, d.Cout, d.KH * d.KW * d.Cin, int(l.args[2]))
                    if only_missing and key[:3] in _TABLES[key[3]]:
                        continue
                    if shape_filter is not None and not shape_filter(key[:3]):      # (re-tune a subset, e.g. after a new tile shape)
                        continue
                    seen.setdefault(key, []).append((l, l_side))
        for key, launches in seen.items():
            d = launches[0][0].args[0]
            bf = key[3]
            best = tune_desc(d, bf, p.ws, p.cnt, iters=iters, cfg_filter=cfg_filter)
            _TABLES[bf][key[:3]] = best
            for l, in_q in launches:
                l.args = (l.args[0], apply_choice(l.args[0], best, p.ws_q if in_q else p.ws, p.cnt_q if in_q else p.cnt), bf)
        if seen:
            p.graphs.invalidate()                           # (captured lists hold the old tile choices)
        return dict(_TABLES[self.mode])
