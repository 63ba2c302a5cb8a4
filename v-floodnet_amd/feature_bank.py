"""``FeatureBank``: the reference's adaptive feature bank API on device-resident storage.

Drop-in for ``video_module.model.FeatureBank`` (``FeatureBank.py:8-149``) as used by
``test_video_seg.py:83,101,112,123`` and read by the matcher (``AFB_URR.py:140-174``):

    fb = FeatureBank(obj_n, memory_budget, device, update_rate=0.1, thres_close=0.95)
    fb.init_bank(k4_list, v4_list)            # lists of [128,HW] / [512,HW]
    fb.update(k4_list, v4_list, frame_idx)    # merge (scatter_mean) / append / LFU evict
    fb.keys[i] [128,B]  fb.values[i] [512,B]  fb.info[i] [B,2]  fb.peak_n  fb.replace_n  fb.class_budget

Storage is MI355X-first rather than the reference's per-frame ``torch.cat``: one
pre-allocated entry-major slab per bank (``[obj][cap][128]``, ``[obj][cap][512]``,
``[obj][cap][2]``) with the live length kept in *device* memory, so a whole
``update`` is a fixed sequence of kernel launches with no host round trip; the public
``keys / values / info`` attributes are transposed views of the slabs (reading them
is the only thing that synchronises).  ``cap = class_budget + HW`` entries: at the
default budget that is 2 x 260 MB -- sized for 288 GB of HBM, not for an 11 GB card.
"""
import numpy as np
import torch

from . import _lib, ops
from ._lib import ptr, stream, check, BankDesc, BankScanDesc

DK, DV = 128, 512
MAX_SPLIT = 20            # memory-read apply slices (o_part slabs)
MAX_SPLIT_SCAN = 256      # bank slices (work items per query tile and object) of the persistent scan kernels
QT, QT_SCAN, CH = 64, 128, 64
MAX_HW = 32768           # include/vfn_hip.h VFN_BANK_MAX_HW


def pick_scan_slices(hw, obj_n, b_upper, target_items=1536, min_chunks=8):
    """Bank slices per (query tile, object) for the persistent scan kernels: about ``target_items`` work items in all
    (three per resident workgroup slot, so the queue drains evenly), but at least ``min_chunks`` 64-entry chunks each
    (a slice reloads its query fragments)."""
    nchunks = max(1, (b_upper + CH - 1) // CH)
    qtiles = (hw + QT_SCAN - 1) // QT_SCAN
    import os
    if os.environ.get('VFN_NSPLIT_SCAN'):
        return max(1, min(int(os.environ['VFN_NSPLIT_SCAN']), nchunks, MAX_SPLIT_SCAN))
    want = max(1, target_items // (qtiles * obj_n))
    return max(1, min(want, nchunks // min_chunks if nchunks >= min_chunks else 1, MAX_SPLIT_SCAN))


def pick_nsplit(hw, obj_n, b_upper, qt=QT, max_split=MAX_SPLIT):
    """Bank slices per query tile: fill the 256 CUs x 2 resident workgroups in whole rounds."""
    nchunks = max(1, (b_upper + CH - 1) // CH)
    qtiles = (hw + qt - 1) // qt
    import os
    env = 'VFN_NSPLIT' if qt == QT else 'VFN_NSPLIT_SCAN'
    if os.environ.get(env):
        return min(int(os.environ[env]), nchunks, max_split)
    best, best_eff = 1, -1.0
    for s in range(1, max(1, min(nchunks // 3, max_split)) + 1):      # >= 3 chunks per slice
        blocks = qtiles * obj_n * s
        eff = blocks / (((blocks + 255) // 256) * 256)
        if eff >= best_eff:
            best, best_eff = s, eff
    return best


class FeatureBank:
    def __init__(self, obj_n, memory_budget, device, update_rate=0.1, thres_close=0.95, precision=None):
        self.obj_n = obj_n
        self.update_rate = update_rate
        self.thres_close = thres_close
        self.device = torch.device(device)
        # arithmetic of the cosine match in update() ('fp32' | 'bf16x3' | 'bf16', as AFB_URR(precision=...))
        import os
        self.precision = precision or os.environ.get('VFN_PRECISION', 'fp32')
        if self.precision not in ops.MODES:
            raise ValueError(f'precision must be one of {sorted(ops.MODES)}, got {self.precision!r}')

        self._graph_kv = None
        self.peak_n = np.zeros(obj_n)
        self.replace_n = np.zeros(obj_n)

        self.class_budget = memory_budget // obj_n          # FeatureBank.py:20-22
        if obj_n == 2:
            self.class_budget = 0.8 * self.class_budget

        self._cap = 0
        self._hw = 0
        self._len_host = None        # exact lengths after the last sync
        self._len_upper = None       # upper bound valid without a sync
        self._n_updates = 0
        self._hist = {}
        self._dirty = False
        self._kbuf = self._vbuf = self._ibuf = None
        self._scratch = None
        self._norms_valid = self._lp_valid = False    # bank norms on the device describe the bank as it is (carried across updates)
        self._klp = self._vlp = None  # split-bf16 image of keys / values for the reduced-precision kernels (lp_image)

    # ------------------------------------------------------------------ storage
    def _require_gpu(self):
        if self.device.type != 'cuda':
            raise RuntimeError('FeatureBank lives in GPU memory (HIP kernels only, no CPU fallback); '
                               f'got device {self.device}')
        _lib.lib()

    def _alloc(self, hw, n_init):
        self._require_gpu()
        dev = self.device
        o = self.obj_n
        cap = (int(max(self.class_budget, n_init)) + 2 * hw + 2 * CH + 63) // 64 * 64
        self._cap, self._hw = cap, hw
        self._kbuf = torch.empty(o, cap, DK, device=dev)
        self._vbuf = torch.empty(o, cap, DV, device=dev)
        self._ibuf = torch.zeros(o, cap, 2, device=dev)
        self._len_dev = torch.zeros(o, dtype=torch.int32, device=dev)
        self._stats = torch.zeros(o, 4, dtype=torch.int32, device=dev)
        self._knorm = torch.empty(o, cap, device=dev)
        self._vnorm = torch.empty(o, cap, device=dev)
        self._kinv = torch.empty(o, cap, device=dev)
        self._cnt = torch.zeros(o, cap, dtype=torch.int32, device=dev)
        self._keep_dst = torch.empty(o, cap, dtype=torch.int32, device=dev)
        self._plan = torch.zeros(o, 4, dtype=torch.int32, device=dev)
        self._new = torch.empty(o, hw, DK + DV, device=dev)       # staging for foreign-layout inputs
        self._nknorm = torch.empty(o, hw, device=dev)
        self._nvnorm = torch.empty(o, hw, device=dev)
        self._nkinv = torch.empty(o, hw, device=dev)
        self._midx = torch.empty(o, hw, dtype=torch.int32, device=dev)
        self._mcorr = torch.empty(o, hw, device=dev)
        self._app_pos = torch.empty(o, hw, dtype=torch.int32, device=dev)
        self._part = torch.empty(o, MAX_SPLIT_SCAN, hw, 2, device=dev)
        self._work = torch.zeros(4, dtype=torch.int32, device=dev)          # queue head of the persistent scan kernel
        self._stats_pinned = torch.zeros(o, 4, dtype=torch.int32).pin_memory()
        self._scratch = None
        self._klp = self._vlp = None
        self._norms_valid = self._lp_valid = False

    def _ensure_scratch(self):
        if self._scratch is None:
            self._scratch = (torch.empty_like(self._kbuf), torch.empty_like(self._vbuf), torch.empty_like(self._ibuf))
        return self._scratch

    def _bank_desc(self):
        o, cap = self.obj_n, self._cap
        bd = BankDesc()
        bd.bank_k, bd.bank_v, bd.info = ptr(self._kbuf), ptr(self._vbuf), ptr(self._ibuf)
        bd.bank_len, bd.bank_len_rw = ptr(self._len_dev), ptr(self._len_dev)
        bd.stride_k, bd.stride_v, bd.stride_info, bd.stride_n = cap * DK, cap * DV, cap * 2, cap
        bd.HW, bd.obj_n, bd.cap = self._hw, o, cap
        return bd

    def lp_image(self):
        """Device pointers (keys, values) of the bank's split-bf16 image for the reduced-precision kernels
        (``vfn_bank_refresh_lp``: per entry keys [128 hi | 128 lo] bf16; values in blocks of 8 entries,
        [hi | lo plane][512 channels][8 entries] bf16, the B operand of the P^T V MFMAs as it is loaded -- the bytes of the f32
        rows again), brought up to date first.  ``update`` re-splits only the entries it changed; anything else that
        touches the bank (``keys`` / ``values`` views, ``append``, ``remove``) makes the next call rebuild it.
        ``VFN_LP_IMAGE=0``: (None, None) -- the kernels then split their operands on the fly (same results, slower)."""
        import os
        if os.environ.get('VFN_LP_IMAGE', '1') == '0' or self._kbuf is None:
            return None, None
        if self._klp is None:
            o, cap = self.obj_n, self._cap
            # zero-initialised, one chunk of slack: every row a kernel can address holds finite values
            self._klp = torch.zeros(o * cap * DK * 2 + CH * DK * 2, dtype=torch.int16, device=self.device)
            self._vlp = torch.zeros(o * cap * DV * 2 + CH * DV * 2, dtype=torch.int16, device=self.device)
            self._lp_valid = False
        if not self._lp_valid:
            bd = self._bank_desc()
            check(_lib.lib().vfn_bank_refresh_lp(_lib.C.byref(bd), ptr(self._klp), ptr(self._vlp), 1, stream()),
                  'vfn_bank_refresh_lp')
            self._lp_valid = True
        return self._klp, self._vlp

    # ------------------------------------------------------------------ lengths
    def _sync_len(self):
        if self._dirty:
            self._stats_pinned.copy_(self._stats, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            self.absorb_stats(self._stats_pinned)
        return self._len_host

    def absorb_stats(self, stats_host, in_flight=0):
        """Take a host copy of the device ``stats`` block (len, peak, replace, n_append per object).
        ``in_flight``: updates already enqueued behind the one these statistics describe (a pipelined loop):
        the upper bound the grids are sized with keeps room for them."""
        st = stats_host.numpy()
        self._len_host = [int(st[i, 0]) for i in range(self.obj_n)]
        self._len_upper = [min(n + in_flight * self._hw, self._cap) for n in self._len_host]
        self._hist[self._n_updates - in_flight] = list(self._len_host)
        for k in [k for k in self._hist if k < self._n_updates - 2]:
            del self._hist[k]
        for i in range(self.obj_n):
            self.peak_n[i] = max(self.peak_n[i], float(st[i, 1]))
            self.replace_n[i] = float(st[i, 2])
        self._dirty = in_flight > 0
        for i in range(self.obj_n):
            if st[i, 3] < 0:                         # flagged by bank_plan_kernel; that object was left untouched
                code = int(st[i, 3])
                self._stats[i, 3] = 0
                if code == -1:                       # int(LFU.min()) with a NaN score (0 hits / 0 age), FeatureBank.py:123
                    raise ValueError('cannot convert float NaN to integer (FeatureBank.remove: an entry born at '
                                     f'frame_idx with no hits, object {i})')
                raise OverflowError(f'cannot convert float infinity to integer (FeatureBank.remove, object {i})')

    def stats_device(self):
        return self._stats

    def len_upper(self):
        """Upper bound of the bank lengths that is valid without a synchronisation.  The kernels read the true lengths
        from device memory; this bound only sizes grids and picks the bank slicing of the scans / the memory read.
        The slicing fixes the order in which partial sums meet, so the bound is made a function of the UPDATE HISTORY
        alone, not of how far the host has got with its bookkeeping: with U updates enqueued it is (lengths after update
        U-1) + HW -- known both to a sequential loop (which has absorbed update U) and to a pipelined one (which has
        absorbed update U-1 while update U is in flight).  ``ClipRunner.step`` and ``launch`` / ``collect`` therefore
        slice alike and give bit-identical results.  (Anything deeper in flight falls back to the safe running bound.)"""
        u = self._n_updates
        if u == 0 and 0 in self._hist:
            return max(self._hist[0])
        prev = self._hist.get(u - 1)
        if prev is not None:
            return min(max(prev) + self._hw, self._cap)
        return max(self._len_upper)

    # ------------------------------------------------------------------ reference attributes
    @property
    def keys(self):
        if self._kbuf is None:
            return None
        n = self._sync_len()
        self._norms_valid = self._lp_valid = False           # the views are writable: a caller may edit entries
        return [self._kbuf[i, :n[i]].t() for i in range(self.obj_n)]

    @keys.setter
    def keys(self, value):
        if value is None:
            self._kbuf = None
            return
        raise AttributeError('assign through init_bank()/append(): the bank lives in pre-allocated device slabs')

    @property
    def values(self):
        if self._vbuf is None:
            return None
        n = self._sync_len()
        self._norms_valid = self._lp_valid = False
        return [self._vbuf[i, :n[i]].t() for i in range(self.obj_n)]

    @values.setter
    def values(self, value):
        if value is None:
            self._vbuf = None
            return
        raise AttributeError('assign through init_bank()/append(): the bank lives in pre-allocated device slabs')

    @property
    def info(self):
        if self._ibuf is None:
            return [None for _ in range(self.obj_n)]
        n = self._sync_len()
        return [self._ibuf[i, :n[i]] for i in range(self.obj_n)]

    # ------------------------------------------------------------------ API
    @torch.no_grad()
    def _write_columns(self, keys, values, start, frame_idx, hit_init):
        """Copy per-object [128,n] / [512,n] columns into the slabs at row ``start[i]`` (plumbing copies)."""
        for i in range(self.obj_n):
            n = keys[i].shape[1]
            s = start[i]
            self._kbuf[i, s:s + n].copy_(keys[i].t())
            self._vbuf[i, s:s + n].copy_(values[i].t())
            self._ibuf[i, s:s + n, 0] = float(frame_idx)
            self._ibuf[i, s:s + n, 1] = float(hit_init)

    def init_bank(self, keys, values, frame_idx=0):
        """FeatureBank.py:27-36."""
        self._require_gpu()
        hw = keys[0].shape[1]
        n_init = max(k.shape[1] for k in keys)
        self._alloc(hw, n_init)
        self._write_columns(keys, values, [0] * self.obj_n, frame_idx, 0.0)
        lens = [int(k.shape[1]) for k in keys]
        self._set_lengths(lens)
        # training (train_video_seg.py:66-69): the keys / values memorize returned are nodes of an autograd graph; the slabs hold
        # their values, these references carry the gradient that segment's backward sends to the bank back into memorize
        # (vfloodnet_amd.autograd).  Any later change of the bank drops them.
        self._graph_kv = (list(keys), list(values)) if any(getattr(t_, 'requires_grad', False) for t_ in list(keys) + list(values)) else None

    def _set_lengths(self, lens):
        self._len_host = list(lens)
        self._len_upper = list(lens)
        self._n_updates = 0                  # updates enqueued since the lengths were last set from the host
        self._hist = {0: list(lens)}         # update count -> exact lengths after that update (the last three)
        for i in range(self.obj_n):
            self.peak_n[i] = max(self.peak_n[i], lens[i])
        # through pinned memory, without blocking: a copy from pageable memory holds the host until the stream has reached it -- in the
        # training step that was the end of memorize's forward pass, once per step, with the device idle behind it (round 5)
        st = torch.zeros(self.obj_n, 4, dtype=torch.int32, pin_memory=True)
        for i in range(self.obj_n):
            st[i, 0] = lens[i]
            st[i, 1] = int(self.peak_n[i])
            st[i, 2] = int(self.replace_n[i])
        self._stats.copy_(st, non_blocking=True)
        ln = torch.empty(len(lens), dtype=torch.int32, pin_memory=True)
        for i, v in enumerate(lens):
            ln[i] = v
        self._len_dev.copy_(ln, non_blocking=True)
        self._dirty = False

    def append(self, keys, values, frame_idx=0):
        """FeatureBank.py:38-51 (append-only entry point; hit accumulator starts at 20)."""
        if self._kbuf is None:
            return self.init_bank(keys, values, frame_idx)
        lens = self._sync_len()
        need = max(lens[i] + keys[i].shape[1] for i in range(self.obj_n))
        if need > self._cap:
            self._grow(need)
        self._write_columns(keys, values, lens, frame_idx, 20.0)
        self._graph_kv = None
        self._norms_valid = self._lp_valid = False
        self._set_lengths([lens[i] + int(keys[i].shape[1]) for i in range(self.obj_n)])

    def _grow(self, need):
        cap = (int(need * 1.5) + self._hw + 63) // 64 * 64
        lens = self._len_host
        for name in ('_kbuf', '_vbuf', '_ibuf', '_knorm', '_vnorm', '_kinv', '_cnt', '_keep_dst'):
            old = getattr(self, name)
            new = torch.zeros(old.shape[0], cap, *old.shape[2:], dtype=old.dtype, device=old.device)
            new[:, :self._cap] = old
            setattr(self, name, new)
        self._cap = cap
        self._scratch = None
        self._klp = self._vlp = None
        self._lp_valid = False
        del lens

    def _stage_new(self, prev_key, prev_value):
        """Return (tensor, ld) holding the new features as [obj][HW][640] (key | value)."""
        hw = prev_key[0].shape[1]
        k0 = prev_key[0]
        ld = DK + DV
        ok = k0.is_cuda and k0.dtype == torch.float32 and k0.stride() == (1, ld) and hw == self._hw
        if ok:
            base = k0.data_ptr()
            for i in range(self.obj_n):
                k, v = prev_key[i], prev_value[i]
                if (k.stride() != (1, ld) or v.stride() != (1, ld) or k.shape != (DK, hw) or v.shape != (DV, hw)
                        or k.data_ptr() != base + i * hw * ld * 4 or v.data_ptr() != k.data_ptr() + DK * 4):
                    ok = False
                    break
        if ok:
            return k0, ld            # already the engine's [obj][HW][640] slab: zero-copy
        for i in range(self.obj_n):
            self._new[i, :, :DK].copy_(prev_key[i].t())
            self._new[i, :, DK:].copy_(prev_value[i].t())
        return self._new, ld

    def update(self, prev_key, prev_value, frame_idx, update_rate=-1):
        """FeatureBank.py:53-115: cosine match -> merge (scatter_mean + blend) / append / LFU evict."""
        self._require_gpu()
        self._graph_kv = None
        if update_rate == -1:
            update_rate = self.update_rate
        hw = prev_key[0].shape[1]
        if hw != self._hw:
            raise RuntimeError(f'feature count changed ({hw} vs {self._hw}): one bank serves one frame size')
        if hw > MAX_HW:
            raise RuntimeError(f'{hw} features per frame exceed the bank kernels\' limit of {MAX_HW} (VFN_BANK_MAX_HW)')
        L = _lib.lib()
        o, cap = self.obj_n, self._cap
        new, ld = self._stage_new(prev_key, prev_value)
        s = stream()

        # norms: bank keys / values (device length) -- carried from the previous update when nothing else has touched the
        # bank since (vfn_bank_refresh_norms below), recomputed in full otherwise -- and new keys / values
        if not self._norms_valid:
            ops.row_norms(self._kbuf, cap * DK, DK, DK, self._len_dev, 0, o, self._knorm, self._kinv, cap)
            ops.row_norms(self._vbuf, cap * DV, DV, DV, self._len_dev, 0, o, self._vnorm, None, cap)
        ops.row_norms(new, hw * ld, ld, DK, None, hw, o, self._nknorm, self._nkinv, hw)
        check(L.vfn_row_norms(_lib.C.c_void_p(new.data_ptr() + DK * 4), hw * ld, ld, DV, None, hw, o,
                              ptr(self._nvnorm), None, hw, s), 'vfn_row_norms')

        # cosine arg-max over the bank per new feature
        b_up = self.len_upper()
        nsplit = pick_scan_slices(hw, o, b_up)
        d = BankScanDesc()
        d.q, d.bank_k, d.bank_len, d.rowscale, d.part = ptr(new), ptr(self._kbuf), ptr(self._len_dev), ptr(self._kinv), ptr(self._part)
        d.stride_q, d.stride_k, d.stride_rs = hw * ld, cap * DK, cap
        d.scale = 1.0
        d.ldq, d.q_per_obj, d.HW, d.obj_n, d.nsplit, d.mode = ld, 1, hw, o, nsplit, 1
        d.precision = ops.MODES[self.precision]
        d.work_counter = ptr(self._work)
        if d.precision:
            klp, _ = self.lp_image()
            d.bank_k_lp = ptr(klp) if klp is not None else None
        check(L.vfn_bank_scan(_lib.C.byref(d), s), 'vfn_bank_scan')
        check(L.vfn_bank_scan_finish(ptr(self._part), nsplit, hw, o, 1, None, ptr(self._midx), ptr(self._mcorr),
                                     ptr(self._nkinv), s), 'vfn_bank_scan_finish')

        # merge + append (+ evict)
        may_evict = self.class_budget < b_up + hw
        if may_evict:
            sk, sv, si = self._ensure_scratch()
        else:
            sk, sv, si = self._kbuf, self._vbuf, self._ibuf       # never touched: no eviction possible
        bd = BankDesc()
        bd.bank_k, bd.bank_v, bd.info = ptr(self._kbuf), ptr(self._vbuf), ptr(self._ibuf)
        bd.scratch_k, bd.scratch_v, bd.scratch_info = ptr(sk), ptr(sv), ptr(si)
        bd.bank_len, bd.bank_len_rw = ptr(self._len_dev), ptr(self._len_dev)
        bd.bank_knorm, bd.bank_vnorm = ptr(self._knorm), ptr(self._vnorm)
        bd.match_idx, bd.match_corr = ptr(self._midx), ptr(self._mcorr)
        bd.new_k, bd.new_knorm, bd.new_vnorm = ptr(new), ptr(self._nknorm), ptr(self._nvnorm)
        bd.app_pos, bd.keep_dst, bd.plan, bd.stats = ptr(self._app_pos), ptr(self._keep_dst), ptr(self._plan), ptr(self._stats)
        bd.stride_k, bd.stride_v, bd.stride_info, bd.stride_n, bd.stride_new = cap * DK, cap * DV, cap * 2, cap, hw * ld
        bd.class_budget = float(self.class_budget)
        bd.thres_close, bd.update_rate, bd.new_hit_init = float(self.thres_close), float(update_rate), 0.0
        bd.frame_idx, bd.ld_new, bd.voff, bd.HW, bd.obj_n, bd.cap = int(frame_idx), ld, DK, hw, o, cap
        bd.rm_class, bd.rm_request = -1, 0
        check(L.vfn_bank_merge(_lib.C.byref(bd), s), 'vfn_bank_merge')
        check(L.vfn_bank_append(_lib.C.byref(bd), s), 'vfn_bank_append')
        check(L.vfn_bank_refresh_norms(_lib.C.byref(bd), ptr(self._knorm), ptr(self._kinv), ptr(self._vnorm), s),
              'vfn_bank_refresh_norms')
        self._norms_valid = True
        if self._klp is not None and self._lp_valid:     # re-split the merged / appended entries only
            check(L.vfn_bank_refresh_lp(_lib.C.byref(bd), ptr(self._klp), ptr(self._vlp), 0, s), 'vfn_bank_refresh_lp')

        self._dirty = True
        self._len_upper = [min(n + hw, cap) for n in self._len_upper]
        self._n_updates += 1

    def remove(self, class_idx, request_n, frame_idx):
        """FeatureBank.py:117-143: LFU eviction of one object until ``class_budget - bank_n - request_n >= 0``;
        returns that balance.  (``update`` runs the same plan fused with the append; this entry point is the
        reference's public method.)"""
        self._require_gpu()
        self._graph_kv = None
        L = _lib.lib()
        o, cap, hw = self.obj_n, self._cap, self._hw
        self._sync_len()
        sk, sv, si = self._ensure_scratch()
        bd = BankDesc()
        bd.bank_k, bd.bank_v, bd.info = ptr(self._kbuf), ptr(self._vbuf), ptr(self._ibuf)
        bd.scratch_k, bd.scratch_v, bd.scratch_info = ptr(sk), ptr(sv), ptr(si)
        bd.bank_len, bd.bank_len_rw = ptr(self._len_dev), ptr(self._len_dev)
        bd.match_corr = ptr(self._mcorr)
        bd.app_pos, bd.keep_dst, bd.plan, bd.stats = ptr(self._app_pos), ptr(self._keep_dst), ptr(self._plan), ptr(self._stats)
        bd.stride_k, bd.stride_v, bd.stride_info, bd.stride_n, bd.stride_new = cap * DK, cap * DV, cap * 2, cap, 0
        bd.class_budget = float(self.class_budget)
        bd.thres_close, bd.update_rate, bd.new_hit_init = float(self.thres_close), float(self.update_rate), 0.0
        bd.frame_idx, bd.ld_new, bd.voff, bd.HW, bd.obj_n, bd.cap = int(frame_idx), DK + DV, DK, hw, o, cap
        bd.rm_class, bd.rm_request = int(class_idx), int(request_n)
        check(L.vfn_bank_remove(_lib.C.byref(bd), stream()), 'vfn_bank_remove')
        self._norms_valid = self._lp_valid = False
        self._dirty = True
        n = self._sync_len()
        return (self.class_budget - n[class_idx]) - request_n

    # ------------------------------------------------------------------ snapshot / resume
    def state_dict(self, device='cpu'):
        """The bank as plain tensors and numbers, for ``torch.save``: a long stream can stop and pick up again
        (``load_state_dict``) on the frame after.  The reference keeps no inference state (``FeatureBank.py:10-51`` holds
        lists of live tensors only; SURVEY §5 lists the snapshot as optional).  What is kept is what the next frame depends
        on: the live entries in slab order (keys, values, ``info`` = birth frame + hit accumulator), the peak / replace
        statistics, and the update history the slicing of the scans is derived from (``len_upper``) -- with it a resumed run
        slices the bank exactly as the uninterrupted one and gives bit-identical results.  Norms and the split-bf16 image
        are functions of the entries and are rebuilt.  Synchronises (it reads the device lengths)."""
        if self._kbuf is None:
            raise RuntimeError('FeatureBank.state_dict: the bank is empty (init_bank first)')
        n = self._sync_len()
        dev = torch.device(device)
        take = lambda t: t.detach().to(dev, copy=True)
        return dict(version=1, obj_n=self.obj_n, hw=self._hw, dk=DK, dv=DV, class_budget=float(self.class_budget),
                    update_rate=float(self.update_rate), thres_close=float(self.thres_close), precision=self.precision,
                    lens=[int(x) for x in n], n_updates=int(self._n_updates),
                    hist={int(k): [int(x) for x in v] for k, v in self._hist.items()},
                    peak_n=[float(x) for x in self.peak_n], replace_n=[float(x) for x in self.replace_n],
                    keys=[take(self._kbuf[i, :n[i]]) for i in range(self.obj_n)],
                    values=[take(self._vbuf[i, :n[i]]) for i in range(self.obj_n)],
                    info=[take(self._ibuf[i, :n[i]]) for i in range(self.obj_n)])

    @torch.no_grad()
    def load_state_dict(self, sd):
        """Bring a bank written by ``state_dict`` back (entries as [n, 128] / [n, 512] / [n, 2] rows per object).  Budget,
        update rate and merge threshold stay those this bank was constructed with (a stream may be resumed under a
        different budget); the object count and the feature geometry must match."""
        if sd.get('version') != 1:
            raise ValueError(f'FeatureBank.load_state_dict: unknown snapshot version {sd.get("version")!r}')
        if sd['obj_n'] != self.obj_n or sd['dk'] != DK or sd['dv'] != DV:
            raise ValueError(f'FeatureBank.load_state_dict: snapshot of {sd["obj_n"]} objects with {sd["dk"]} / {sd["dv"]} channels, '
                             f'bank of {self.obj_n} with {DK} / {DV}')
        lens = [int(x) for x in sd['lens']]
        for i in range(self.obj_n):
            k, v, inf = sd['keys'][i], sd['values'][i], sd['info'][i]
            if tuple(k.shape) != (lens[i], DK) or tuple(v.shape) != (lens[i], DV) or tuple(inf.shape) != (lens[i], 2):
                raise ValueError(f'FeatureBank.load_state_dict: object {i}: entries {tuple(k.shape)} / {tuple(v.shape)} / '
                                 f'{tuple(inf.shape)} do not match the recorded length {lens[i]}')
        self._alloc(int(sd['hw']), max(lens))
        for i in range(self.obj_n):
            self._kbuf[i, :lens[i]].copy_(sd['keys'][i])
            self._vbuf[i, :lens[i]].copy_(sd['values'][i])
            self._ibuf[i, :lens[i]].copy_(sd['info'][i])
        self.peak_n = np.maximum(self.peak_n, np.asarray(sd['peak_n'], dtype=np.float64))
        self.replace_n = np.asarray(sd['replace_n'], dtype=np.float64).copy()
        self._set_lengths(lens)
        self._n_updates = int(sd['n_updates'])
        self._hist = {int(k): [int(x) for x in v] for k, v in sd['hist'].items()}
        self._graph_kv = None

    def print_peak_mem(self):
        """FeatureBank.py:145-149."""
        self._sync_len()
        ur = self.peak_n / self.class_budget
        rr = self.replace_n / self.class_budget
        print(f'Obj num: {self.obj_n}.', f'Budget / obj: {self.class_budget}.', f'UR: {ur}.', f'Replace: {rr}.')
