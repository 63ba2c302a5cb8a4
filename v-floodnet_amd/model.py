"""``AFB_URR`` facade: the reference's model API on top of the HIP engine.

Drop-in for ``video_module.model.AFB_URR`` (``AFB_URR.py:242-321``) as used by
``test_video_seg.py:42-44,100,108,111``:

    model = AFB_URR(device, update_bank=True, load_imagenet_params=False)
    model = model.to(device); model.eval()
    model.load_state_dict(checkpoint['model'], strict=False)       # 562 reference keys
    k4_list, v4_list = model.memorize(frame, mask)                  # lists of [128,HW] / [512,HW]
    score, _ = model.segment(frame, fb)                             # logits [bs,obj_n,h,w]; bumps fb.info

The module owns parameters only (``weights.py``); ``memorize`` / ``segment`` run
entirely in hand-written HIP kernels through ``engine.Engine``.  There is no
eager / CPU fallback: on a non-GPU device, or without the compiled library,
both calls raise ``RuntimeError`` (the exception type the reference's matcher
already handles, ``AFB_URR.py:147``).
"""
import torch
from torch import nn

from . import weights as W


class AFB_URR(nn.Module):
    def __init__(self, device, update_bank, load_imagenet_params=False, _allow_cpu_container=False, precision=None):
        super().__init__()
        # 'fp32' (default; exact-f32 matrix cores, the parity configuration), 'bf16x3' (operands split into two bf16,
        # three bf16 MFMAs per product: ~2^-16 relative) or 'bf16' (operands rounded to bf16: 2^-9); f32 accumulation
        # and f32 tensors in every mode.  For BASELINE configs C3 / C5; the reference has no reduced-precision mode.
        # VFN_PRECISION sets the default.
        import os
        self.precision = precision or os.environ.get('VFN_PRECISION', 'fp32')
        if load_imagenet_params:
            # AFB_URR.py:39,69 would download torchvision ImageNet weights; inference always
            # overwrites them from a checkpoint (test_video_seg.py:51) and there is no network.
            raise RuntimeError('load_imagenet_params=True is a training-time option and is not supported')
        self.device = torch.device(device)
        self.update_bank = bool(update_bank)
        self.encoder_m = W.make_encoder_m()
        self.encoder_q = W.make_encoder_q()
        self.keyval_r4 = W.make_keyval(1024, 128, 512)
        self.decoder = W.make_decoder()
        self._engine = None
        self._engine_version = -1
        self._engine_sentinel = -1
        self._sentinels = None
        self._eval_calls = 0
        self._allow_cpu_container = _allow_cpu_container

    # -- weight lifecycle ---------------------------------------------------
    def _invalidate(self):
        self._engine = None
        self._sentinels = None

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._invalidate()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._invalidate()
        try:
            self.device = next(self.parameters()).device
        except StopIteration:  # pragma: no cover
            pass
        return out

    def _refresh(self, trusted=False):
        """The parameters were updated in place: the engine's derived tensors follow them where they lie (engine.Engine.refresh:
        two kernel launches; plans, buffers and the cached backward pass stay) -- or, if a parameter moved to another dtype /
        device / layout, the engine is dropped and rebuilt on the next call."""
        if self._engine is None:
            return
        try:
            self._engine.refresh(check=not trusted)       # (trusted: the caller owns the parameters' storage, refresh.Refresher.run)
            self._engine_version = self._param_version()
            self._engine_sentinel = self._sentinel_version()
        except RuntimeError:
            self._engine = None

    def _param_version(self):
        return sum(p._version for p in self.parameters())

    FULL_CHECK_EVERY = 32      # eval-mode calls between two comparisons of EVERY parameter's version counter

    def _sentinel_version(self):
        """Version counters of the first and the last parameter: what the eval-mode calls compare (an optimizer step moves every
        parameter, so two of them tell; summing all 300 on every ``memorize`` / ``segment`` costs the inference loop 0.2 ms per frame)."""
        if self._sentinels is None:
            ps = list(self.parameters())
            self._sentinels = (ps[0], ps[-1])
        return self._sentinels[0]._version + self._sentinels[1]._version

    def train(self, mode=True):
        """``model.eval()`` after the last ``optimizer.step()`` (torch's own AdamW on the nn.Parameters) must
        not leave the engine one step behind: the switch compares every parameter's version counter and refreshes the derived tensors."""
        out = super().train(mode)
        if self._engine is not None and self._engine_version != self._param_version():
            self._refresh()
        return out

    def engine(self):
        # an optimizer that steps the nn.Parameters in place (torch.optim.AdamW, train_video_seg.py:109) leaves the engine's packed
        # filters / folded BatchNorm constants stale -- every in-place update bumps the tensors' version counters.  Training mode
        # compares all of them on every call, eval mode the two sentinels (and train() / eval() all of them once more)
        if self._engine is not None:
            if self.training:
                stale = self._engine_version != self._param_version()
            else:
                # the two sentinels on every call; every FULL_CHECK_EVERY-th call all of them (an in-place update that
                # touches neither the first nor the last parameter -- fine-tuning the middle of the network, a manual p.copy_ on
                # one layer -- would otherwise go unnoticed until the next train() / eval()): 0.2 ms / 32 = 6 us per call.  Between
                # two full checks such an update is served by the old packed filters for at most FULL_CHECK_EVERY - 1 calls;
                # ``model.eval()`` (or ``model.train(False)``) right after the update closes that window at once.
                self._eval_calls += 1
                stale = self._engine_sentinel != self._sentinel_version()
                if not stale and self._eval_calls % self.FULL_CHECK_EVERY == 0:
                    stale = self._engine_version != self._param_version()
            if stale:
                self._refresh()
        if self._engine is None:
            from .engine import Engine
            self._engine = Engine(self)
            self._engine_version = self._param_version()
            self._engine_sentinel = self._sentinel_version()
        return self._engine

    # -- reference API ------------------------------------------------------
    def memorize(self, frame, mask):
        """AFB_URR.py:255-272.  frame f32[1,3,h,w] in [0,1]; mask [1,K,h,w] (u8 or float).  Identical in eval and in
        training mode (the reference pads here in both, :259); BatchNorm always uses its running statistics -- in
        training mode that is the frozen-BN setting of train_video_seg.py:103-106.  In training mode with autograd on the
        returned keys / values are nodes of an autograd graph (vfloodnet_amd.autograd), as train_video_seg.py:66-74 expects."""
        from . import autograd as A
        if A.wants_graph(self):
            return A.memorize(self, frame, mask)
        with torch.no_grad():
            return self.engine().memorize(frame, mask, training=self.training)

    def segment(self, frame, fb_global):
        if fb_global.obj_n >= 2:
            from . import autograd as A
            if A.wants_graph(self):
                # train_video_seg.py:69-74: ``scores, uncertainty`` carry the graph; ``loss.backward()`` runs the HIP backward
                return A.segment(self, frame, fb_global)
        return self._segment_no_grad(frame, fb_global)

    @torch.no_grad()
    def _segment_no_grad(self, frame, fb_global):
        """AFB_URR.py:274-318, forward only.  frame f32[bs,3,h,w]; returns (logits f32[bs,obj_n,h,w], uncertainty).
        eval: pads to a multiple of 16, uncertainty is None (test_video_seg.py:108).  After ``model.train()``: the
        training branch -- no padding (:278) and the scalar uncertainty of :302-305 as a 0-dim tensor
        (train_video_seg.py:69,73-74) -- with BatchNorm frozen as train_video_seg.py:103-106 sets it.  (This is the graph-free
        form, used under ``torch.no_grad()`` and by ``vfloodnet_amd.train.train_step``, which differentiates the activations
        this call keeps; with autograd on, ``segment`` goes through ``vfloodnet_amd.autograd``.)"""
        if fb_global.obj_n < 2:
            # the reference fails here as well: calc_uncertainty takes the top-2 over the object axis
            # (myutils/data.py:40-46, `score.topk(k=2, dim=1)` -> "selected index k out of range")
            raise RuntimeError('segment needs at least two objects (background + 1): selected index k out of range')
        score = self.engine().segment(frame, fb_global, self.update_bank, training=self.training)
        if self.training:
            from . import ops
            return score, ops.segment_uncertainty(score.contiguous())
        return score, None

    @torch.no_grad()
    def segment_group(self, frames, fb_global, prefetch=None):
        """``segment`` for G consecutive frames f32[G,3,h,w] that see the same bank (the frames between two ``memorize`` calls
        when only every n-th frame is memorised): one batched pass, logits f32[G,obj_n,h,w] (Engine.segment_group).  Eval mode only;
        an extension -- the reference segments frame by frame (test_video_seg.py:108)."""
        if self.training:
            raise RuntimeError('segment_group is the inference loop\'s batching; the training step batches through train.train_step')
        if fb_global.obj_n < 2:
            raise RuntimeError('segment needs at least two objects (background + 1): selected index k out of range')
        return self.engine().segment_group(frames, fb_global, self.update_bank, prefetch=prefetch)

    def forward(self, x):  # AFB_URR.py:320-321
        pass
