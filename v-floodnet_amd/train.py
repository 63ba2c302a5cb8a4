"""One optimisation step of ``train_video_seg.py`` on the HIP path.

The reference's inner loop (``train_video_seg.py:56-76``):

    fb_global = FeatureBank(obj_n, budget, device)
    k4_list, v4_list = model.memorize(frames[0:1], masks[0:1]);  fb_global.init_bank(k4_list, v4_list)
    scores, uncertainty = model.segment(frames[1:], fb_global)
    label = torch.argmax(masks[1:], dim=1).long()
    optimizer.zero_grad();  loss = criterion(scores, label) + lu * uncertainty;  loss.backward();  optimizer.step()

with ``criterion = CrossEntropyLoss()`` (:162), ``optimizer = torch.optim.AdamW(params, lr)`` (:109), BatchNorm frozen
(:103-106, ``myutils.set_bn_eval``: running statistics, affine parameters still trained), ``update_bank=False`` (:101).

Here: ``train_step`` runs that body.  The bank is fixed while the batch is segmented, so the batch's samples are independent
given the bank: the part of ``segment`` that depends on the frame alone (query encoder, KeyValue, the decoder's skip branches) runs
for all samples in one pass (``Engine.query_batch``) and is differentiated in one pass (``ModelBackward.finish_query``); the rest runs
sample by sample: each sample is segmented, its loss gradient formed (``ops.segment_loss``; the batch means of the cross entropy
and of the uncertainty become a factor 1/bs on every sample's gradient) and carried back to the parameters and to the
bank's keys / values (``backward.ModelBackward.segment_sample``; two activation plans alternate, so the next sample's forward runs
while the side stream still computes this sample's weight gradients); the
gradients that reached the bank are summed over the samples and go through ``memorize`` once (``finish_memorize``).
``AdamW`` keeps every parameter of the model in ONE flat f32 buffer (the ``nn.Parameter``s become views of it) next to
flat ``exp_avg`` / ``exp_avg_sq`` buffers, so a step is one ``vfn_adamw_f32`` launch over 38 M floats (HBM-bound: 5 floats
moved per parameter).  Every convolution, reduction, adjoint and the optimizer run in the HIP library; there is no autograd graph
and no eager fallback.  What is left to tensor operators is glue (profiles/r04_train_kernel_stats.csv: ~500 small launches, ~2 ms
of a 31 ms step): concatenations / slices of gradients that cross a layer boundary, the sum over the objects, the zero-padded
operands of the memory read's five small GEMMs, the stem's input normalisation.

Across steps nothing is rebuilt: ``train_step`` ends with ``model._refresh()`` (``engine.Engine.refresh``: everything derived from
the parameters -- packed filters, data-gradient filters, Winograd banks, folded BatchNorm constants -- rewritten in place by two
launches), the engine's plans and the cached backward pass (``Engine.backward()``) persist, and the step's gradients reach the
optimizer's flat buffer in one launch (``AdamW.set_grads``).
"""
import os
import time

import torch

from . import _lib, ops
from ._lib import ptr, stream, check
from .feature_bank import FeatureBank
from .backward import ModelBackward


class AdamW:
    """``torch.optim.AdamW(params, lr, betas, eps, weight_decay)`` (train_video_seg.py:109; torch defaults: betas (0.9, 0.999),
    eps 1e-8, weight_decay 1e-2, no amsgrad) over the parameters of one ``AFB_URR``.

    ``named_params``: iterable of (state-dict name, nn.Parameter) -- ``model.named_parameters()``.  The parameters are moved
    into one flat device buffer (their ``.data`` become views of it, values unchanged)."""

    def __init__(self, named_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        items = [(n, p) for n, p in named_params if p.requires_grad]
        if not items:
            raise ValueError('optimizer got an empty parameter list')
        dev = items[0][1].device
        if dev.type != 'cuda':
            raise RuntimeError('AdamW runs on the HIP device only (no CPU fallback): move the model to the GPU first')
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.names = [n for n, _ in items]
        self.offsets = {}
        off = 0
        for n, p in items:
            if p.dtype != torch.float32:
                raise TypeError(f'{n}: f32 parameters only')
            self.offsets[n] = (off, p.numel(), tuple(p.shape))
            off += (p.numel() + 3) // 4 * 4                                # 16-byte aligned views
        self.n = off
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        for n, p in items:
            o, k, shp = self.offsets[n]
            view = self.flat[o:o + k].view(shp)
            view.copy_(p.data)
            p.data = view
        self.grad = torch.zeros_like(self.flat)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = 0
        self._gather = None            # (device table of set_grads)
        # torch's schedulers keep the un-decayed rate in the param group ('initial_lr'); StepLR below sets it, state_dict()
        # carries it, so that a resumed run decays from the right base and a checkpoint written here resumes in the reference
        # (torch's StepLR(last_epoch != -1) raises KeyError without it)
        self.initial_lr = None

    def zero_grad(self):
        self.grad.zero_()

    def owns(self, model):
        """Are the model's optimised parameters still the views of this optimizer's flat buffer (nobody re-seated ``p.data``)?
        One address per parameter against the recorded offset."""
        base = self.flat.data_ptr()
        for n, p in model.named_parameters():
            off = self.offsets.get(n)
            if off is not None and p.data_ptr() != base + 4 * off[0]:
                return False
        return True

    def grad_view(self, name):
        o, k, shp = self.offsets[name]
        return self.grad[o:o + k].view(shp)

    def param_view(self, name):
        o, k, shp = self.offsets[name]
        return self.flat[o:o + k].view(shp)

    def set_grads(self, grads):
        """grads: state-dict name -> tensor (``ModelBackward.grads``).  Every optimised parameter must have one."""
        missing = [n for n in self.names if n not in grads]
        if missing:
            raise KeyError(f'no gradient for {missing[:4]} ({len(missing)} parameters)')
        # One launch for all of them (vfn_gather_strided_f32): a weight gradient arrives as a strided [Cout,Cin,kh,kw] view of its
        # packed accumulator, the rest as small vectors; the table of (address, strides, offset in the flat buffer) lives on the
        # device and is re-used while the tensors keep their addresses (the caching allocator hands the same blocks out every step).
        key, ts = [], []
        for n in self.names:
            g = grads[n]
            o, k, shp = self.offsets[n]
            if g.numel() != k or g.dim() > 4 or g.dtype != torch.float32 or not g.is_cuda:
                raise ValueError(f'gradient of {n}: {tuple(g.shape)} {g.dtype} for a parameter of shape {shp}')
            ts.append((g, o))
            key.append((g.data_ptr(), tuple(g.shape), g.stride()))
        if self._gather is None or self._gather[0] != key:
            from ._lib import GatherEntry
            L = _lib.lib()
            per_block = L.vfn_refresh_elems_per_block()
            tab = (GatherEntry * len(ts))()
            block = 0
            for i, (g, o) in enumerate(ts):
                e = tab[i]
                e.src, e.dst_offset, e.block0 = g.data_ptr(), o, block
                pad = 4 - g.dim()
                for d_ in range(4):
                    e.shape[d_] = 1 if d_ < pad else g.shape[d_ - pad]
                    e.stride[d_] = 0 if d_ < pad else g.stride(d_ - pad)
                block += (g.numel() + per_block - 1) // per_block
            dev_tab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.flat.device)
            self._gather = (key, dev_tab, len(ts), block)
        _, dev_tab, n_, blocks = self._gather
        check(_lib.lib().vfn_gather_strided_f32(ptr(dev_tab), n_, blocks, ptr(self.grad), stream()), 'vfn_gather_strided_f32')

    def step(self):
        self.step_count += 1
        check(_lib.lib().vfn_adamw_f32(ptr(self.flat), ptr(self.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.n, self.lr,
                                       self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count, stream()),
              'vfn_adamw_f32')

    # checkpoint pieces of train_video_seg.py:186-193 ('optimizer': optimizer.state_dict()) and :129 (optimizer.load_state_dict): the
    # dictionary has torch.optim.AdamW's own layout -- parameters numbered in model.parameters() order -- so a checkpoint written by
    # the reference resumes here and the other way round
    def state_dict(self):
        state = {}
        for i, n in enumerate(self.names):
            o, k, shp = self.offsets[n]
            state[i] = {'step': torch.tensor(float(self.step_count)), 'exp_avg': self.exp_avg[o:o + k].view(shp).clone(),
                        'exp_avg_sq': self.exp_avg_sq[o:o + k].view(shp).clone()}
        group = {'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': self.weight_decay, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'params': list(range(len(self.names)))}
        if self.initial_lr is not None:
            group['initial_lr'] = self.initial_lr
        return {'state': state if self.step_count else {}, 'param_groups': [group]}

    def load_state_dict(self, sd):
        groups = sd['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self.names):
            raise ValueError(f"loaded state dict has {sum(len(g['params']) for g in groups)} parameters in {len(groups)} group(s), "
                             f'this optimizer has {len(self.names)} in one')
        g = groups[0]
        if g.get('amsgrad') or g.get('maximize'):
            raise ValueError('amsgrad / maximize are not implemented (train_video_seg.py:109 uses neither)')
        self.lr, self.betas, self.eps, self.weight_decay = float(g['lr']), (float(g['betas'][0]), float(g['betas'][1])), float(g['eps']), float(g['weight_decay'])
        # (a group written without a scheduler has no 'initial_lr': it stays unset, so that StepLR(last_epoch != -1) raises KeyError on
        # it exactly as torch.optim.lr_scheduler does -- a resumed run is never silently re-based on an already decayed rate)
        self.initial_lr = float(g['initial_lr']) if 'initial_lr' in g else None
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, n in enumerate(self.names):
            st = sd['state'].get(g['params'][i])
            if st is None:
                continue
            o, k, shp = self.offsets[n]
            self.exp_avg[o:o + k].view(shp).copy_(st['exp_avg'])
            self.exp_avg_sq[o:o + k].view(shp).copy_(st['exp_avg_sq'])
            steps.add(int(st['step']))                      # (an int before torch 1.12, a tensor since)
        if len(steps) > 1:
            raise ValueError(f'parameters with different step counts {sorted(steps)}: one flat buffer steps them together')
        self.step_count = steps.pop() if steps else 0


@torch.no_grad()
def forward_backward(model, frames, masks, lu=0.5, budget=300000):
    """The loop body of train_video_seg.py:56-74 up to and including ``loss.backward()``.

    frames f32 [T,3,H,W] in [0,1], masks [T,obj_n,H,W] one-hot (the dataset's sample with the leading 1 removed, :63);
    the model must be on the GPU and in training mode (``model.train()``, update_bank False).
    Returns (loss, uncertainty, grads): python floats as ``loss.item()`` / ``uncertainty.item()`` give them, and
    state-dict name -> gradient tensor for every parameter."""
    _check_step_inputs(model, frames, masks)
    with _host_single_threaded():
        return _forward_backward(model, frames, masks, lu, budget)


def _check_step_inputs(model, frames, masks):
    if not model.training:
        raise RuntimeError('forward_backward needs model.train() (the training branch of segment keeps its activations)')
    if model.update_bank:
        raise RuntimeError('training runs with update_bank=False (train_video_seg.py:101)')
    T, obj_n = frames.shape[0], masks.shape[1]
    if T < 2:
        raise ValueError('a training sample needs a reference frame and at least one frame to segment')
    if obj_n < 2:      # as model.segment: the reference fails in calc_uncertainty's top-2 (myutils/data.py:40-46)
        raise RuntimeError('segment needs at least two objects (background + 1): selected index k out of range')


last_enqueue_s = 0.0
_stats_host = {}                 # device -> pinned f32[3] the step's (loss, cross entropy, uncertainty) land in
_BATCH_QUERY = os.environ.get('VFN_TRAIN_BATCH_QUERY', '1') == '1'   # the query encoder over all frames of a sample at once (fwd + bwd)
_BATCH_DECODER = os.environ.get('VFN_TRAIN_BATCH_DECODER', '1') == '1'   # ... and the decoder (fwd + bwd); needs the batched query encoder


def _forward_backward(model, frames, masks, lu, budget, lazy=False):
    t_start = time.perf_counter()
    T, obj_n = frames.shape[0], masks.shape[1]
    dev = model.device
    frames, masks = frames.to(dev), masks.to(dev)
    bs = T - 1
    fb = FeatureBank(obj_n, budget, dev)
    k4_list, v4_list = model.memorize(frames[0:1], masks[0:1])
    fb.init_bank(k4_list, v4_list)
    label = torch.argmax(masks[1:], dim=1)
    eng = model.engine()
    if _BATCH_QUERY:
        eng.query_batch(frames[1:], obj_n)       # the frame-only part of segment for all bs frames at once
    mb = eng.backward()
    g_bk = g_bv = None
    stats_sum = torch.zeros(3, device=dev)
    if _BATCH_QUERY and _BATCH_DECODER:
        # round 5: the bank-dependent half too -- one memory read per frame, then the decoder forward and backward once over
        # frames x objects (engine.DecoderBatch); the criterion over the whole batch is train_video_seg.py:72-74 as written
        scores = eng.segment_batch(fb)
        stats, dscore = ops.segment_loss(scores, label, lu)
        stats_sum = stats[:3] * bs
        mb.segment_batch(fb, dscore)
        bs_loop = 0
    else:
        bs_loop = bs
    for i in range(bs_loop):
        score, _ = model.segment(frames[1 + i:2 + i], fb)
        stats, dscore = ops.segment_loss(score.contiguous(), label[i:i + 1], lu)
        stats_sum += stats[:3]
        if bs > 1:
            dscore *= 1.0 / bs                                              # mean over the batch: CE over bs*H*W pixels, uncertainty.mean()
        bk, bv = mb.segment_sample(fb, dscore[0])
        if bk is None:                           # (batched sample: the memory read is differentiated in finish_query)
            continue
        if g_bk is None:
            g_bk, g_bv = bk, bv
        else:
            g_bk = [a + b for a, b in zip(g_bk, bk)]
            g_bv = [a + b for a, b in zip(g_bv, bv)]
    bk, bv = mb.finish_query()                   # (the memory read's and the query encoder's backward of the batched samples, all frames at once)
    if bk is not None:
        assert g_bk is None
        g_bk, g_bv = bk, bv
    mb.finish_memorize(frames[0:1], masks[0:1], g_bk, g_bv)
    global last_enqueue_s
    last_enqueue_s = time.perf_counter() - t_start                          # host time to enqueue the whole step (bench_train_step.py)
    if lazy:
        # train_step: the statistics travel to pinned host memory behind the backward pass and an event marks their arrival; the host
        # reads them after it has enqueued the optimizer and the refresh, WITHOUT waiting for those -- the next step's first launches
        # are built while the device still runs AdamW and the filter refresh (0.7 ms), instead of after them
        host = _stats_host.get(dev)
        if host is None:
            host = _stats_host[dev] = torch.empty(3, dtype=torch.float32, pin_memory=True)
        host.copy_(stats_sum / bs, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return (host, ev), mb.grads
    st = (stats_sum / bs).tolist()                                          # one D2H per step, as loss.item() is
    return st[0], st[2], mb.grads


class _host_single_threaded:
    """The step's host side is ~6 700 kernel launches and a few hundred tiny CPU tensor operations (filter packing after every
    optimiser step).  One CPU tensor operation above torch's parallel grain wakes the whole OpenMP pool (256 threads on the GPU
    box), whose spinning starves the launching thread: 230-390 ms per step against 116 ms with intra-op parallelism off.  Nothing
    in the step has CPU work worth a second thread, so it runs with torch.set_num_threads(1) and restores the setting."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        if self.n != 1:
            torch.set_num_threads(1)

    def __exit__(self, *exc):
        if self.n != 1:
            torch.set_num_threads(self.n)
        return False


@torch.no_grad()
def train_step(model, optimizer, frames, masks, lu=0.5, budget=300000):
    """train_video_seg.py:56-76 for one sample of the dataloader.  Returns (loss, uncertainty) as python floats."""
    with _host_single_threaded():
        optimizer.zero_grad()
        _check_step_inputs(model, frames, masks)
        stats, grads = _forward_backward(model, frames, masks, lu, budget, lazy=True)
        optimizer.set_grads(grads)
        optimizer.step()
        # the engine's packed filters / folded BatchNorm constants follow in place; the address check of the refresh tables is
        # skipped once they have been verified under THIS optimizer's ownership of the parameters (its constructor re-seats p.data)
        own = optimizer.owns(model)
        model._refresh(trusted=own and model.__dict__.get('_refresh_owner') is optimizer)
        model.__dict__['_refresh_owner'] = optimizer if own else None
        model.engine()               # (or are rebuilt here, inside the single-threaded region, if a parameter moved)
        # the step's one device-to-host read, AFTER the optimizer and the refresh have been enqueued: read right behind the backward
        # pass (round 4) it left the device idle while the host built the optimizer's launches (0.4 + 0.7 ms in a kernel trace)
        host, ev = stats
        ev.synchronize()
        st = host.tolist()
    return st[0], st[2]


class StepLR:
    """``torch.optim.lr_scheduler.StepLR(optimizer, step_size, gamma, last_epoch)`` for ``AdamW`` above
    (train_video_seg.py:146-147,181), in torch's own (chainable) form: the constructor records the optimizer's
    ``initial_lr`` (a fresh run) or requires it (``last_epoch != -1``: a resumed optimizer, whose ``lr`` is the already
    decayed rate ``load_state_dict`` restored) and takes one step; a step multiplies the CURRENT rate by gamma whenever the new
    epoch is a positive multiple of step_size.  So ``optimizer.load_state_dict(ckpt['optimizer'])`` followed by
    ``StepLR(..., last_epoch=start_epoch - 1)`` (train_video_seg.py:129,146) continues the schedule instead of decaying twice."""

    def __init__(self, optimizer, step_size, gamma=0.1, last_epoch=-1):
        self.opt, self.step_size, self.gamma = optimizer, int(step_size), float(gamma)
        if last_epoch == -1:
            if getattr(optimizer, 'initial_lr', None) is None:
                optimizer.initial_lr = optimizer.lr
        elif getattr(optimizer, 'initial_lr', None) is None:
            raise KeyError("param 'initial_lr' is not specified in param_groups[0] when resuming an optimizer")
        self.base_lr = optimizer.initial_lr
        self.last_epoch = last_epoch
        self.step()

    @property
    def epoch(self):
        return self.last_epoch

    def get_last_lr(self):
        return [self.opt.lr]

    def step(self):
        self.last_epoch += 1
        if self.last_epoch != 0 and self.last_epoch % self.step_size == 0:
            self.opt.lr = self.opt.lr * self.gamma


def train_model(model, dataloader, optimizer, lu=0.5, budget=300000, progress=None):
    """``train_model`` of train_video_seg.py:51-89: one pass over the dataloader; samples are
    (frames [1,T,3,H,W], masks [1,T,obj_n,H,W], obj_n, info) as ``Water_Image_Train_DS`` yields them; single-object samples are
    skipped (:61-62).  Returns the mean loss (``stats.avg``)."""
    n, loss_sum = 0, 0.0
    for sample in dataloader:
        frames, masks, obj_n = sample[0], sample[1], sample[2]
        obj_n = int(obj_n.item() if torch.is_tensor(obj_n) else obj_n)
        if obj_n == 1:
            continue
        loss, unc = train_step(model, optimizer, frames[0], masks[0], lu, budget)
        n += 1
        loss_sum += loss
        if progress is not None:
            progress(n, loss, loss_sum / n, unc)
    return loss_sum / max(n, 1)
