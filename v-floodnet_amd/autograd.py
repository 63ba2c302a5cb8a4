"""The training boundary of the reference, unchanged: ``torch.autograd`` around the HIP forward / backward.

``train_video_seg.py:65-74`` reads

    k4_list, v4_list = model.memorize(frames[0:1], masks[0:1]);  fb_global.init_bank(k4_list, v4_list)
    scores, uncertainty = model.segment(frames[1:], fb_global)
    loss = criterion(scores, label) + args.lu * uncertainty
    optimizer.zero_grad();  loss.backward();  optimizer.step()          # torch.optim.AdamW (:109)

With the model in training mode and autograd on, ``AFB_URR.memorize`` / ``segment`` return tensors that are nodes of an
autograd graph: two ``torch.autograd.Function`` s whose forward is the HIP forward (``engine.Engine``) and whose backward is
the HIP backward (``backward.ModelBackward``: every parameter's gradient, and the gradient that reaches the bank's keys /
values, which flows on into ``memorize``'s node through the references ``FeatureBank.init_bank`` keeps).  The criterion, the
optimizer and the scheduler are the caller's own torch objects, as in the reference.  ``vfloodnet_amd.train.train_step`` is the
same step without the graph (fused loss kernel, flat-buffer AdamW); both are held against the reference's own step
(tests/golden/train_step_96x160.npz, oracle/gen_train_golden.py).

Activations: the backward reads the forward's activation buffers, which the next forward through the same plan overwrites.
A batch (``frames[1:]`` holds clip_n - 1 samples) runs its samples one after the other through those buffers, so the backward
RE-RUNS the forward of every sample but the one that ran last (activation recomputation: + one forward per such sample);
``memorize`` is re-run only if another ``memorize`` came in between.
"""
import torch

from . import ops
from .engine import DK, DV


def trainable(model):
    """(names, parameters) that take gradients, in ``model.named_parameters()`` order (what torch.optim sees)."""
    items = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    return [n for n, _ in items], [p for _, p in items]


def wants_graph(model):
    return torch.is_grad_enabled() and model.training and any(p.requires_grad for p in model.parameters())


def _param_grads(names, params, grads):
    out = []
    for n, p in zip(names, params):
        g = grads.get(n)
        out.append(None if g is None else g.reshape(p.shape))
    return tuple(out)


class _Memorize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, frame, mask, *params):
        eng = model.engine()
        with torch.no_grad():
            k_list, v_list = eng.memorize(frame, mask, training=True)
        ctx.model, ctx.frame, ctx.mask = model, frame.detach(), mask.detach()
        ctx.K = len(k_list)
        ctx.token = eng.mem_count
        return tuple(k_list) + tuple(v_list)

    @staticmethod
    @torch.no_grad()
    def backward(ctx, *g):
        from .backward import ModelBackward
        model = ctx.model
        eng = model.engine()
        K = ctx.K
        if eng.mem_count != ctx.token:                     # another memorize overwrote the activations: run this one again
            eng.memorize(ctx.frame, ctx.mask, training=True)
        plan = eng.last_memorize
        zk = lambda: torch.zeros(plan.HW, DK, device=eng.device)
        zv = lambda: torch.zeros(plan.HW, DV, device=eng.device)
        g_bk = [g[i].t().contiguous() if g[i] is not None else zk() for i in range(K)]             # [128,HW] -> [HW,128]
        g_bv = [g[K + i].t().contiguous() if g[K + i] is not None else zv() for i in range(K)]
        mb = eng.backward()
        mb.finish_memorize(ctx.frame, ctx.mask, g_bk, g_bv)
        names, params = trainable(model)
        return (None, None, None) + _param_grads(names, params, mb.grads)


class _Segment(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, fb, frames, n_kv, *rest):
        eng = model.engine()
        with torch.no_grad():
            score = eng.segment(frames, fb, model.update_bank, training=True)
            score = score.clone() if frames.shape[0] == 1 else score      # (bs = 1 returns the plan's own buffer)
            unc = ops.segment_uncertainty(score.contiguous())
        ctx.model, ctx.fb, ctx.frames, ctx.n_kv = model, fb, frames.detach(), n_kv
        ctx.token = eng.fwd_count
        ctx.save_for_backward(score)
        return score, unc

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g_score, g_unc):
        from .backward import ModelBackward
        model, fb, frames = ctx.model, ctx.fb, ctx.frames
        (score,) = ctx.saved_tensors
        eng = model.engine()
        bs, K = frames.shape[0], fb.obj_n
        if g_unc is None:
            g_unc = torch.zeros((), device=eng.device)
        # dL/dscores in total: the caller's criterion (g_score) + the uncertainty's adjoint, one kernel, no host round trip
        total = ops.segment_uncertainty_backward(score.contiguous(), g_unc.reshape(1).float().contiguous(), g_score)
        mb = eng.backward()
        g_bk = g_bv = None
        # the sample that ran last still has its activations in the plan; the others are run again (see the module docstring)
        order = [bs - 1] + list(range(bs - 1)) if eng.fwd_count == ctx.token else list(range(bs))
        for n_done, b in enumerate(order):
            if not (n_done == 0 and eng.fwd_count == ctx.token):
                eng.segment(frames[b:b + 1], fb, False, training=True)
            bk, bv = mb.segment_sample(fb, total[b])
            if g_bk is None:
                g_bk, g_bv = list(bk), list(bv)
            else:
                g_bk = [a + c for a, c in zip(g_bk, bk)]
                g_bv = [a + c for a, c in zip(g_bv, bv)]
        kv_grads = ()
        if ctx.n_kv:                                       # the bank's keys / values as memorize returned them: [128,HW] / [512,HW]
            kv_grads = tuple(x.t() for x in g_bk) + tuple(x.t() for x in g_bv)
        names, params = trainable(model)
        return (None, None, None, None) + kv_grads + _param_grads(names, params, mb.grads)


def memorize(model, frame, mask):
    _, params = trainable(model)
    out = _Memorize.apply(model, frame, mask, *params)
    K = len(out) // 2
    return list(out[:K]), list(out[K:])


def segment(model, frames, fb):
    _, params = trainable(model)
    kv = []
    if getattr(fb, '_graph_kv', None) is not None:
        kv = list(fb._graph_kv[0]) + list(fb._graph_kv[1])
    return _Segment.apply(model, fb, frames, len(kv), *kv, *params)
