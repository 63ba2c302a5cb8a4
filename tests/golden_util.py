"""Shared helpers for the golden-vector tests (fixtures made from the reference by oracle/gen_golden.py)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SEED = 20200212
_SD = None


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def meta():
    return json.load(open(os.path.join(GOLDEN, 'meta.json')))


def checksum(t):
    t = t.double()
    return [float(t.sum()), float(t.abs().sum()), float((t * t).sum())]


def state_dict():
    """The synthetic checkpoint, regenerated from the seed (132 MB: not committed) and checked against
    the checksums recorded when the fixtures were generated."""
    global _SD
    if _SD is None:
        import vfloodnet_amd  # noqa: F401
        from tools import synth
        _SD = synth.make_state_dict(SEED)
        for k, ref in meta()['weights_checksum'].items():
            got = checksum(_SD[k])
            assert np.allclose(got, ref, rtol=1e-4, atol=1e-6), f'synthetic weights drifted at {k}: {got} vs {ref}'
    return _SD


def t(a):
    return torch.from_numpy(np.asarray(a))


def close_logits(s, ref, atol=1e-3, patol=5e-7):
    """|dlogit| < atol, or |dsigmoid| < patol where logit is ill-conditioned (|logit| near the clamp)."""
    dl = (s - ref).abs()
    dp = (torch.sigmoid(s) - torch.sigmoid(ref)).abs()
    return bool(((dl < atol) | (dp < patol)).all()), float(dl.max()), float(dp.max())


def miou(a, b):
    v = []
    for c in (0, 1):
        inter = ((a == c) & (b == c)).sum().item()
        union = ((a == c) | (b == c)).sum().item()
        v.append(1.0 if union == 0 else inter / union)
    return sum(v) / 2


# ---- the reference's own training step (oracle/gen_train_golden.py -> tests/golden/train_step_96x160.npz)
TRAIN_H, TRAIN_W, TRAIN_K, TRAIN_LU, TRAIN_LR, TRAIN_NS = 96, 160, 2, 0.5, 1e-5, 16


def train_sample():
    """The 3-frame sample the fixture was generated on (clip seed 6, 5 % label noise): frames [3,3,H,W], masks [3,K,H,W], lab."""
    import vfloodnet_amd  # noqa: F401
    from tools import synth
    frames, m0 = synth.clip(6, 3, TRAIN_H, TRAIN_W)
    gen = torch.Generator().manual_seed(11)
    lab = torch.stack([m0.long()] + [torch.roll(m0.long(), (2 * k_, 5 * k_), (0, 1)) for k_ in (1, 2)], 0)
    flip = torch.rand(3, TRAIN_H, TRAIN_W, generator=gen) < 0.05
    lab = torch.where(flip, 1 - lab, lab)
    masks = torch.nn.functional.one_hot(lab, TRAIN_K).permute(0, 3, 1, 2).float()
    return frames, masks, lab


def train_positions(numel, name):
    import zlib
    g = np.random.default_rng(zlib.crc32(name.encode()))
    return np.sort(g.integers(0, numel, size=TRAIN_NS)).astype(np.int64)


def train_names():
    return open(os.path.join(GOLDEN, 'train_step_names.txt')).read().split()


def compare_grads_with_reference(grads, g, names=None):
    """``grads``: state-dict name -> gradient tensor (any device / dtype).  Per tensor, against the reference's own backward:
    relative error of the L2 norm, of the 16 sampled elements and (for the 24 tensors stored whole) of every element, each
    relative to the tensor's largest gradient magnitude.  Returns name -> worst of them."""
    names = names or train_names()
    worst = {}
    for i, n in enumerate(names):
        x = grads[n].detach().double().cpu().flatten()
        nrm, _sum, amax = g['grad_stats'][i]
        scale = max(amax, 1e-30)
        e = abs(float(x.norm()) - nrm) / max(nrm, 1e-30)
        idx = torch.from_numpy(train_positions(x.numel(), n))
        e = max(e, float((x[idx] - torch.from_numpy(g['grad_samples'][i]).double()).abs().max()) / scale)
        if 'full_grad.' + n in g:
            e = max(e, float((x - torch.from_numpy(g['full_grad.' + n]).double().flatten()).abs().max()) / scale)
        worst[n] = e
    return worst
