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
