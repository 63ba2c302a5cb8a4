"""Seeded synthetic inputs: clips, first-frame masks and a calibrated checkpoint.

The reference ships neither its pretrained checkpoint
(``test_video_seg.py:28`` -> ``records/video_seg_checkpoint_20200212-001734.pth``)
nor redistributable frames, and there is no network, so every test / benchmark
runs on data generated here:

* ``clip(seed, T, H, W)``: frame 0 = smooth RGB gradient + sinusoidal
  water/land boundary + uniform noise texture; frame t = frame 0 rolled by
  (2t, 5t) pixels.  ``first_mask`` = the analytic boundary (1 = water).
* ``make_checkpoint(seed)``: name-keyed random weights (``N(0, 1/fan_in)``
  convolutions, BN gamma ``1+0.1 N``, beta / biases ``0.05 N``) whose BatchNorm
  running statistics are set by one batch-statistics pass of a synthetic frame
  through both encoders, saved in the reference's checkpoint schema
  ``{'epoch','model','loss','seed'}`` (``train_video_seg.py:159-177``).  With
  PyTorch's default init the net saturates at the logit clamp; this recipe
  gives logits in roughly [-14, 8] with a healthy margin distribution.

The calibration pass below is data generation (plain torch CPU ops), not the
product forward; the product forward is the HIP engine.
"""
import math
import zlib

import torch
from torch.nn import functional as F

from . import weights as W

CALIB_HW = (240, 432)


def frame0(seed, H, W_):
    g = torch.Generator().manual_seed(int(seed))
    ys = torch.linspace(0, 1, H).view(H, 1)
    xs = torch.linspace(0, 1, W_).view(1, W_)
    boundary = 0.55 + 0.12 * torch.sin(2 * math.pi * (1.5 * xs + 0.13 * seed)) \
        + 0.05 * torch.sin(2 * math.pi * (4.0 * xs + 0.29 * seed))
    water = (ys > boundary).float()                      # [H,W], 1 below the shoreline
    base = torch.stack([0.25 + 0.5 * xs.expand(H, W_),
                        0.30 + 0.4 * ys.expand(H, W_),
                        0.55 - 0.3 * (xs * ys)], 0)
    tint = torch.tensor([-0.15, -0.05, 0.20]).view(3, 1, 1)
    noise = torch.rand(3, H, W_, generator=g) - 0.5
    img = base + water.unsqueeze(0) * tint + 0.25 * noise * (0.5 + 0.5 * water.unsqueeze(0))
    return img.clamp(0, 1).contiguous(), water.to(torch.uint8).contiguous()


def clip(seed, T, H, W_):
    """-> frames f32[T,3,H,W] in [0,1], first_mask u8[H,W] (1 = water)."""
    f0, m0 = frame0(seed, H, W_)
    frames = torch.stack([torch.roll(f0, shifts=(2 * t, 5 * t), dims=(1, 2)) for t in range(T)], 0)
    return frames.contiguous(), m0


def onehot(mask_u8, obj_n=2):
    """``ToOnehot`` semantics (transforms.py:383-421): channel 0 = 1 - sum(objects)."""
    m = torch.zeros(obj_n, *mask_u8.shape, dtype=torch.uint8)
    for i in range(1, obj_n):
        m[i] = (mask_u8 == i).to(torch.uint8)
    m[0] = 1 - m[1:].sum(0).to(torch.uint8)
    return m


# --------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------
def _gen_for(name, seed):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ int(seed)) & 0x7FFFFFFF)


def random_state_dict(template_sd, seed):
    sd = {}
    for name, t in template_sd.items():
        g = _gen_for(name, seed)
        if name.endswith('num_batches_tracked'):
            sd[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith('.mean') or name.endswith('.std'):
            sd[name] = t.detach().clone().float().cpu()
        elif name.endswith('running_mean'):
            sd[name] = torch.zeros(t.shape)
        elif name.endswith('running_var'):
            sd[name] = torch.ones(t.shape)
        elif t.dim() == 4:
            fan_in = t.shape[1] * t.shape[2] * t.shape[3]
            sd[name] = torch.randn(t.shape, generator=g) / math.sqrt(fan_in)
        elif '.bn' in name or 'downsample.1' in name:
            if name.endswith('weight'):
                sd[name] = 1 + 0.1 * torch.randn(t.shape, generator=g)
            else:
                sd[name] = 0.05 * torch.randn(t.shape, generator=g)
        else:  # conv biases
            sd[name] = 0.05 * torch.randn(t.shape, generator=g)
    return sd


def _calib_bn(x, sd, prefix):
    """Batch-statistics BN (train mode, momentum=None -> stats = this batch), records them."""
    mean = x.mean(dim=(0, 2, 3))
    var_b = x.var(dim=(0, 2, 3), unbiased=False)
    n = x.numel() / x.shape[1]
    sd[prefix + '.running_mean'] = mean.clone()
    sd[prefix + '.running_var'] = (var_b * n / max(n - 1, 1)).clone()   # running_var is unbiased
    y = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var_b.view(1, -1, 1, 1) + W.BN_EPS)
    return y * sd[prefix + '.weight'].view(1, -1, 1, 1) + sd[prefix + '.bias'].view(1, -1, 1, 1)


def _calib_layer(x, sd, prefix, blocks, stride):
    for b in range(blocks):
        p = f'{prefix}.{b}'
        s = stride if b == 0 else 1
        out = F.relu(_calib_bn(F.conv2d(x, sd[p + '.conv1.weight']), sd, p + '.bn1'))
        out = F.relu(_calib_bn(F.conv2d(out, sd[p + '.conv2.weight'], stride=s, padding=1), sd, p + '.bn2'))
        out = _calib_bn(F.conv2d(out, sd[p + '.conv3.weight']), sd, p + '.bn3')
        if b == 0:
            idn = _calib_bn(F.conv2d(x, sd[p + '.downsample.0.weight'], stride=s), sd, p + '.downsample.1')
        else:
            idn = x
        x = F.relu(out + idn)
    return x


def _calibrate(sd, seed):
    H, W_ = CALIB_HW
    f0, m0 = frame0(seed, H, W_)
    oh = onehot(m0).float()
    mean = sd['encoder_q.mean']
    std = sd['encoder_q.std']
    f = (f0.unsqueeze(0) - mean) / std
    # query encoder
    x = F.conv2d(f, sd['encoder_q.conv1.weight'], stride=2, padding=3)
    x = F.relu(_calib_bn(x, sd, 'encoder_q.bn1'))
    x = F.max_pool2d(x, 3, 2, 1)
    x = _calib_layer(x, sd, 'encoder_q.res2', 3, 1)
    x = _calib_layer(x, sd, 'encoder_q.res3', 4, 2)
    _calib_layer(x, sd, 'encoder_q.res4', 6, 2)
    # memory encoder (two objects: background, water)
    fm = f.expand(2, -1, -1, -1)
    m = oh.unsqueeze(1)
    o = (1 - m).clamp(0, 1)
    x = F.conv2d(fm, sd['encoder_m.conv1.weight'], stride=2, padding=3) \
        + F.conv2d(m, sd['encoder_m.conv1_m.weight'], stride=2, padding=3) \
        + F.conv2d(o, sd['encoder_m.conv1_o.weight'], stride=2, padding=3)
    x = F.relu(_calib_bn(x, sd, 'encoder_m.bn1'))
    x = F.max_pool2d(x, 3, 2, 1)
    x = _calib_layer(x, sd, 'encoder_m.res2', 3, 1)
    x = _calib_layer(x, sd, 'encoder_m.res3', 4, 2)
    _calib_layer(x, sd, 'encoder_m.res4', 6, 2)


def make_state_dict(seed=20200212):
    """Deterministic calibrated state dict (CPU tensors, reference key names)."""
    from .model import AFB_URR
    with torch.no_grad():
        tmpl = AFB_URR(torch.device('cpu'), update_bank=True, _allow_cpu_container=True).state_dict()
        sd = random_state_dict(tmpl, seed)
        nthr = torch.get_num_threads()
        torch.set_num_threads(1)           # fixed reduction order -> bit-reproducible statistics
        try:
            _calibrate(sd, seed)
        finally:
            torch.set_num_threads(nthr)
    return sd


def make_checkpoint(path, seed=20200212):
    """Write a reference-schema checkpoint (train_video_seg.py:159-177)."""
    sd = make_state_dict(seed)
    torch.save({'epoch': 0, 'model': sd, 'loss': 0.0, 'seed': int(seed)}, path)
    return path
