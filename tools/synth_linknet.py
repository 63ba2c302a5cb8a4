"""Synthetic, name-keyed weights for the bootstrap model (oracle/linknet_ref.py): the trained pickle
``records/link_efficientb4_model.pth`` (test_video_seg.py:68) is not available and there is no network.  Every tensor is drawn
from a generator seeded by its state-dict name; the BatchNorm running statistics are then calibrated on one synthetic frame so
that the 32 random blocks stay at unit scale (the same recipe as tools/synth.py for AFB_URR)."""
import math
import zlib

import torch

from oracle import linknet_ref as R


def _gen_for(name, seed):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ int(seed)) & 0x7FFFFFFF)


def make_state_dict(seed=20200212, H=416, W=416):
    sd = {}
    for name, shp in R.template().items():
        g = _gen_for(name, seed)
        if name.endswith('num_batches_tracked'):
            sd[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith('running_mean'):
            sd[name] = torch.zeros(shp)
        elif name.endswith('running_var'):
            sd[name] = torch.ones(shp)
        elif len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            gain = 1.0 if '_se_' in name else 1.6                      # swish / relu roughly halve the variance
            sd[name] = gain * torch.randn(shp, generator=g) / math.sqrt(fan_in)
        elif '_se_reduce.bias' in name or '_se_expand.bias' in name:
            sd[name] = 0.2 * torch.randn(shp, generator=g)
        elif name.endswith('.bias') and ('bn' in name.split('.')[-2] or name.split('.')[-2] == '1'):
            sd[name] = 0.1 * torch.randn(shp, generator=g)              # BatchNorm beta
        elif name.endswith('.weight'):
            sd[name] = 1 + 0.1 * torch.randn(shp, generator=g)          # BatchNorm gamma
        else:
            sd[name] = 0.05 * torch.randn(shp, generator=g)             # convolution biases
    # residual branches start small (as zero-init gamma would): keeps 32 blocks from random-walking
    for i, b in enumerate(R.blocks()):
        if b['s'] == 1 and b['cin'] == b['cout']:
            sd[f'encoder._blocks.{i}._bn2.weight'] *= 0.5
    sd['segmentation_head.0.weight'] = sd["segmentation_head.0.weight"] * 1.0
    x = frame(seed, H, W)
    with torch.no_grad():
        R.forward(sd, x, calibrate=True)
    return sd


def frame(seed, H, W):
    """An ImageNet-normalised synthetic frame [1,3,H,W]: smooth gradient + texture (tools/synth.py's recipe)."""
    from tools import synth
    f0, _ = synth.frame0(seed & 0xffff, H, W)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return (f0.unsqueeze(0) - mean) / std
