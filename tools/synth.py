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

from vfloodnet_amd import weights as W

CALIB_HW = (240, 426)          # padded to 240x432 by pad_divide_by, like 480x854 -> 480x864


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
    # multi-octave texture (cell sizes 1,2,4,...,32 px): local statistics are similar at every
    # frame size, so BatchNorm statistics calibrated at one size carry over to the others
    noise = torch.zeros(3, H, W_)
    for o in range(6):
        c = 1 << o
        n = torch.rand(1, 3, (H + c - 1) // c + 1, (W_ + c - 1) // c + 1, generator=g) - 0.5
        if c > 1:
            n = F.interpolate(n, scale_factor=c, mode='bilinear', align_corners=False)
        noise += n[0, :, :H, :W_]
    noise = noise / math.sqrt(6.0) * 1.4
    img = base + water.unsqueeze(0) * tint + 0.25 * noise * (0.5 + 0.5 * water.unsqueeze(0))
    return img.clamp(0, 1).contiguous(), water.to(torch.uint8).contiguous()


def frame0_hard(seed, H, W_):
    """A frame whose water is NOT separable by colour (round 6, VERDICT r5 item 5: ``frame0``'s tinted water trains to saturated
    logits -- median |margin| at the +-16 clamp -- which says little about bf16 on a real checkpoint).  Same shoreline and base
    gradient as ``frame0``; the two regions share their colour statistics up to a 0.03 tint and differ in TEXTURE only: the water's
    multi-octave noise is smeared horizontally (a 9-pixel box along x: streaks), the land's is isotropic, both at a lower contrast
    under independent pixel noise.  -> (img f32[3,H,W] in [0,1], water u8[H,W])."""
    g = torch.Generator().manual_seed(int(seed) * 7919 + 17)
    ys = torch.linspace(0, 1, H).view(H, 1)
    xs = torch.linspace(0, 1, W_).view(1, W_)
    boundary = 0.55 + 0.12 * torch.sin(2 * math.pi * (1.5 * xs + 0.13 * seed)) \
        + 0.05 * torch.sin(2 * math.pi * (4.0 * xs + 0.29 * seed))
    water = (ys > boundary).float()
    # (no vertical gradient: with frame0's base the water, which lies below the shoreline, is greener than the land -- a colour cue)
    lowf = F.interpolate(torch.rand(1, 3, H // 64 + 3, W_ // 64 + 3, generator=g) - 0.5, scale_factor=64, mode='bilinear', align_corners=False)
    base = 0.45 + 0.10 * xs.expand(3, H, W_) + 0.25 * lowf[0, :, 32:32 + H, 32:32 + W_]
    noise = torch.zeros(3, H, W_)
    for o in range(6):
        c = 1 << o
        n = torch.rand(1, 3, (H + c - 1) // c + 1, (W_ + c - 1) // c + 1, generator=g) - 0.5
        if c > 1:
            n = F.interpolate(n, scale_factor=c, mode='bilinear', align_corners=False)
        noise += n[0, :, :H, :W_]
    noise = noise / math.sqrt(6.0) * 1.4
    streak = F.avg_pool2d(F.pad(noise.unsqueeze(0), (4, 4, 0, 0), mode='circular'), (1, 9), stride=1)[0] * 1.8
    w3 = water.unsqueeze(0)
    tint = torch.tensor([-0.02, 0.0, 0.03]).view(3, 1, 1)
    pix = (torch.rand(3, H, W_, generator=g) - 0.5) * 0.10
    img = base + w3 * tint + 0.22 * (w3 * streak + (1 - w3) * noise) + pix
    return img.clamp(0, 1).contiguous(), water.to(torch.uint8).contiguous()


def hard_step(H, W_, net_size=480):
    """Pixels (dy, dx) a ``clip_hard`` frame moves per time step: (2, 5) at the network's resolution; for an enlarged clip a multiple
    of the enlargement's numerator (720p = 3/2: (3, 6); 1080p = 9/4: (9, 18)), so that every frame comes back from the loop's resize
    on the SAME sampling phase -- with (2, 5) at 720p only every third frame did, and the texture task (whose cue is a few pixels
    wide) was segmented at mIoU 0.9 on those frames and 0.3 on the others."""
    short = min(H, W_)
    if short <= net_size:
        return (2, 5)
    from fractions import Fraction
    num = Fraction(short, net_size).numerator
    return (num, 2 * num)


def clip_hard(seed, T, H, W_, device=None, net_size=480, in_place=False):
    """``clip`` / ``clip_on_device`` over ``frame0_hard``.  The hard task lives in the TEXTURE, whose scale the loop's resize to a
    ``net_size``-pixel short edge (test_video_seg.py:46,107) changes: a frame larger than that is synthesised at the network's
    resolution and enlarged (bicubic; the mask nearest), so that the network sees the texture statistics it was trained on
    (synthesised at 1080p and shrunk 2.25 x, the f32 network itself reached mIoU 0.30 against the ground truth -- a clip on which no
    precision question can be asked)."""
    short = min(H, W_)
    if short > net_size:
        h, w = (net_size, int(net_size * W_ / H)) if H <= W_ else (int(net_size * H / W_), net_size)
        f0, m0 = frame0_hard(seed, h, w)
        f0 = F.interpolate(f0.unsqueeze(0), size=(H, W_), mode='bicubic', align_corners=False)[0].clamp(0, 1).contiguous()
        m0 = F.interpolate(m0.view(1, 1, h, w).float(), size=(H, W_), mode='nearest')[0, 0].to(torch.uint8).contiguous()
    else:
        f0, m0 = frame0_hard(seed, H, W_)
    if device is not None:
        f0 = f0.to(device)
    dy, dx = hard_step(H, W_, net_size)
    if in_place:                              # (a long high-resolution stream: built frame by frame in device memory, as clip_on_device)
        frames = torch.empty(T, 3, H, W_, device=f0.device)
        for t in range(T):
            frames[t] = torch.roll(f0, shifts=(dy * t, dx * t), dims=(1, 2))
        return frames, m0
    frames = torch.stack([torch.roll(f0, shifts=(dy * t, dx * t), dims=(1, 2)) for t in range(T)], 0)
    return frames.contiguous(), m0


def noisy_labels(mask, gen, band=28, p_flip=0.35, max_shift=24):
    """Annotation noise along the shoreline (training labels of the hard task): the mask is displaced vertically by a random
    -max_shift .. max_shift pixels and, inside a band of +-``band`` pixels around ITS boundary, pixels are flipped with probability
    ``p_flip`` -- about 10 % of a 400 x 400 frame's labels disagree with the image.  mask: long / u8 [H,W] (1 = water) on any device;
    ``gen``: a host generator (the noise is drawn on the host: bit-reproducible).  -> same dtype / device."""
    H, W_ = mask.shape
    sh = int(torch.randint(-max_shift, max_shift + 1, (1,), generator=gen))
    m = torch.roll(mask, sh, 0)
    mf = m.float().view(1, 1, H, W_)
    k = 2 * band + 1
    local = F.avg_pool2d(F.pad(mf, (0, 0, band, band), mode='replicate'), (k, 1), stride=1)[0, 0]
    near = (local > 0.02) & (local < 0.98)                     # within ``band`` rows of the (displaced) boundary
    flip = (torch.rand(H, W_, generator=gen) < p_flip).to(mask.device) & near
    return torch.where(flip, 1 - m, m)


def clip(seed, T, H, W_):
    """-> frames f32[T,3,H,W] in [0,1], first_mask u8[H,W] (1 = water)."""
    f0, m0 = frame0(seed, H, W_)
    frames = torch.stack([torch.roll(f0, shifts=(2 * t, 5 * t), dims=(1, 2)) for t in range(T)], 0)
    return frames.contiguous(), m0


def clip_on_device(seed, T, H, W_, device):
    """``clip`` built in device memory (frame 0 is synthesised on the host, the rolls run on the GPU): a long
    high-resolution stream -- 2000 x 1080p = 50 GB -- never exists in host memory."""
    f0, m0 = frame0(seed, H, W_)
    f0 = f0.to(device)
    frames = torch.empty(T, 3, H, W_, device=device)
    for t in range(T):
        frames[t] = torch.roll(f0, shifts=(2 * t, 5 * t), dims=(1, 2))
    return frames, m0


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


DEFAULT_KNOBS = dict(key_scale=1.0, mask_scale=1.0, res_scale=0.4, dec_res_scale=0.5, out_scale=1.0,
                     logit_std=1.0, local_std=0.25)


def random_state_dict(template_sd, seed, knobs=None):
    kn = dict(DEFAULT_KNOBS)
    kn.update(knobs or {})
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
    for name in sd:
        if name.startswith('keyval_r4.Key.'):
            sd[name] = sd[name] * kn['key_scale']
        elif name in ('encoder_m.conv1_m.weight', 'encoder_m.conv1_o.weight'):
            sd[name] = sd[name] * kn['mask_scale']
        elif name.endswith('bn3.weight'):
            sd[name] = sd[name] * kn['res_scale']
        elif name.startswith('decoder.') and ('.ResMM.conv2.' in name or '.ResFS.conv2.' in name):
            sd[name] = sd[name] * kn['dec_res_scale']
        elif name in ('decoder.pred2.weight', 'decoder.local_pred2.weight'):
            sd[name] = sd[name] * kn['out_scale']
    return sd


def _bn_apply(x, sd, prefix, calibrate):
    """BatchNorm: batch statistics (recorded as the running stats) when calibrating, eval otherwise."""
    if calibrate:
        mean = x.mean(dim=(0, 2, 3))
        var_b = x.var(dim=(0, 2, 3), unbiased=False)
        n = x.numel() / x.shape[1]
        sd[prefix + '.running_mean'] = mean.clone()
        sd[prefix + '.running_var'] = (var_b * n / max(n - 1, 1)).clone()   # running_var is unbiased
        y = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var_b.view(1, -1, 1, 1) + W.BN_EPS)
        return y * sd[prefix + '.weight'].view(1, -1, 1, 1) + sd[prefix + '.bias'].view(1, -1, 1, 1)
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                        sd[prefix + '.weight'], sd[prefix + '.bias'], False, 0.0, W.BN_EPS)


def _layer_fwd(x, sd, prefix, blocks, stride, calibrate):
    for b in range(blocks):
        p = f'{prefix}.{b}'
        s = stride if b == 0 else 1
        out = F.relu(_bn_apply(F.conv2d(x, sd[p + '.conv1.weight']), sd, p + '.bn1', calibrate))
        out = F.relu(_bn_apply(F.conv2d(out, sd[p + '.conv2.weight'], stride=s, padding=1), sd, p + '.bn2', calibrate))
        out = _bn_apply(F.conv2d(out, sd[p + '.conv3.weight']), sd, p + '.bn3', calibrate)
        if b == 0:
            idn = _bn_apply(F.conv2d(x, sd[p + '.downsample.0.weight'], stride=s), sd, p + '.downsample.1', calibrate)
        else:
            idn = x
        x = F.relu(out + idn)
    return x


def _trunk_fwd(x, sd, prefix, calibrate):
    r1 = F.relu(_bn_apply(x, sd, prefix + '.bn1', calibrate))
    x = F.max_pool2d(r1, 3, 2, 1)
    r2 = _layer_fwd(x, sd, prefix + '.res2', 3, 1, calibrate)
    r3 = _layer_fwd(r2, sd, prefix + '.res3', 4, 2, calibrate)
    r4 = _layer_fwd(r3, sd, prefix + '.res4', 6, 2, calibrate)
    return r4, r3, r2, r1


def _enc_q(sd, frame, calibrate):
    f = (frame - sd['encoder_q.mean']) / sd['encoder_q.std']
    return _trunk_fwd(F.conv2d(f, sd['encoder_q.conv1.weight'], stride=2, padding=3), sd, 'encoder_q', calibrate)


def _enc_m(sd, frame, masks, calibrate):
    """frame [1,3,h,w]; masks [K,h,w] float."""
    f = ((frame - sd['encoder_m.mean']) / sd['encoder_m.std']).expand(masks.shape[0], -1, -1, -1)
    m = masks.unsqueeze(1)
    o = (1 - m).clamp(0, 1)
    x = F.conv2d(f, sd['encoder_m.conv1.weight'], stride=2, padding=3) \
        + F.conv2d(m, sd['encoder_m.conv1_m.weight'], stride=2, padding=3) \
        + F.conv2d(o, sd['encoder_m.conv1_o.weight'], stride=2, padding=3)
    return _trunk_fwd(x, sd, 'encoder_m', calibrate)[0]


def _c3(sd, p, x):
    return F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], padding=1)


def _rb(sd, p, x):
    return x + _c3(sd, p + '.conv2', F.relu(_c3(sd, p + '.conv1', F.relu(x))))


def _calibrate(sd, seed, kn):
    """Data-dependent part of the recipe, on one synthetic frame pair (plain torch CPU ops):
    BatchNorm running statistics of both encoders, then the two prediction heads are rescaled and
    re-centred so the logits neither saturate at the clamp (AFB_URR.py:309) nor collapse to ties."""
    H, W_ = CALIB_HW
    f0, m0 = frame0(seed, H, W_)
    f1 = torch.roll(f0, shifts=(2, 5), dims=(1, 2))
    oh = onehot(m0).float()
    from vfloodnet_amd.engine import pad_divide_by
    pad, _, _ = pad_divide_by(H, W_)                 # myutils/data.py:132-149: zero pad *before* normalisation
    f0, f1, oh = F.pad(f0, pad), F.pad(f1, pad), F.pad(oh, pad)
    r4, r3, r2, r1 = _enc_q(sd, f1.unsqueeze(0), True)
    r4m = _enc_m(sd, f0.unsqueeze(0), oh, True)
    # memory read of frame 1 against the bank of frame 0 (AFB_URR.py:136-178)
    kq = _c3(sd, 'keyval_r4.Key', r4).flatten(2)[0]
    vq = _c3(sd, 'keyval_r4.Value', r4).flatten(2)[0]
    km = _c3(sd, 'keyval_r4.Key', r4m).flatten(2)
    vm = _c3(sd, 'keyval_r4.Value', r4m).flatten(2)
    gh, gw = r4.shape[2:]
    res = []
    for i in range(2):
        p = torch.softmax(km[i].t() @ kq / math.sqrt(km.shape[1]), dim=0)
        res.append(torch.cat([vm[i] @ p, vq], 0))
    x = torch.stack(res, 0).view(2, -1, gh, gw)
    D = 'decoder'
    p = _rb(sd, D + '.ResMM', _c3(sd, D + '.convFM', x))
    for rf, feat in (('.RF3', r3), ('.RF2', r2)):
        s_ = _rb(sd, D + rf + '.ResFS', _c3(sd, D + rf + '.convFS', feat))
        p = _rb(sd, D + rf + '.ResMM', s_ + F.interpolate(p, scale_factor=2, mode='bilinear', align_corners=False))
    feat = F.relu(p)
    c = _c3(sd, D + '.pred2', feat)
    diff = c[:, 1] - c[:, 0]
    g = kn['logit_std'] / max(float(diff.std()), 1e-6)
    sd[D + '.pred2.weight'] = sd[D + '.pred2.weight'] * g
    sd[D + '.pred2.bias'] = sd[D + '.pred2.bias'] * g
    sd[D + '.pred2.bias'][1] -= float((diff * g).median())
    # local head (AFB_URR.py:226-235)
    c = F.interpolate(_c3(sd, D + '.pred2', feat), scale_factor=2, mode='bilinear', align_corners=False)
    rough = torch.softmax(torch.softmax(c, dim=1)[:, 1].unsqueeze(0), dim=1)[0].unsqueeze(1)
    r1e = r1.expand(2, -1, -1, -1)
    r1_local = F.avg_pool2d(r1e * rough, 7, 1, 3) / (F.avg_pool2d(rough, 7, 1, 3) + 1e-8)
    q = _rb(sd, D + '.local_ResMM', _c3(sd, D + '.local_convFM', torch.cat([r1e, r1_local], 1)))
    qf = F.relu(q)
    c2 = _c3(sd, D + '.local_pred2', qf)
    d2 = c2[:, 1] - c2[:, 0]
    g2 = kn['local_std'] / max(float(d2.std()), 1e-6)
    sd[D + '.local_pred2.weight'] = sd[D + '.local_pred2.weight'] * g2
    sd[D + '.local_pred2.bias'] = sd[D + '.local_pred2.bias'] * g2
    sd[D + '.local_pred2.bias'][1] -= float((d2 * g2).median())


def make_state_dict(seed=20200212, **knobs):
    """Deterministic calibrated state dict (CPU tensors, reference key names)."""
    from vfloodnet_amd.model import AFB_URR
    with torch.no_grad():
        tmpl = AFB_URR(torch.device('cpu'), update_bank=True, _allow_cpu_container=True).state_dict()
        sd = random_state_dict(tmpl, seed, knobs)
        nthr = torch.get_num_threads()
        torch.set_num_threads(1)           # fixed reduction order -> bit-reproducible statistics
        try:
            kn = dict(DEFAULT_KNOBS)
            kn.update(knobs)
            _calibrate(sd, seed, kn)
        finally:
            torch.set_num_threads(nthr)
    return sd


def make_checkpoint(path, seed=20200212):
    """Write a reference-schema checkpoint (train_video_seg.py:159-177)."""
    sd = make_state_dict(seed)
    torch.save({'epoch': 0, 'model': sd, 'loss': 0.0, 'seed': int(seed)}, path)
    return path
