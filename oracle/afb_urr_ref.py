"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement (torch CPU fp32 ops) of the reference's video-segmentation hot
path, written as flat functions over a reference-named state dict.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package.

Parity pin: ``oracle/gen_golden.py`` runs the *reference itself* (imported from
/root/reference under ``oracle/refstubs.py``) on seeded inputs and commits the
outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every
function here against those vectors.  The third-party pieces the reference
reaches (torchvision resnet50 / TF.resize, torch_scatter, cv2 CCL) are absent
from /root/reference and are restated from their published semantics
(``refstubs.py`` header) -- for those boundaries parity is pinned to the stub's
documented semantics, not to the original binaries.

Each function cites the reference lines it follows (paths relative to
/root/reference).
"""
import math

import numpy as np
import torch
from torch.nn import functional as F

BN_EPS = 1e-5


# ----------------------------------------------------------------- helpers
def pad_divide_by(in_list, d, in_size):
    """myutils/data.py:132-149 -- symmetric zero pad to a multiple of d, extra pixel bottom/right."""
    h, w = in_size
    new_h = h + d - h % d if h % d > 0 else h
    new_w = w + d - w % d if w % d > 0 else w
    lh = int((new_h - h) / 2)
    uh = int(new_h - h) - lh
    lw = int((new_w - w) / 2)
    uw = int(new_w - w) - lw
    pad = (lw, uw, lh, uh)
    return [F.pad(x, pad) for x in in_list], pad


def calc_uncertainty(score):
    """myutils/data.py:40-46."""
    top, _ = score.topk(k=2, dim=1)
    u = top[:, 0] / (top[:, 1] + 1e-8)
    return torch.exp(1 - u).unsqueeze(1)


def _bn(x, sd, p):
    """BatchNorm2d eval (AFB_URR.py:56,86 and the torchvision bottlenecks)."""
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], False, 0.0, BN_EPS)


def _bottleneck(x, sd, p, stride, has_down):
    """torchvision ResNet v1.5 Bottleneck: stride on the 3x3."""
    out = F.relu(_bn(F.conv2d(x, sd[p + '.conv1.weight']), sd, p + '.bn1'))
    out = F.relu(_bn(F.conv2d(out, sd[p + '.conv2.weight'], stride=stride, padding=1), sd, p + '.bn2'))
    out = _bn(F.conv2d(out, sd[p + '.conv3.weight']), sd, p + '.bn3')
    if has_down:
        x = _bn(F.conv2d(x, sd[p + '.downsample.0.weight'], stride=stride), sd, p + '.downsample.1')
    return F.relu(out + x)


def _layer(x, sd, p, blocks, stride):
    for b in range(blocks):
        x = _bottleneck(x, sd, f'{p}.{b}', stride if b == 0 else 1, b == 0)
    return x


def _trunk_tail(x, sd, p):
    r1 = F.relu(_bn(x, sd, p + '.bn1'))
    x = F.max_pool2d(r1, 3, 2, 1)
    r2 = _layer(x, sd, p + '.res2', 3, 1)
    r3 = _layer(r2, sd, p + '.res3', 4, 2)
    r4 = _layer(r3, sd, p + '.res4', 6, 2)
    return r4, r3, r2, r1


def encoder_q(sd, in_f):
    """AFB_URR.py:82-93."""
    f = (in_f - sd['encoder_q.mean']) / sd['encoder_q.std']
    x = F.conv2d(f, sd['encoder_q.conv1.weight'], stride=2, padding=3)
    return _trunk_tail(x, sd, 'encoder_q')


def encoder_m(sd, in_f, in_m, in_o):
    """AFB_URR.py:52-63."""
    f = (in_f - sd['encoder_m.mean']) / sd['encoder_m.std']
    x = F.conv2d(f, sd['encoder_m.conv1.weight'], stride=2, padding=3) \
        + F.conv2d(in_m, sd['encoder_m.conv1_m.weight'], stride=2, padding=3) \
        + F.conv2d(in_o, sd['encoder_m.conv1_o.weight'], stride=2, padding=3)
    r4, _, _, r1 = _trunk_tail(x, sd, 'encoder_m')
    return r4, r1


def keyval(sd, x):
    """AFB_URR.py:105-111."""
    k = F.conv2d(x, sd['keyval_r4.Key.weight'], sd['keyval_r4.Key.bias'], padding=1)
    v = F.conv2d(x, sd['keyval_r4.Value.weight'], sd['keyval_r4.Value.bias'], padding=1)
    return k.view(*k.shape[:2], -1), v.view(*v.shape[:2], -1)


def _conv3(sd, p, x):
    return F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], padding=1)


def _resblock(sd, p, x):
    """AFB_URR.py:23-30 (pre-activation, un-activated skip; indim == outdim here)."""
    r = _conv3(sd, p + '.conv1', F.relu(x))
    r = _conv3(sd, p + '.conv2', F.relu(r))
    return x + r


def _refine(sd, p, f, pm):
    """AFB_URR.py:122-127."""
    s = _resblock(sd, p + '.ResFS', _conv3(sd, p + '.convFS', f))
    m = s + F.interpolate(pm, scale_factor=2, mode='bilinear', align_corners=False)
    return _resblock(sd, p + '.ResMM', m)


def matcher(fb, q_in, q_out, update_bank=True, thres_valid=1e-3):
    """AFB_URR.py:136-178 (memory read + hit-count side effect on fb.info)."""
    outs = []
    for i in range(fb.obj_n):
        d_key, bank_n = fb.keys[i].shape
        p = torch.matmul(fb.keys[i].transpose(0, 1), q_in) / math.sqrt(d_key)
        p = F.softmax(p, dim=1)
        mem = torch.matmul(fb.values[i], p)
        outs.append(torch.cat([mem, q_out], dim=1))
        if update_bank:
            cnt = torch.where(p > thres_valid, torch.ones_like(p), torch.zeros_like(p)).sum(dim=2)[0]
            fb.info[i][:, 1] += torch.log(cnt + 1)
    return torch.stack(outs, dim=0).transpose(0, 1)


def decoder_global(sd, patch_match, r3, r2):
    """AFB_URR.py:209-212: the decoder's global branch, p = pred2(relu(RF2(r2, RF3(r3, ResMM(convFM(patch_match)))))), before
    the interpolation and the local refinement (differentiated by tests/test_backward_gpu.py with torch.autograd)."""
    D = 'decoder'
    p = _resblock(sd, D + '.ResMM', _conv3(sd, D + '.convFM', patch_match))
    p = _refine(sd, D + '.RF3', r3, p)
    p = _refine(sd, D + '.RF2', r2, p)
    return _conv3(sd, D + '.pred2', F.relu(p))


def decoder(sd, patch_match, r3, r2, r1, feature_shape, return_parts=False):
    """AFB_URR.py:208-239."""
    D = 'decoder'
    p = decoder_global(sd, patch_match, r3, r2)
    p = F.interpolate(p, scale_factor=2, mode='bilinear', align_corners=False)

    bs, obj_n, h, w = feature_shape
    rough = F.softmax(p, dim=1)[:, 1].view(bs, obj_n, h, w)
    rough = F.softmax(rough, dim=1)
    unc = calc_uncertainty(rough)
    unc = unc.expand(-1, obj_n, -1, -1).reshape(bs * obj_n, 1, h, w)
    rough = rough.view(bs * obj_n, 1, h, w)
    r1_w = r1 * rough
    r1_local = F.avg_pool2d(r1_w, 7, 1, 3)
    r1_local = r1_local / (F.avg_pool2d(rough, 7, 1, 3) + 1e-8)
    r1_conf = F.max_pool2d(rough, 7, 1, 3)
    lm = torch.cat([r1, r1_local], dim=1)
    q = _resblock(sd, D + '.local_ResMM', _conv3(sd, D + '.local_convFM', lm))
    q = r1_conf * _conv3(sd, D + '.local_pred2', F.relu(q))
    p2 = p + unc * q
    out = F.interpolate(p2, scale_factor=2, mode='bilinear', align_corners=False)
    out = F.softmax(out, dim=1)[:, 1]
    if return_parts:
        return out, dict(p_up=p, rough=rough, unc=unc, r1_local=r1_local, r1_conf=r1_conf, q=q)
    return out


def memorize(sd, frame, mask):
    """AFB_URR.py:255-272."""
    _, K, H, W = mask.shape
    (frame, mask), pad = pad_divide_by([frame, mask], 16, (frame.shape[2], frame.shape[3]))
    frame = frame.expand(K, -1, -1, -1)
    mask = mask[0].unsqueeze(1).to(torch.float64 if frame.dtype == torch.float64 else torch.float32)  # .float()
    mask_inv = (torch.ones_like(mask) - mask).clamp(0, 1)
    r4, _ = encoder_m(sd, frame, mask, mask_inv)
    k4, v4 = keyval(sd, r4)
    return [k4[i] for i in range(K)], [v4[i] for i in range(K)]


def segment(sd, frame, fb, update_bank=True, training=False):
    """AFB_URR.py:274-318.  ``frame`` f32[bs,3,h,w] (the loop uses bs = 1, training bs = clip_n - 1,
    train_video_seg.py:69).  ``training``: the branch ``model.train()`` takes with BatchNorm frozen
    (train_video_seg.py:103-106): no padding (:278) and the scalar uncertainty of :302-305."""
    obj_n = fb.obj_n
    pad = (0, 0, 0, 0)
    if not training:
        [frame], pad = pad_divide_by([frame], 16, (frame.shape[2], frame.shape[3]))
    r4, r3, r2, r1 = encoder_q(sd, frame)
    bs, _, gh, gw = r4.shape
    k4, v4 = keyval(sd, r4)
    res = matcher(fb, k4, v4, update_bank)
    res = res.reshape(bs * obj_n, v4.shape[1] * 2, gh, gw)
    r3e = r3.unsqueeze(1).expand(-1, obj_n, -1, -1, -1).reshape(bs * obj_n, *r3.shape[1:])
    r2e = r2.unsqueeze(1).expand(-1, obj_n, -1, -1, -1).reshape(bs * obj_n, *r2.shape[1:])
    r1e = r1.unsqueeze(1).expand(-1, obj_n, -1, -1, -1).reshape(bs * obj_n, *r1.shape[1:])
    score = decoder(sd, res, r3e, r2e, r1e, (bs, obj_n, r1.shape[2], r1.shape[3]))
    score = score.view(bs, obj_n, *frame.shape[-2:])
    uncertainty = None
    if training:
        uncertainty = calc_uncertainty(F.softmax(score, dim=1))
        uncertainty = uncertainty.view(bs, -1).norm(p=2, dim=1) / math.sqrt(frame.shape[-2] * frame.shape[-1])
        uncertainty = uncertainty.mean()
    score = torch.clamp(score, 1e-7, 1 - 1e-7)
    score = torch.log(score / (1 - score))
    if not training:
        if pad[2] + pad[3] > 0:
            score = score[:, :, pad[2]:-pad[3], :]
        if pad[0] + pad[1] > 0:
            score = score[:, :, :, pad[0]:-pad[1]]
    return score, uncertainty


# ----------------------------------------------------------------- feature bank
def scatter_mean(src, index, dim, out):
    """torch_scatter 2.0.8 ``scatter_mean`` with ``out=`` (FeatureBank.py:78,92)."""
    out.scatter_add_(dim, index, src)
    count = torch.zeros_like(out)
    count.scatter_add_(dim, index, torch.ones_like(src))
    count.clamp_(min=1)
    out.true_divide_(count)
    return out


class FeatureBankRef:
    """FeatureBank.py:8-149."""

    def __init__(self, obj_n, memory_budget, device='cpu', update_rate=0.1, thres_close=0.95):
        self.obj_n = obj_n
        self.update_rate = update_rate
        self.thres_close = thres_close
        self.device = device
        self.info = [None for _ in range(obj_n)]
        self.peak_n = np.zeros(obj_n)
        self.replace_n = np.zeros(obj_n)
        self.class_budget = memory_budget // obj_n
        if obj_n == 2:
            self.class_budget = 0.8 * self.class_budget
        self.keys = None
        self.values = None

    def init_bank(self, keys, values, frame_idx=0):
        self.keys = keys
        self.values = values
        for c in range(self.obj_n):
            n = keys[c].shape[1]
            self.info[c] = torch.zeros((n, 2))
            self.info[c][:, 0] = frame_idx
            self.peak_n[c] = max(self.peak_n[c], n)

    def append(self, keys, values, frame_idx=0):
        if self.keys:
            for c in range(self.obj_n):
                self.keys[c] = torch.cat([self.keys[c], keys[c]], dim=1)
                self.values[c] = torch.cat([self.values[c], values[c]], dim=1)
                n = keys[c].shape[1]
                ni = torch.ones((n, 2)) * 20
                ni[:, 0] = frame_idx
                self.info[c] = torch.cat([self.info[c], ni], dim=0)
                self.peak_n[c] = max(self.peak_n[c], self.info[c].shape[0])
        else:
            self.init_bank(keys, values, frame_idx)

    def update(self, prev_key, prev_value, frame_idx, update_rate=-1):
        if update_rate == -1:
            update_rate = self.update_rate
        for c in range(self.obj_n):
            d_key, bank_n = self.keys[c].shape
            d_val = self.values[c].shape[0]
            nk = F.normalize(self.keys[c], dim=0)
            npk = F.normalize(prev_key[c], dim=0)
            mag_k = self.keys[c].norm(p=2, dim=0)
            corr = torch.mm(nk.transpose(0, 1), npk)
            rel_idx = corr.argmax(dim=0, keepdim=True)
            rel_corr = torch.gather(corr, 0, rel_idx)

            sel = (rel_corr[0] > self.thres_close).nonzero(as_tuple=False)
            cls_idx = rel_idx[0, sel[:, 0]]
            uniq, _ = cls_idx.unique(dim=0, return_counts=True)

            kupd = torch.zeros((d_key, bank_n), dtype=self.keys[c].dtype)   # dtype=torch.float in the reference
            scatter_mean(npk[:, sel[:, 0]], cls_idx.unsqueeze(0).expand(d_key, -1), 1, kupd)
            self.keys[c][:, uniq] = mag_k[uniq] * ((1 - update_rate) * nk[:, uniq] + update_rate * kupd[:, uniq])

            nv = F.normalize(self.values[c], dim=0)
            npv = F.normalize(prev_value[c], dim=0)
            mag_v = self.values[c].norm(p=2, dim=0)
            vupd = torch.zeros((d_val, bank_n), dtype=self.values[c].dtype)
            scatter_mean(npv[:, sel[:, 0]], cls_idx.unsqueeze(0).expand(d_val, -1), 1, vupd)
            self.values[c][:, uniq] = mag_v[uniq] * ((1 - update_rate) * nv[:, uniq] + update_rate * vupd[:, uniq])

            sel = (rel_corr[0] <= self.thres_close).nonzero(as_tuple=False)
            if self.class_budget < bank_n + sel.shape[0]:
                self.remove(c, sel.shape[0], frame_idx)
            self.keys[c] = torch.cat([self.keys[c], prev_key[c][:, sel[:, 0]]], dim=1)
            self.values[c] = torch.cat([self.values[c], prev_value[c][:, sel[:, 0]]], dim=1)
            ni = torch.zeros((sel.shape[0], 2))
            ni[:, 0] = frame_idx
            self.info[c] = torch.cat([self.info[c], ni], dim=0)
            self.peak_n[c] = max(self.peak_n[c], self.info[c].shape[0])
            self.info[c][:, 1] = torch.clamp(self.info[c][:, 1], 0, 1e5)

    def remove(self, c, request_n, frame_idx):
        old = self.keys[c].shape[1]
        lfu = frame_idx - self.info[c][:, 0]
        lfu = self.info[c][:, 1] / lfu
        thr = int(lfu.min()) + 1
        while True:
            keep = lfu > thr
            self.keys[c] = self.keys[c][:, keep]
            self.values[c] = self.values[c][:, keep]
            self.info[c] = self.info[c][keep]
            lfu = lfu[keep]
            balance = (self.class_budget - self.keys[c].shape[1]) - request_n
            if balance < 0:
                thr = int(lfu.min()) + 1
            else:
                break
        self.replace_n[c] += old - self.keys[c].shape[1]
        return balance


# ----------------------------------------------------------------- loop pieces
def tf_resize(img, size, mode):
    """torchvision 0.9.2 tensor ``TF.resize(img, int size, mode)`` (test_video_seg.py:88-89,107,114)."""
    h, w = img.shape[-2:]
    if isinstance(size, int):
        short, long = (w, h) if w <= h else (h, w)
        if short == size:
            return img
        new_short, new_long = size, int(size * long / short)
        new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    else:
        new_h, new_w = size
    odt = img.dtype
    x = img if odt.is_floating_point else img.float()
    if mode == 'nearest':
        x = F.interpolate(x, size=[new_h, new_w], mode='nearest')
    else:
        x = F.interpolate(x, size=[new_h, new_w], mode=mode, align_corners=False)
        if odt == torch.uint8:
            x = x.clamp(0, 255)
    if not odt.is_floating_point:
        x = torch.round(x).to(odt)
    return x


def postprocessing_pred(pred):
    """myutils/data.py:17-37: keep the largest 8-connected water blob (all-background -> all-ones quirk)."""
    from scipy import ndimage
    labels, n = ndimage.label(pred != 0, structure=np.ones((3, 3), np.int32))
    label_cnt = n + 1
    if label_cnt == 2:
        out = labels if labels[0, 0] == pred[0, 0] else 1 - labels
    else:
        max_cnt, max_label = 0, 0
        for i in range(label_cnt):
            m = labels == i
            if pred[m][0] == 0:
                continue
            cnt = int(m.sum())
            if cnt > max_cnt:
                max_cnt, max_label = cnt, i
        out = labels == max_label
    return out.astype(np.uint8)


def run_clip(sd, frames, first_mask_u8, budget=250000, update_rate=0.1, thres_close=0.95,
             size=480, mem_every=1, return_scores=False):
    """test_video_seg.py:83-121 without file I/O: frames f32[T,3,H0,W0], first mask u8[H0,W0] (>0 = water).

    Returns labels u8[T,H0,W0] *before* post-processing (frame 0 = the given mask), the
    per-frame bank sizes and the bank.  ``mem_every`` > 1 is the harness option of BASELINE
    config C3 (the reference memorises every frame).
    """
    T = frames.shape[0]
    H0, W0 = frames.shape[-2:]
    m = (first_mask_u8 > 0).to(torch.uint8)
    onehot = torch.stack([1 - m, m], 0).unsqueeze(0)                       # Water_DS.py:93-101
    fb = FeatureBankRef(2, budget, 'cpu', update_rate, thres_close)
    f0 = tf_resize(frames[0:1], size, 'bicubic')
    m0 = tf_resize(onehot, size, 'nearest')
    labels = [onehot[0].argmax(0).to(torch.uint8)]
    sizes = []
    scores = []
    k, v = memorize(sd, f0, m0)
    fb.init_bank(k, v)
    for t in range(1, T):
        fr = tf_resize(frames[t:t + 1], size, 'bicubic')
        score, _ = segment(sd, fr, fb)
        pm = F.softmax(score, dim=1)
        if t % mem_every == 0:
            k, v = memorize(sd, fr, pm)
            fb.update(k, v, t)
        pred = tf_resize(pm, [H0, W0], 'bicubic')
        labels.append(pred[0].argmax(0).to(torch.uint8))
        sizes.append([int(fb.keys[c].shape[1]) for c in range(2)])
        if return_scores:
            scores.append(score)
    out = dict(labels=torch.stack(labels, 0), bank_sizes=sizes, fb=fb)
    if return_scores:
        out['scores'] = scores
    return out
