"""CPU restatement of the first-frame bootstrap model: ``smp.Linknet(encoder_name='efficientnet-b4', classes=1,
activation='sigmoid')`` as the reference trains it (train_image_seg.py:82-89) and runs it (test_image_seg.py:95-124,133;
called from test_video_seg.py:67-69 when a clip has no first-frame mask).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke, bench.py's cpu_baseline): the product path never imports this.

**PARITY UNPINNED.**  The arithmetic lives in two third-party packages that are NOT under /root/reference and not installed
here: ``segmentation_models_pytorch==0.2.0`` (requirements.txt:9), which pins ``efficientnet-pytorch==0.6.3``; the trained
weights (``records/link_efficientb4_model.pth``, a pickled module, test_video_seg.py:68) are absent as well, and the
reference holds no test or golden output for this model.  What follows restates the two packages' PUBLISHED architecture:

* EfficientNet-B4 (width 1.4, depth 1.8; Tan & Le 2019; ``efficientnet_pytorch/model.py``, ``utils.py`` of 0.6.3):
  stem conv 3x3/2 -> 48, BN(eps 1e-3), swish; 32 MBConv blocks in 7 stages
  (repeats, kernel, stride, expand, in, out) = (2,3,1,1,48,24) (4,3,2,6,24,32) (4,5,2,6,32,56) (6,3,2,6,56,112)
  (6,5,1,6,112,160) (8,5,2,6,160,272) (2,3,1,6,272,448); a block = [1x1 expand, BN, swish] (expand != 1), depthwise kxk, BN,
  swish, squeeze-excite (mean -> 1x1 to max(1, int(0.25 * block input filters)) -> swish -> 1x1 -> sigmoid gate), 1x1 project,
  BN, + input when stride 1 and in == out.  "Static same padding" of 0.6.3: every convolution pads as TensorFlow's SAME would for
  the model's NATIVE 380-pixel input (``image_size=global_params.image_size`` for every layer): symmetric (k-1)/2 for stride
  1; for stride 2: k = 3 -> (0 before, 1 after), k = 5 -> (1 before, 2 after).
  smp's encoder (``encoders/efficientnet.py``) returns [x, stem, block 6, block 10, block 22, block 32] outputs (channels
  3, 48, 32, 56, 160, 448); ``_conv_head`` / ``_bn1`` stay in the state dict and are not used.
* LinkNet decoder (``linknet/decoder.py``): five blocks (448->160, 160->56, 56->32, 32->48, 48->32), each
  Conv1x1(in -> in/4)+BN+ReLU, ConvTranspose2d(in/4 -> in/4, k 4, s 2, p 1, bias)+BN+ReLU, Conv1x1(in/4 -> out)+BN+ReLU, then
  ``+ skip`` (the encoder feature of that resolution; none for the last block); head Conv1x1(32 -> 1, bias) + sigmoid.

State-dict names are the two packages' (``encoder._blocks.3._depthwise_conv.weight``, ``decoder.blocks.0.block.1.0.weight``,
``segmentation_head.0.bias`` ...), so that ``torch.load(path).state_dict()`` of the reference's pickled model would load.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS_ENC = 1e-3          # efficientnet_pytorch: batch_norm_epsilon
BN_EPS_DEC = 1e-5          # nn.BatchNorm2d default (smp modules.Conv2dReLU / TransposeX2)

# (repeats, kernel, stride, expand, in, out) after round_filters(width 1.4) / round_repeats(depth 1.8)
STAGES = ((2, 3, 1, 1, 48, 24), (4, 3, 2, 6, 24, 32), (4, 5, 2, 6, 32, 56), (6, 3, 2, 6, 56, 112),
          (6, 5, 1, 6, 112, 160), (8, 5, 2, 6, 160, 272), (2, 3, 1, 6, 272, 448))
STAGE_IDXS = (6, 10, 22, 32)                     # smp: features after these many blocks
ENC_CHANNELS = (3, 48, 32, 56, 160, 448)
DEC_CHANNELS = (448, 160, 56, 32, 48, 32)        # encoder channels reversed (first skip dropped) + prefinal 32


def blocks():
    """-> list of dicts(k, s, e, cin, cout, sq) for the 32 MBConv blocks."""
    out = []
    for rep, k, s, e, cin, cout in STAGES:
        for r in range(rep):
            ci = cin if r == 0 else cout
            out.append(dict(k=k, s=s if r == 0 else 1, e=e, cin=ci, cout=cout, sq=max(1, int(ci * 0.25))))
    return out


def same_pad(k, s):
    """(before, after) of the 0.6.3 static same padding (computed for the native 380-pixel input at every layer)."""
    ih = 380
    oh = math.ceil(ih / s)
    p = max((oh - 1) * s + (k - 1) + 1 - ih, 0)
    return p // 2, p - p // 2


def template():
    """name -> shape of every entry of the model's state dict (buffers included)."""
    t = {}

    def bn(prefix, c):
        t[prefix + '.weight'] = (c,); t[prefix + '.bias'] = (c,)
        t[prefix + '.running_mean'] = (c,); t[prefix + '.running_var'] = (c,); t[prefix + '.num_batches_tracked'] = ()
    t['encoder._conv_stem.weight'] = (48, 3, 3, 3)
    bn('encoder._bn0', 48)
    for i, b in enumerate(blocks()):
        p = f'encoder._blocks.{i}'
        oup = b['cin'] * b['e']
        if b['e'] != 1:
            t[p + '._expand_conv.weight'] = (oup, b['cin'], 1, 1)
            bn(p + '._bn0', oup)
        t[p + '._depthwise_conv.weight'] = (oup, 1, b['k'], b['k'])
        bn(p + '._bn1', oup)
        t[p + '._se_reduce.weight'] = (b['sq'], oup, 1, 1); t[p + '._se_reduce.bias'] = (b['sq'],)
        t[p + '._se_expand.weight'] = (oup, b['sq'], 1, 1); t[p + '._se_expand.bias'] = (oup,)
        t[p + '._project_conv.weight'] = (b['cout'], oup, 1, 1)
        bn(p + '._bn2', b['cout'])
    t['encoder._conv_head.weight'] = (1792, 448, 1, 1)       # kept by smp, unused by the encoder's forward
    bn('encoder._bn1', 1792)
    for j in range(5):
        cin, cout = DEC_CHANNELS[j], DEC_CHANNELS[j + 1]
        p = f'decoder.blocks.{j}.block'
        t[p + '.0.0.weight'] = (cin // 4, cin, 1, 1); bn(p + '.0.1', cin // 4)
        t[p + '.1.0.weight'] = (cin // 4, cin // 4, 4, 4); t[p + '.1.0.bias'] = (cin // 4,); bn(p + '.1.1', cin // 4)
        t[p + '.2.0.weight'] = (cout, cin // 4, 1, 1); bn(p + '.2.1', cout)
    t['segmentation_head.0.weight'] = (1, 32, 1, 1)
    t['segmentation_head.0.bias'] = (1,)
    return t


def _swish(x):
    return x * torch.sigmoid(x)


def _bn(x, sd, prefix, eps, calibrate):
    """eval-mode BatchNorm; ``calibrate``: take the statistics from this batch and record them as the running ones (synthetic
    weights only: keeps 32 random blocks at unit scale)."""
    if calibrate:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        sd[prefix + '.running_mean'] = mean.detach().clone()
        sd[prefix + '.running_var'] = var.detach().clone()
    return F.batch_norm(x, sd[prefix + '.running_mean'].to(x.dtype), sd[prefix + '.running_var'].to(x.dtype),
                        sd[prefix + '.weight'].to(x.dtype), sd[prefix + '.bias'].to(x.dtype), False, 0.0, eps)


def _conv_same(x, w, k, s, groups=1):
    b, a = same_pad(k, s)
    if b or a:
        x = F.pad(x, (b, a, b, a))
    return F.conv2d(x, w.to(x.dtype), stride=s, groups=groups)


def encoder(sd, x, calibrate=False):
    """x [N,3,H,W] (ImageNet-normalised) -> the six features smp's encoder returns."""
    feats = [x]
    x = _swish(_bn(_conv_same(x, sd['encoder._conv_stem.weight'], 3, 2), sd, 'encoder._bn0', BN_EPS_ENC, calibrate))
    feats.append(x)
    for i, b in enumerate(blocks()):
        p = f'encoder._blocks.{i}'
        inp = x
        oup = b['cin'] * b['e']
        if b['e'] != 1:
            x = _swish(_bn(F.conv2d(x, sd[p + '._expand_conv.weight'].to(x.dtype)), sd, p + '._bn0', BN_EPS_ENC, calibrate))
        x = _swish(_bn(_conv_same(x, sd[p + '._depthwise_conv.weight'], b['k'], b['s'], groups=oup), sd, p + '._bn1', BN_EPS_ENC,
                       calibrate))
        g = F.adaptive_avg_pool2d(x, 1)
        g = _swish(F.conv2d(g, sd[p + '._se_reduce.weight'].to(x.dtype), sd[p + '._se_reduce.bias'].to(x.dtype)))
        g = F.conv2d(g, sd[p + '._se_expand.weight'].to(x.dtype), sd[p + '._se_expand.bias'].to(x.dtype))
        x = torch.sigmoid(g) * x
        x = _bn(F.conv2d(x, sd[p + '._project_conv.weight'].to(x.dtype)), sd, p + '._bn2', BN_EPS_ENC, calibrate)
        if b['s'] == 1 and b['cin'] == b['cout']:
            x = x + inp                                            # (drop_connect is a training-time operation)
        if i + 1 in STAGE_IDXS:
            feats.append(x)
    return feats


def decoder(sd, feats, calibrate=False):
    feats = feats[1:][::-1]                                        # drop the input, deepest first
    x, skips = feats[0], feats[1:]
    for j in range(5):
        p = f'decoder.blocks.{j}.block'
        x = F.relu(_bn(F.conv2d(x, sd[p + '.0.0.weight'].to(x.dtype)), sd, p + '.0.1', BN_EPS_DEC, calibrate))
        x = F.conv_transpose2d(x, sd[p + '.1.0.weight'].to(x.dtype), sd[p + '.1.0.bias'].to(x.dtype), stride=2, padding=1)
        x = F.relu(_bn(x, sd, p + '.1.1', BN_EPS_DEC, calibrate))
        x = F.relu(_bn(F.conv2d(x, sd[p + '.2.0.weight'].to(x.dtype)), sd, p + '.2.1', BN_EPS_DEC, calibrate))
        if j < len(skips):
            x = x + skips[j]
    return x


def forward(sd, x, calibrate=False):
    """``model(x)`` / ``model.predict(x)`` (eval mode, no grad): x [N,3,H,W] with H, W multiples of 32 -> probabilities
    [N,1,H,W]."""
    if x.shape[-1] % 32 or x.shape[-2] % 32:
        raise RuntimeError(f'Wrong input shape height={x.shape[-2]}, width={x.shape[-1]}. Expected image height and width divisible by 32.')
    d = decoder(sd, encoder(sd, x, calibrate), calibrate)
    z = F.conv2d(d, sd['segmentation_head.0.weight'].to(x.dtype), sd['segmentation_head.0.bias'].to(x.dtype))
    return torch.sigmoid(z)


def logits(sd, x):
    d = decoder(sd, encoder(sd, x))
    return F.conv2d(d, sd['segmentation_head.0.weight'].to(x.dtype), sd['segmentation_head.0.bias'].to(x.dtype))
