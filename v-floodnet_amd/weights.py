"""Parameter containers with the reference's 562 state-dict names.

The reference network (``video_module/model/AFB_URR.py:242-253``) is a plain
``nn.Module`` whose checkpoint (``train_video_seg.py:159-177``) is the only way
real weights arrive, so the product keeps *exactly* the same parameter tree:

    encoder_m.{conv1_m,conv1_o,conv1,bn1,res2,res3,res4,mean,std}     AFB_URR.py:33-50
    encoder_q.{conv1,bn1,res2,res3,res4,mean,std}                      AFB_URR.py:66-80
    keyval_r4.{Key,Value}                                               AFB_URR.py:96-103
    decoder.{convFM,ResMM,RF3,RF2,pred2,local_convFM,local_ResMM,local_pred2}   AFB_URR.py:181-202

``res2/res3/res4`` are torchvision ResNet-50 ``layer1..3`` (v1.5 bottlenecks:
``conv1 1x1 / conv2 3x3 (stride here) / conv3 1x1 x4 / downsample.{0,1}``).

These modules are *containers only*: nothing here has a ``forward``.  The math
runs in the HIP engine (``engine.py``) on weights repacked by ``pack_*`` below
(NHWC / [Cout][kh][kw][Cin] order, eval-mode BatchNorm expressed as a per-channel
``scale, shift`` epilogue).
"""
import torch
from torch import nn


def _conv(cin, cout, k, stride=1, bias=True):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=bias)


class _Holder(nn.Module):
    """A module that only owns children / parameters."""

    def forward(self, *a, **k):  # pragma: no cover - containers are never called
        raise RuntimeError('parameter container: the forward pass lives in the HIP engine')


def _bottleneck(cin, planes, stride, with_down):
    b = _Holder()
    b.conv1 = _conv(cin, planes, 1, bias=False)
    b.bn1 = nn.BatchNorm2d(planes)
    b.conv2 = _conv(planes, planes, 3, stride=stride, bias=False)
    b.bn2 = nn.BatchNorm2d(planes)
    b.conv3 = _conv(planes, planes * 4, 1, bias=False)
    b.bn3 = nn.BatchNorm2d(planes * 4)
    if with_down:
        b.downsample = nn.Sequential(_conv(cin, planes * 4, 1, stride=stride, bias=False),
                                     nn.BatchNorm2d(planes * 4))
    b.stride = stride
    return b


def _res_layer(cin, planes, blocks, stride):
    layers = [_bottleneck(cin, planes, stride, True)]
    for _ in range(1, blocks):
        layers.append(_bottleneck(planes * 4, planes, 1, False))
    return nn.Sequential(*layers)


def _trunk(holder):
    holder.conv1 = _conv(3, 64, 7, stride=2, bias=False)
    holder.bn1 = nn.BatchNorm2d(64)
    holder.res2 = _res_layer(64, 64, 3, 1)
    holder.res3 = _res_layer(256, 128, 4, 2)
    holder.res4 = _res_layer(512, 256, 6, 2)
    holder.register_buffer('mean', torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
    holder.register_buffer('std', torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))


def make_encoder_m():
    e = _Holder()
    e.conv1_m = _conv(1, 64, 7, stride=2, bias=False)
    e.conv1_o = _conv(1, 64, 7, stride=2, bias=False)
    _trunk(e)
    return e


def make_encoder_q():
    e = _Holder()
    _trunk(e)
    return e


def make_keyval(indim=1024, keydim=128, valdim=512):
    kv = _Holder()
    kv.Key = _conv(indim, keydim, 3)
    kv.Value = _conv(indim, valdim, 3)
    kv.keydim, kv.valdim = keydim, valdim
    return kv


def _resblock(dim):
    r = _Holder()
    r.conv1 = _conv(dim, dim, 3)
    r.conv2 = _conv(dim, dim, 3)
    return r


def _refine(inplanes, planes):
    r = _Holder()
    r.convFS = _conv(inplanes, planes, 3)
    r.ResFS = _resblock(planes)
    r.ResMM = _resblock(planes)
    return r


def make_decoder(mdim_global=256, mdim_local=32):
    d = _Holder()
    d.convFM = _conv(1024, mdim_global, 3)
    d.ResMM = _resblock(mdim_global)
    d.RF3 = _refine(512, mdim_global)
    d.RF2 = _refine(256, mdim_global)
    d.pred2 = _conv(mdim_global, 2, 3)
    d.local_convFM = _conv(128, mdim_local, 3)
    d.local_ResMM = _resblock(mdim_local)
    d.local_pred2 = _conv(mdim_local, 2, 3)
    return d


# --------------------------------------------------------------------------
# repacking for the HIP kernels
# --------------------------------------------------------------------------
BN_EPS = 1e-5


def bn_scale_shift(bn):
    """Eval-mode BatchNorm ``y=(x-mu)/sqrt(var+eps)*g+b`` as ``y = x*scale + shift``."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + BN_EPS)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale.contiguous(), shift.contiguous()


def pack_conv_weight(w):
    """[Cout,Cin,kh,kw] -> [Cout, kh*kw*Cin] with K ordered (kh, kw, cin): the
    implicit-GEMM kernels walk K one filter tap at a time over NHWC activations."""
    cout, cin, kh, kw = w.shape
    return w.detach().float().permute(0, 2, 3, 1).reshape(cout, kh * kw * cin).contiguous()


def conv_epilogue(conv, bn=None):
    """(scale, shift) per output channel: BN fold or plain bias."""
    cout = conv.weight.shape[0]
    dev = conv.weight.device
    if bn is not None:
        return bn_scale_shift(bn)
    scale = torch.ones(cout, device=dev)
    shift = conv.bias.detach().float().clone() if conv.bias is not None else torch.zeros(cout, device=dev)
    return scale, shift.contiguous()
