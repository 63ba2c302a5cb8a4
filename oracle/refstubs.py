"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Stub modules that let the *reference* (``/root/reference``, read-only, present
only in the build container) be imported on CPU, so that golden vectors can be
generated from the reference's own Python.

The reference imports four third-party packages that are absent from this
image.  Each stub restates the published semantics of the pinned version the
reference's README names; none of this is reference source:

* ``torchvision==0.9.2`` (README.md:44,57)
    - ``torchvision.models.resnet50``: ResNet-50 v1.5 layer definitions
      (Bottleneck = 1x1 -> 3x3(stride) -> 1x1(x4), downsample.{0,1}), layers
      3-4-6-3.  Only the module *structure* / state-dict names matter here; the
      weights are always overwritten by a checkpoint.
    - ``torchvision.transforms.functional.resize`` on tensors: short-edge
      rule, identity short-circuit, ``int(size*long/short)``,
      ``F.interpolate(bicubic|nearest, align_corners=False)`` without antialias.
    - ``ToTensor``: HWC uint8 -> CHW float / 255.
* ``torch-scatter==2.0.8`` (README.md:58): ``scatter_mean(src, index, dim, out)``
  = scatter_add into ``out``, count = scatter_add(ones), count.clamp_(min=1),
  ``out.true_divide_(count)``.
* ``opencv-python==4.4.0.46`` (requirements.txt:11):
  ``connectedComponentsWithAlgorithm(img, 8, CV_32S, CCL_GRANA)`` -> 8-connected
  labelling of non-zero pixels, labels numbered in raster-scan order of first
  appearance (background 0); ``cvtColor(RGB2BGR)``; ``imwrite``.

Usage (build container only)::

    from oracle import refstubs
    ref = refstubs.import_reference()      # -> namespace with AFB_URR, FeatureBank, myutils, ...
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

REFERENCE_ROOT = '/root/reference'


# --------------------------------------------------------------------------
# torchvision.models.resnet50  (structure only)
# --------------------------------------------------------------------------
class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class _ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 3, 1)
        self.layer2 = self._make_layer(128, 4, 2)
        self.layer3 = self._make_layer(256, 6, 2)
        self.layer4 = self._make_layer(512, 3, 2)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(_Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)


def _resnet50(pretrained=False, **kw):
    if pretrained:
        raise RuntimeError('no network: ImageNet weights unavailable')
    return _ResNet50()


# --------------------------------------------------------------------------
# torchvision.transforms(.functional)
# --------------------------------------------------------------------------
class _InterpolationMode:
    NEAREST = 'nearest'
    BILINEAR = 'bilinear'
    BICUBIC = 'bicubic'


def _is_pil_image(img):
    from PIL import Image
    return isinstance(img, Image.Image)


def _tv_resize(img, size, interpolation=_InterpolationMode.BILINEAR):
    """torchvision 0.9.2 ``functional_tensor.resize`` for a tensor input."""
    if not isinstance(img, torch.Tensor):
        raise TypeError('stub resize handles tensors only')
    h, w = img.shape[-2:]
    if isinstance(size, int):
        size = [size]
    if len(size) == 1:
        req = size[0]
        short, long = (w, h) if w <= h else (h, w)
        if short == req:
            return img
        new_short, new_long = req, int(req * long / short)
        new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    else:
        new_h, new_w = size
    squeeze = img.dim() == 3
    x = img.unsqueeze(0) if squeeze else img
    odt = x.dtype
    need_cast = odt not in (torch.float32, torch.float64)
    if need_cast:
        x = x.to(torch.float32)
    if interpolation == _InterpolationMode.NEAREST:
        x = F.interpolate(x, size=[new_h, new_w], mode='nearest')
    else:
        x = F.interpolate(x, size=[new_h, new_w], mode=interpolation, align_corners=False)
        if interpolation == _InterpolationMode.BICUBIC and odt == torch.uint8:
            x = x.clamp(0, 255)
    if need_cast:
        if odt in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
            x = torch.round(x)
        x = x.to(odt)
    return x.squeeze(0) if squeeze else x


class _ToTensor:
    def __call__(self, pic):
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
        if t.dtype == torch.uint8:
            return t.to(torch.float32).div(255)
        return t


class _Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class _Resize:
    """torchvision 0.9.2 ``transforms.Resize(size)``, default interpolation BILINEAR: a PIL image goes through
    ``img.resize((w, h), PIL.Image.BILINEAR)`` (``functional_pil.resize``), a tensor through ``_tv_resize``
    (``F.interpolate(bilinear, align_corners=False)``, no antialias).  Used by test_image_seg.py:57-61,109."""

    def __init__(self, size, interpolation=_InterpolationMode.BILINEAR):
        self.size, self.interpolation = size, interpolation

    def __call__(self, img):
        if _is_pil_image(img):
            from PIL import Image
            assert not isinstance(self.size, int) and len(self.size) == 2 and self.interpolation == _InterpolationMode.BILINEAR
            return img.resize((self.size[1], self.size[0]), Image.BILINEAR)
        return _tv_resize(img, self.size, self.interpolation)


class _Normalize:
    """``transforms.Normalize(mean, std)``: ``(x - mean[:, None, None]) / std[:, None, None]`` (tensor.sub_().div_())."""

    def __init__(self, mean, std):
        self.mean, self.std = torch.as_tensor(mean, dtype=torch.float32), torch.as_tensor(std, dtype=torch.float32)

    def __call__(self, t):
        return t.clone().sub_(self.mean.view(-1, 1, 1)).div_(self.std.view(-1, 1, 1))


# --------------------------------------------------------------------------
# torch_scatter.scatter_mean
# --------------------------------------------------------------------------
def _scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    assert out is not None
    out.scatter_add_(dim, index, src)
    count = torch.zeros_like(out)
    count.scatter_add_(dim, index, torch.ones_like(src))
    count.clamp_(min=1)
    out.true_divide_(count)
    return out


# --------------------------------------------------------------------------
# cv2 (only what myutils / test_video_seg touch)
# --------------------------------------------------------------------------
def _ccl_raster8(img):
    """8-connected labelling, labels in raster order of first pixel (OpenCV order)."""
    from scipy import ndimage
    lab, n = ndimage.label(img != 0, structure=np.ones((3, 3), np.int32))
    return n + 1, lab.astype(np.int32)


def _make_cv2():
    cv2 = types.ModuleType('cv2')
    cv2.CV_32S = 4
    cv2.CCL_GRANA = 1
    cv2.COLOR_RGB2BGR = 4
    cv2.COLOR_BGR2RGB = 4

    def connectedComponentsWithAlgorithm(image, connectivity, ltype, ccltype):
        assert connectivity == 8
        return _ccl_raster8(image)

    def cvtColor(img, code):
        return np.ascontiguousarray(img[..., ::-1])

    def imwrite(path, img):
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(img[..., ::-1])).save(path)
        return True

    cv2.connectedComponentsWithAlgorithm = connectedComponentsWithAlgorithm
    cv2.cvtColor = cvtColor
    cv2.imwrite = imwrite
    return cv2


def install():
    """Insert the stub modules into ``sys.modules`` (idempotent)."""
    if 'torchvision' in sys.modules and getattr(sys.modules['torchvision'], '_vfn_stub', False):
        return
    tv = types.ModuleType('torchvision')
    tv._vfn_stub = True
    tv.__path__ = []
    models = types.ModuleType('torchvision.models')
    models.resnet50 = _resnet50
    transforms = types.ModuleType('torchvision.transforms')
    transforms.__path__ = []
    functional = types.ModuleType('torchvision.transforms.functional')
    functional.resize = _tv_resize
    functional.InterpolationMode = _InterpolationMode
    functional._is_pil_image = _is_pil_image
    transforms.functional = functional
    transforms.InterpolationMode = _InterpolationMode
    transforms.ToTensor = _ToTensor
    transforms.Compose = _Compose
    transforms.Resize = _Resize
    transforms.Normalize = _Normalize
    tv.models = models
    tv.transforms = transforms
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = models
    sys.modules['torchvision.transforms'] = transforms
    sys.modules['torchvision.transforms.functional'] = functional

    ts = types.ModuleType('torch_scatter')
    ts.scatter_mean = _scatter_mean
    sys.modules['torch_scatter'] = ts

    sys.modules['cv2'] = _make_cv2()


def import_reference():
    """Import the reference's hot-path modules under the stubs.  Container only."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f'{REFERENCE_ROOT} is not present (GPU box?): the reference '
                           'can only be imported in the build container')
    install()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        import myutils
        from video_module.model import AFB_URR, FeatureBank
        from video_module.dataset import Video_DS
        import test_video_seg
        import test_image_seg
    ns = types.SimpleNamespace(myutils=myutils, AFB_URR=AFB_URR, FeatureBank=FeatureBank,
                               Video_DS=Video_DS, test_video_seg=test_video_seg, test_image_seg=test_image_seg)
    return ns
