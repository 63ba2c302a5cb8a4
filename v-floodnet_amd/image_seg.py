"""First-frame bootstrap plumbing: the reference's ``test_image_seg.py`` contract (BASELINE config C1).

``test_video_seg.py:64-69`` calls ``test_waterseg(model_path, first_frame, name, out_dir, device)`` when the
first-frame mask PNG is missing.  The network behind it (LinkNet + EfficientNet-b4 from the un-vendored
``segmentation_models_pytorch``, a whole pickled module at ``records/link_efficientb4_model.pth``) is *not* part
of the hot path; what is kept here is everything around ``model.predict``:

    norm_imagenet   Resize(416x416, bilinear) -> ToTensor -> Normalize(mean, std)      test_image_seg.py:44-64
    predict_pil     ... -> model.predict -> Resize(back, bilinear) -> round -> postprocessing_pred
                    -> mode-P PNG with the palette                                       test_image_seg.py:95-124
    predict_one     mask + overlay files                                                 test_image_seg.py:67-92
    test_waterseg   file / folder dispatch, output tree                                  test_image_seg.py:127-151

``model`` is any object with ``predict(x: float[1,3,416,416]) -> float[1,1,416,416]`` in [0,1] (the smp API).
This is host plumbing (PIL + a handful of torch CPU ops on one 416x416 image), exactly as in the reference.
"""
import os
from glob import glob
from pathlib import Path

import numpy as np
import torch
from PIL import Image
from torch.nn import functional as F

from .data import add_overlay, color_palette, load_image_in_PIL, postprocessing_pred

MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)


def norm_imagenet(img_pil, dims):
    """test_image_seg.py:44-64: PIL bilinear resize to ``dims`` (h, w), /255, ImageNet normalisation."""
    img = img_pil.resize((dims[1], dims[0]), Image.BILINEAR)
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(img).transpose(2, 0, 1))).float().div(255)
    return (t - MEAN) / STD


def predict_pil(model, img_pil, model_dims, device):
    """test_image_seg.py:95-124 -> mode-P PIL image of 0/1 labels with the reference palette."""
    img_np = np.array(img_pil)
    x = norm_imagenet(img_pil, model_dims).unsqueeze(0)
    try:
        prediction = model.predict(x.to(device))
    except Exception:                                   # the reference retries on the host tensor
        print('Did not convert input image to cuda.')
        prediction = model.predict(x)
    prediction = torch.as_tensor(prediction).float().cpu()
    # tf.Resize on a tensor = bilinear interpolate, align_corners=False, no antialias (torchvision 0.9.2)
    prediction = F.interpolate(prediction, size=[img_np.shape[0], img_np.shape[1]], mode='bilinear', align_corners=False)
    pred = postprocessing_pred(prediction.squeeze().round().numpy().astype(np.uint8))
    out = Image.fromarray(pred).convert('P')
    out.putpalette(color_palette)
    return out


def predict_one(path, model, mask_outdir, overlay_outdir, device):
    """test_image_seg.py:67-92."""
    img_pil = load_image_in_PIL(path)
    prediction = predict_pil(model, img_pil, model_dims=(416, 416), device=device)
    basename = str(Path(os.path.basename(path)).stem)
    prediction.save(os.path.join(mask_outdir, basename + '.png'))
    bgr = np.ascontiguousarray(np.array(img_pil)[..., ::-1])
    overlay = add_overlay(bgr, np.array(prediction))
    Image.fromarray(np.ascontiguousarray(overlay[..., ::-1])).save(os.path.join(overlay_outdir, basename + '.png'))


def test_waterseg(model_path, test_path, test_name, out_path, device, model=None):
    """test_image_seg.py:127-151.  ``model`` overrides ``torch.load(model_path)`` (the pickled smp module
    needs its package to unpickle; a stand-in with ``.predict`` is enough for the plumbing)."""
    if model is None:
        # the reference unpickles a whole smp module and calls its predict (:133); here the parameters are read by name out of
        # whatever the file holds (the pickled module, if its package is importable, or a state dict) and predict runs on the
        # HIP path (linknet.LinknetB4)
        from .linknet import LinknetB4
        model = LinknetB4.from_checkpoint(model_path, device)
    out_path = os.path.join(out_path, test_name)
    mask_out = os.path.join(out_path, 'mask')
    overlay_out = os.path.join(out_path, 'overlay')
    os.makedirs(mask_out, exist_ok=True)
    os.makedirs(overlay_out, exist_ok=True)
    if os.path.isfile(test_path):
        predict_one(test_path, model, mask_out, overlay_out, device)
    elif os.path.isdir(test_path):
        for path in glob(os.path.join(test_path, '*.jpg')) + glob(os.path.join(test_path, '*.png')):
            predict_one(path, model, mask_out, overlay_out, device)
    else:
        print('Error: Unknown path: ', test_path)
        exit(-1)


test_waterseg.__test__ = False       # not a pytest test despite the reference's name
