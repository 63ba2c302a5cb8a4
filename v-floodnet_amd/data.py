"""Helpers of the reference's ``myutils/data.py`` that the inference loop touches.

``postprocessing_pred`` / ``save_seg_mask`` / ``save_overlay`` / ``color_palette`` /
``load_image_in_PIL`` / ``pad_divide_by`` / ``calc_uncertainty`` keep the reference's names and
argument meaning (myutils/data.py:14-90,132-149); OpenCV is not a dependency: the connected
components come from ``vfn_postprocess_pred_u8`` and the overlay PNG is written with PIL.
"""
import numpy as np
import torch
from PIL import Image

from . import ops
from .engine import pad_divide_by as _pad_amounts

color_palette = [0, 0, 0, 0, 0, 128, 0, 128, 0, 128, 0, 0] + [100, 100, 100] * 252     # data.py:14


def postprocessing_pred(pred):
    """data.py:17-37: keep the largest 8-connected water component (all-background -> all ones)."""
    return ops.postprocess_pred(np.asarray(pred, dtype=np.uint8))


def save_seg_mask(pred, seg_path, palette=color_palette):
    """data.py:49-53: mode-P PNG with the reference palette."""
    seg_img = Image.fromarray(np.asarray(pred, dtype=np.uint8))
    seg_img.putpalette(palette)
    seg_img.save(seg_path)


def add_overlay(img, mask, colors=color_palette, alpha=0.4, cscale=1):
    """Host evaluation of the reference overlay (myutils/data.py:56-75; img BGR uint8 [H,W,3], mask uint8 [H,W]) --
    used where no GPU is involved (BASELINE config C1, ``image_seg.predict_one``); the video loop uses
    ``vfn_overlay_u8``.  Stated as one rule per pixel instead of the reference's per-label painting loop:

    * labels are painted in ascending order and the smallest label present is never painted (background);
    * a pixel of a painted label L shows ``trunc(img*alpha + (1-alpha)*cscale*palette[L])`` (float64, BGR);
    * the 1-pixel outline of label j (its 4-neighbour dilation minus itself) is drawn right after j is painted, so a
      pixel ends up black iff one of its 4 neighbours carries a painted label greater than its own.
    """
    mask = np.asarray(mask)
    present = np.unique(mask)
    pal = np.zeros((256, 3), np.float64)
    rows = np.asarray(colors, dtype=np.float64).reshape(-1, 3)[:256]
    pal[:len(rows)] = rows[:, ::-1] * cscale                         # palette is RGB, the image BGR
    painted = mask != present[0]
    out = img.copy()
    blend = img[painted] * alpha + (1 - alpha) * pal[mask[painted]]
    out[painted] = blend                                             # float64 -> uint8: truncation
    lab = np.where(painted, mask.astype(np.int32), -1)               # background never outlines anything
    nb = np.full(mask.shape, -1, np.int32)
    nb[1:, :] = np.maximum(nb[1:, :], lab[:-1, :])
    nb[:-1, :] = np.maximum(nb[:-1, :], lab[1:, :])
    nb[:, 1:] = np.maximum(nb[:, 1:], lab[:, :-1])
    nb[:, :-1] = np.maximum(nb[:, :-1], lab[:, 1:])
    out[nb > lab] = 0
    return out


def save_overlay(img, mask, overlay_path, colors=[255, 0, 0], alpha=0.4, cscale=1):
    """data.py:78-84: img float [3,H,W] in [0,1] (any device) -> BGR overlay PNG."""
    img = (img.permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8)
    bgr = np.ascontiguousarray(img[..., ::-1])
    ov = add_overlay(bgr, mask, colors, alpha, cscale)
    Image.fromarray(np.ascontiguousarray(ov[..., ::-1])).save(overlay_path)      # cv2.imwrite(BGR) == save(RGB)


def save_overlay_device(img, mask_dev, overlay_path, colors=color_palette, alpha=0.4, cscale=1):
    """save_overlay with the blend / contour / uint8 conversion on the GPU (``vfn_overlay_u8``, bit-identical to
    add_overlay): img float [3,H,W] and mask uint8 [H,W] on the device; only the RGB uint8 image crosses PCIe."""
    ov = ops.overlay_device(img.contiguous(), mask_dev, colors, alpha, cscale)
    Image.fromarray(ov.cpu().numpy()).save(overlay_path)


class AsyncWriter:
    """PNG encoding off the loop thread: zlib-compressing a 480p overlay takes 15-25 ms on one core -- more than two
    frames of GPU time -- so ``video_seg.main`` hands finished arrays to a small thread pool (Pillow releases the GIL
    while it encodes).  Files are byte-identical to the synchronous path; ``close()`` waits and re-raises."""

    def __init__(self, workers=4):
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=workers)
        self._pending = []

    def submit(self, fn, *args):
        self._pending.append(self._pool.submit(fn, *args))
        if len(self._pending) > 64:                      # bound the queue (and surface errors early)
            self._pending.pop(0).result()

    def close(self):
        for f in self._pending:
            f.result()
        self._pending = []
        self._pool.shutdown(wait=True)


def _save_rgb(arr, path):
    Image.fromarray(arr).save(path)


def save_overlay_device_async(writer, img, mask_dev, overlay_path, colors=color_palette, alpha=0.4, cscale=1):
    """``save_overlay_device`` with the PNG encoding on ``writer``'s threads."""
    ov = ops.overlay_device(img.contiguous(), mask_dev, colors, alpha, cscale).cpu().numpy()
    writer.submit(_save_rgb, ov, overlay_path)


def load_image_in_PIL(path, mode='RGB'):
    """data.py:87-90."""
    img = Image.open(path)
    img.load()
    return img.convert(mode)


def pad_divide_by(in_list, d, in_size):
    """data.py:132-149 for callers that want the padded tensors (the engine fuses the padding instead)."""
    pad, _, _ = _pad_amounts(in_size[0], in_size[1], d)
    return [torch.nn.functional.pad(x, pad) for x in in_list], pad
