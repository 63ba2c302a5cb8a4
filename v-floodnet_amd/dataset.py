"""``Video_DS``: the test-time dataset of ``video_module/dataset/Water_DS.py:87-111``.

Same constructor and item contract: ``Video_DS(img_list, first_frame, first_mask)`` with PIL inputs,
``.obj_n``, ``.first_frame`` float[3,H,W] in [0,1], ``.first_mask`` uint8 one-hot [obj_n,H,W],
``ds[i] -> (frame float[3,H,W], name)`` for ``img_list[1:]``.
"""
import os

import numpy as np
import torch
from torch.utils import data

from .data import load_image_in_PIL


def to_tensor(pic):
    """torchvision ``ToTensor``: HWC uint8 -> CHW float32 / 255."""
    arr = np.asarray(pic)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
    return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t


def to_onehot(mask, max_obj_n):
    """``ToOnehot(max_obj_n, shuffle=False)`` (transforms.py:383-421): channel 0 = 1 - sum(objects)."""
    mask = np.asarray(mask)
    new_mask = np.zeros((max_obj_n, *mask.shape), np.uint8)
    obj_list = [i for i in range(1, int(mask.max()) + 1) if (mask == i).any()][:max_obj_n - 1]
    for i, o in enumerate(obj_list):
        new_mask[i + 1] = (mask == o).astype(np.uint8)
    new_mask[0] = 1 - np.sum(new_mask, axis=0)
    return torch.from_numpy(new_mask), obj_list


class Video_DS(data.Dataset):
    """dataset/Water_DS.py:105-139.  ``raw_u8=True`` (an extension used by ``video_seg.main``): ``__getitem__`` hands
    out the decoded uint8 HWC frame and ``ToTensor`` runs on the GPU (``ops.to_tensor_device``, bit-identical),
    so a frame crosses PCIe as 1 byte per sample."""

    def __init__(self, img_list, first_frame, first_mask, raw_u8=False):
        self.raw_u8 = raw_u8
        self.img_list = img_list[1:]
        self.video_len = len(self.img_list)
        first_mask = np.array(first_mask, np.uint8) > 0
        self.obj_n = int(first_mask.max()) + 1
        mask, _ = to_onehot(first_mask, self.obj_n)
        self.first_mask = mask[:self.obj_n]
        self.first_frame = to_tensor(first_frame)

    def __len__(self):
        return self.video_len

    def __getitem__(self, idx):
        img = load_image_in_PIL(self.img_list[idx], 'RGB')
        frame = torch.from_numpy(np.array(img, np.uint8)) if self.raw_u8 else to_tensor(img)
        img_name = os.path.basename(self.img_list[idx])[:-4]
        return frame, img_name
