"""TEST INFRASTRUCTURE (build container only): BASELINE config C1 through THE REFERENCE's own
``test_image_seg.predict_one`` / ``predict_pil`` / ``test_waterseg`` (test_image_seg.py:67-151) and the reference's
``myutils.add_overlay`` (myutils/data.py:56-75), imported under oracle/refstubs.py, with ``tools/standin.StandIn``
in place of the pickled LinkNet -> tests/golden/image_seg_c1.npz and tests/golden/overlay_cases.npz.

    python oracle/gen_image_seg_golden.py
"""
import os
import shutil
import sys
import tempfile
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')


def overlay_case_inputs():
    """The label maps of tests/test_kernels_gpu.py::test_overlay_device_matches_reference (seeded)."""
    g = torch.Generator().manual_seed(11)
    H, W = 41, 67
    frame = torch.rand(3, H, W, generator=g)
    blob = torch.nn.functional.avg_pool2d(torch.rand(1, 1, H, W, generator=g), 5, 1, 2)[0, 0]
    masks = {'two': (blob > 0.5).to(torch.uint8),
             'three': (blob > 0.45).to(torch.uint8) + (blob > 0.55).to(torch.uint8),
             'no_background': 1 + (blob > 0.5).to(torch.uint8),
             'single': torch.ones(H, W, dtype=torch.uint8),
             'gap': 2 * (blob > 0.5).to(torch.uint8)}                     # labels {0, 2}: palette row 2
    return frame, masks


def main():
    from PIL import Image
    from tools import synth
    from tools.standin import StandIn
    from oracle import refstubs
    ref = refstubs.import_reference()
    tis = ref.test_image_seg
    torch.set_num_threads(1)
    cpu = torch.device('cpu')
    out = {}
    tmp = tempfile.mkdtemp()
    try:
        for name, (seed, H, W) in {'c1': (1, 480, 854), 'small': (3, 120, 214)}.items():
            frames, _ = synth.clip(seed, 1, H, W)
            src = os.path.join(tmp, f'{name}.png')
            u8 = (frames[0] * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
            Image.fromarray(u8).save(src)
            # test_waterseg does torch.load(model_path) of a pickled module (:133): hand it the stand-in instead
            orig_load = torch.load
            torch.load = lambda *a, **k: StandIn()
            try:
                tis.test_waterseg('unused.pth', src, name, os.path.join(tmp, 'out'), cpu)
            finally:
                torch.load = orig_load
            mask = Image.open(os.path.join(tmp, 'out', name, 'mask', f'{name}.png'))
            assert mask.mode == 'P'
            lab = np.array(mask)
            ov = np.array(Image.open(os.path.join(tmp, 'out', name, 'overlay', f'{name}.png')))
            out[f'{name}_seed_hw'] = np.array([seed, H, W])
            out[f'{name}_labels'] = np.packbits(lab, axis=-1)
            out[f'{name}_palette'] = np.array(mask.getpalette()[:768])
            out[f'{name}_overlay_crc'] = np.array(zlib.crc32(ov.tobytes()))
            out[f'{name}_water_frac'] = np.array(lab.mean())
            if name == 'small':
                out['small_overlay'] = ov
                out['small_frame_u8'] = u8
                # norm_imagenet on its own (test_image_seg.py:44-64)
                out['small_norm'] = tis.norm_imagenet(Image.fromarray(u8), (416, 416)).numpy()[:, ::16, ::16].copy()
            print(name, lab.shape, 'water', lab.mean())
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(OUT, 'image_seg_c1.npz'), **out)

    frame, masks = overlay_case_inputs()
    img = (frame.permute(1, 2, 0).numpy() * 255).astype(np.uint8)          # save_overlay's conversion (data.py:79)
    bgr = np.ascontiguousarray(img[..., ::-1])
    ov = {'frame': frame.numpy()}
    for n, m in masks.items():
        ov['mask_' + n] = m.numpy()
        ov['bgr_out_' + n] = ref.myutils.add_overlay(bgr.copy(), m.numpy(), ref.myutils.color_palette)
    np.savez_compressed(os.path.join(OUT, 'overlay_cases.npz'), **ov)
    print('written', [f'{f}: {os.path.getsize(os.path.join(OUT, f)) / 1e3:.0f} kB' for f in ('image_seg_c1.npz', 'overlay_cases.npz')])


if __name__ == '__main__':
    main()
