"""BASELINE config C1: one 480x854 frame through the test_image_seg.py plumbing on CPU (no GPU) with a stand-in
``.predict`` (the LinkNet weights / package are not available; only the contract around it is in scope).

Pinned against THE REFERENCE: tests/golden/image_seg_c1.npz holds what the reference's own ``test_waterseg`` /
``predict_one`` / ``predict_pil`` / ``norm_imagenet`` (test_image_seg.py:44-151) wrote for the same frames with the same
stand-in (oracle/gen_image_seg_golden.py)."""
import os
import zlib

import numpy as np
import pytest
import torch
from PIL import Image

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'image_seg_c1.npz')


def _run(tmp_path, name, seed, H, W):
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import image_seg
    from tools import synth
    from tools.standin import StandIn
    frames, _ = synth.clip(seed, 1, H, W)
    src = tmp_path / f'{name}.png'
    u8 = (frames[0] * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
    Image.fromarray(u8).save(str(src))
    image_seg.test_waterseg('unused.pth', str(src), name, str(tmp_path / 'out'), torch.device('cpu'), model=StandIn())
    mask = Image.open(str(tmp_path / 'out' / name / 'mask' / f'{name}.png'))
    ov = Image.open(str(tmp_path / 'out' / name / 'overlay' / f'{name}.png'))
    return u8, mask, ov


@pytest.mark.parametrize('name', ['c1', 'small'])
def test_image_seg_matches_reference_files(tmp_path, name):
    g = np.load(GOLD)
    seed, H, W = [int(x) for x in g[f'{name}_seed_hw']]
    u8, mask, ov = _run(tmp_path, name, seed, H, W)
    assert mask.mode == 'P' and mask.size == (W, H)
    assert mask.getpalette()[:768] == [int(x) for x in g[f'{name}_palette']]
    lab = np.array(mask)
    assert np.array_equal(lab, np.unpackbits(g[f'{name}_labels'], axis=-1)[:, :W])        # label map: exact
    assert ov.mode == 'RGB' and ov.size == (W, H)
    assert zlib.crc32(np.array(ov).tobytes()) == int(g[f'{name}_overlay_crc'])           # overlay: byte-exact
    if name == 'small':
        assert np.array_equal(u8, g['small_frame_u8'])
        assert np.array_equal(np.array(ov), g['small_overlay'])
    # the output is one 8-connected blob (postprocessing_pred), and the stand-in produced more than one before it
    from scipy import ndimage
    assert ndimage.label(lab, structure=np.ones((3, 3)))[1] == 1
    assert 0.05 < lab.mean() < 0.95


def test_norm_imagenet_matches_reference():
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import image_seg
    g = np.load(GOLD)
    t = image_seg.norm_imagenet(Image.fromarray(g['small_frame_u8']), (416, 416))
    assert t.shape == (3, 416, 416)
    assert np.array_equal(t.numpy()[:, ::16, ::16], g['small_norm'])
