"""Round-4 GPU tests: the look-ahead caches of the main loop on a clip that is resized on the device, and the
bench line's round-4 fields on the driver's command."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 20200212


def miou(a, b):
    v = []
    for c in (0, 1):
        inter = ((a == c) & (b == c)).sum()
        union = ((a == c) | (b == c)).sum()
        v.append(1.0 if union == 0 else inter / union)
    return sum(v) / 2


def test_main_on_a_resized_clip_does_not_reuse_stale_lookahead_entries(gpu, tmp_path, monkeypatch):
    """``video_seg.main`` on 26 frames of 120 x 200 run at a 96-pixel short edge: every decoded frame is a fresh allocation that
    the caching allocator hands out again a few frames later, and the resized-frame cache / the query-side prefetch are keyed on
    allocator addresses.  They now own their source tensors, so a recycled address cannot hit an older frame's entry: the masks
    must agree with the run that never looks ahead (VFN_LOOKAHEAD=0; up to the summation order of the batched query pass) --
    before the fix frames from ~19 on were segmented from the pixels of older frames."""
    from PIL import Image
    from tools import synth
    from vfloodnet_amd import video_seg
    from vfloodnet_amd.data import save_seg_mask, color_palette
    T, H, W = 26, 120, 200
    frames, m0 = synth.clip(13, T, H, W)
    fdir = tmp_path / 'frames'
    fdir.mkdir()
    for t in range(T):
        Image.fromarray((frames[t].permute(1, 2, 0).numpy() * 255).astype(np.uint8)).save(str(fdir / f'{t:05d}.jpg'), quality=95)
    ckpt = str(tmp_path / 'ckpt.pth')
    torch.save({'epoch': 0, 'model': synth.make_state_dict(SEED), 'loss': 0.0, 'seed': SEED}, ckpt)
    out = {}
    for tag, la in (('ahead', '3'), ('none', '0')):
        run = tmp_path / tag
        (run / 'output' / 'segs' / 'clip' / 'mask').mkdir(parents=True)
        save_seg_mask(m0.numpy().astype(np.uint8), str(run / 'output' / 'segs' / 'clip' / 'mask' / '00000.png'), color_palette)
        monkeypatch.chdir(run)
        monkeypatch.setenv('VFN_LOOKAHEAD', la)
        args = argparse.Namespace(gpu=0, budget=250000, viz=False, model_path=ckpt, update_rate=0.1, merge_thres=0.95,
                                  test_path=str(fdir), test_name='clip', size=96, load_workers=2)
        video_seg.main(args, gpu)
        out[tag] = [np.array(Image.open(str(run / 'output' / 'segs' / 'clip' / 'mask' / f'{t:05d}.png'))) for t in range(T)]
    ious = [miou(a, b) for a, b in zip(out['ahead'], out['none'])]
    assert min(ious) > 0.995, [round(float(x), 4) for x in ious]
    # the clip moves: a frame segmented from an older frame's pixels would be far off its neighbour-in-time's mask as well
    assert all(o.shape == (H, W) for o in out['ahead'])
