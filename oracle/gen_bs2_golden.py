"""TEST INFRASTRUCTURE (build container only): ``AFB_URR.segment`` of THE REFERENCE with a batch of frames
(bs = 2, as ``train_video_seg.py:65-69`` calls it with ``frames[1:]``) in eval mode (padded 90x150 frames) and in
training mode with BatchNorm frozen (``model.train(); model.apply(set_bn_eval)``, train_video_seg.py:103-106;
96x160 frames, no padding, scalar uncertainty) -> tests/golden/segment_bs2.npz.

    python oracle/gen_bs2_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED = 20200212


def main():
    from tools import synth
    from oracle import refstubs
    ref = refstubs.import_reference()
    torch.set_num_threads(8)
    cpu = torch.device('cpu')
    sd = synth.make_state_dict(SEED)
    out = {}
    for tag, (H, W, training) in {'eval_90x150': (90, 150, False), 'train_96x160': (96, 160, True)}.items():
        model = ref.AFB_URR(cpu, update_bank=not training, load_imagenet_params=False)
        model.load_state_dict(sd, strict=True)
        if training:
            model.train()
            model.apply(ref.myutils.set_bn_eval)
        else:
            model.eval()
        frames, m0 = synth.clip(6, 3, H, W)
        oh = synth.onehot(m0).unsqueeze(0)
        with torch.no_grad():
            fb = ref.FeatureBank(2, 250000, cpu)
            k, v = model.memorize(frames[0:1], oh)
            fb.init_bank(k, v)
            score, unc = model.segment(frames[1:3], fb)
        assert score.shape == (2, 2, H, W)
        out[f'{tag}_score'] = score.numpy()
        out[f'{tag}_info1'] = np.stack([fb.info[i][:, 1].numpy() for i in range(2)])
        if training:
            out[f'{tag}_uncertainty'] = np.array(float(unc))
        print(tag, 'score range', float(score.min()), float(score.max()), 'uncertainty', None if unc is None else float(unc))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'segment_bs2.npz'), **out)
    print('written', os.path.getsize(os.path.join(ROOT, 'tests', 'golden', 'segment_bs2.npz')) / 1e3, 'kB')


if __name__ == '__main__':
    main()
