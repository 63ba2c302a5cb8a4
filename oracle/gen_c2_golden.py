"""Full BASELINE config C2 (100-frame 480x854 synthetic clip, fp32) through THE REFERENCE's own model and
feature bank on CPU (build container only) -> tests/golden/c2_480x854_100.npz:
bit-packed label maps before post-processing, per-frame bank sizes, peak / replace counters.

Follows test_video_seg.py:83-121 on in-memory tensors (no PNG round trip: frames stay fp32)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(T=100, seed=1):
    import vfloodnet_amd  # noqa: F401
    from tools import synth
    from oracle import refstubs
    from torch.nn import functional as F
    ref = refstubs.import_reference()
    torch.set_num_threads(8)
    cpu = torch.device('cpu')
    sd = synth.make_state_dict(20200212)
    model = ref.AFB_URR(cpu, update_bank=True, load_imagenet_params=False).eval()
    model.load_state_dict(sd, strict=True)
    frames, m0 = synth.clip(seed, T, 480, 854)
    onehot = synth.onehot(m0).unsqueeze(0)
    fb = ref.FeatureBank(2, 250000, cpu, update_rate=0.1, thres_close=0.95)
    labels = [m0.numpy()]
    sizes = []
    t0 = time.time()
    with torch.no_grad():
        k, v = model.memorize(frames[0:1], onehot)
        fb.init_bank(k, v)
        for t in range(1, T):
            score, _ = model.segment(frames[t:t + 1], fb)
            pm = F.softmax(score, dim=1)
            k, v = model.memorize(frames[t:t + 1], pm)
            fb.update(k, v, t)
            labels.append(torch.argmax(pm[0], dim=0).numpy().astype(np.uint8))     # 480x854 == ori size
            sizes.append([int(fb.keys[i].shape[1]) for i in range(2)])
            if t % 10 == 0:
                print(t, sizes[-1], f'{time.time() - t0:.0f}s', flush=True)
    out = os.path.join(ROOT, 'tests', 'golden', f'c2_480x854_{T}.npz')
    np.savez_compressed(out, labels=np.packbits(np.stack(labels), axis=-1), shape=np.array([480, 854]),
                        bank_sizes=np.array(sizes), peak_n=fb.peak_n, replace_n=fb.replace_n, seed=np.array(seed))
    print('written', out, os.path.getsize(out) / 1e6, 'MB')


if __name__ == '__main__':
    main()
