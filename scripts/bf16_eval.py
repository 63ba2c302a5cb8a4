"""bf16 mode on the C2 clip: masks vs the reference's own labels (tests/golden/c2_480x854_100.npz) and vs the
f32 HIP run; frames/s of both."""
import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np, torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR
from tools import synth
from vfloodnet_amd.video_seg import run_clip

gpu = torch.device('cuda', 0)
g = np.load(os.path.join(root, 'tests/golden/c2_480x854_100.npz'))
H, W = [int(x) for x in g['shape']]
ref = np.unpackbits(g['labels'], axis=-1)[..., :W]
T = ref.shape[0]
frames, m0 = synth.clip(int(g['seed']), T, H, W)
frames = frames.to(gpu)
sd = synth.make_state_dict(20200212)

def miou(a, b):
    out = []
    for c in (0, 1):
        i = ((a == c) & (b == c)).sum(); u = ((a == c) | (b == c)).sum()
        out.append(1.0 if u == 0 else i / u)
    return float(sum(out) / 2)

labs = {}
for prec in ('fp32', 'bf16x3', 'bf16'):
    model = AFB_URR(gpu, update_bank=True, precision=prec).to(gpu).eval(); model.load_state_dict(sd)
    run_clip(model, frames[:5], m0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = run_clip(model, frames, m0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lab = out['labels'].numpy(); labs[prec] = lab
    ious = [miou(lab[t], ref[t]) for t in range(1, T)]
    print(f'{prec}: {(T - 1) / dt:6.1f} frames/s ({1e3 * dt / (T - 1):.2f} ms/frame); mIoU vs reference labels min {min(ious):.4f} '
          f'mean {np.mean(ious):.4f}; final bank {out["bank_sizes"][-1]}')
for pr in ('bf16x3', 'bf16'):
    ious = [miou(labs[pr][t], labs['fp32'][t]) for t in range(1, T)]
    print(pr, 'vs fp32 HIP: mIoU min %.4f mean %.4f' % (min(ious), np.mean(ious)))

print(' '.join('%.3f' % i for i in ious))
