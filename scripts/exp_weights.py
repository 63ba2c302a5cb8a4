"""Search synthetic-weight knobs for non-chaotic loop dynamics (fp32 vs fp64 oracle, free-running)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfloodnet_amd
from tools import synth
from oracle import afb_urr_ref as O

def miou(x, y):
    v = []
    for c in (0, 1):
        i = ((x == c) & (y == c)).sum().item(); u = ((x == c) | (y == c)).sum().item()
        v.append(1.0 if u == 0 else i / u)
    return sum(v) / 2

def evaluate(knobs, H=64, W=96, T=int(os.environ.get("T", 8)), size=128):
    sd = synth.make_state_dict(20200212, **knobs)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    frames, m0 = synth.clip(3, T, H, W)
    a = O.run_clip(sd, frames, m0, size=size, return_scores=True)
    b = O.run_clip(sd64, frames.double(), m0, size=size, return_scores=True)
    dps, ms = [], []
    for t in range(1, T):
        sa, sb = a['scores'][t - 1], b['scores'][t - 1]
        dps.append((torch.sigmoid(sa.double()) - torch.sigmoid(sb)).abs().max().item())
        ms.append(miou(a['labels'][t], b['labels'][t]))
    s = b['scores'][-1]
    d = (s[0, 1] - s[0, 0]).abs()
    wf = (s[0, 1] > s[0, 0]).float().mean().item()
    print(knobs, '\n   dprob', ' '.join('%.1e' % x for x in dps), '\n   mIoU', ' '.join('%.4f' % x for x in ms),
          '\n   margin median %.3f p1 %.4f water %.3f sizes %s' % (d.median().item(), d.flatten().kthvalue(max(1, int(0.01 * d.numel()))).values.item(), wf, a['bank_sizes'][-1]), flush=True)

if __name__ == '__main__':
    torch.set_num_threads(8)
    for knobs in eval(sys.argv[1]):
        evaluate(knobs)
