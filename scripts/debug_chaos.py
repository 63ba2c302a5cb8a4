"""Does fp32 rounding noise grow through the memorize->bank feedback with the synthetic weights?
Oracle fp32 vs oracle fp64 on the same clip, free-running."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfloodnet_amd
from tools import synth
from oracle import afb_urr_ref as O

H, W, T, size = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sd = synth.make_state_dict(20200212)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
frames, m0 = synth.clip(3, T, H, W)
torch.set_num_threads(8)
a = O.run_clip(sd, frames, m0, size=size, return_scores=True)
b = O.run_clip(sd64, frames.double(), m0, size=size, return_scores=True)
def miou(x, y):
    v = []
    for c in (0, 1):
        i = ((x == c) & (y == c)).sum().item(); u = ((x == c) | (y == c)).sum().item()
        v.append(1.0 if u == 0 else i / u)
    return sum(v) / 2
for t in range(1, T):
    sa, sb = a['scores'][t - 1], b['scores'][t - 1]
    dp = (torch.sigmoid(sa.double()) - torch.sigmoid(sb)).abs().max().item()
    print(t, 'mIoU %.5f' % miou(a['labels'][t], b['labels'][t]), 'max dprob %.2e' % dp, a['bank_sizes'][t - 1], b['bank_sizes'][t - 1])
