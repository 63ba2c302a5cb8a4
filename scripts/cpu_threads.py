import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from tools import synth
from oracle import afb_urr_ref as O
sd = synth.make_state_dict(20200212)
frames, m0 = synth.clip(1, 2, 480, 854)
oh = synth.onehot(m0).unsqueeze(0)
print('cpu_count', os.cpu_count())
for n in [8, 16, 32, 64, 128]:
    torch.set_num_threads(n)
    k, v = O.memorize(sd, frames[0:1], oh)
    fb = O.FeatureBankRef(2, 250000); fb.init_bank(k, v)
    t0 = time.perf_counter()
    s, _ = O.segment(sd, frames[1:2], fb)
    k, v = O.memorize(sd, frames[1:2], torch.softmax(s, 1))
    fb.update(k, v, 1)
    print(n, 'threads: %.2f s/frame' % (time.perf_counter() - t0), flush=True)
