"""Where the host time of one training step goes (cProfile of train.train_step after two warm-up steps)."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, train as T
from tools import synth
Tn, H, W, K = 6, 400, 400, 2
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(20200212)
model = AFB_URR(dev, update_bank=False).to(dev); model.load_state_dict(sd); model.train()
frames, m0 = synth.clip(3, Tn, H, W)
lab = torch.stack([torch.roll(m0.long(), (2 * t, 5 * t), (0, 1)) for t in range(Tn)], 0)
masks = torch.nn.functional.one_hot(lab, K).permute(0, 3, 1, 2).float()
frames, masks = frames.to(dev), masks.to(dev)
opt = T.AdamW(model.named_parameters(), lr=1e-5)
for s in range(2):
    T.train_step(model, opt, frames, masks, 0.5)
torch.cuda.synchronize()
from vfloodnet_amd import backward as Bk
marks = {}
orig_fin = Bk.ModelBackward.finish_memorize
def fin(self, *a, **k):
    r = orig_fin(self, *a, **k)
    marks['enqueued'] = time.perf_counter()          # everything of forward_backward is in the queues here (before the .tolist() sync)
    return r
Bk.ModelBackward.finish_memorize = fin
for rep in range(3):
    t0 = time.perf_counter()
    T.train_step(model, opt, frames, masks, 0.5)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f'host has enqueued forward+backward after {1e3 * (marks["enqueued"] - t0):.1f} ms, step returns after {1e3 * (t1 - t0):.1f} ms')
Bk.ModelBackward.finish_memorize = orig_fin
if os.environ.get('NO_CPROFILE'):
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
T.train_step(model, opt, frames, masks, 0.5)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(24)
st.sort_stats('cumulative').print_stats(30)
