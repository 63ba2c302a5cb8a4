"""Debug: which torch-level copies / fills / adds one steady-state training step issues (each is its own small launch)."""
import sys, os, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, train as T
from tools import synth
Tn, H, W, K = 6, 400, 400, 2
dev = torch.device('cuda', 0)
model = AFB_URR(dev, update_bank=False).to(dev); model.load_state_dict(synth.make_state_dict(20200212)); model.train()
frames, m0 = synth.clip(3, Tn, H, W)
lab = torch.stack([torch.roll(m0.long(), (2 * t, 5 * t), (0, 1)) for t in range(Tn)], 0)
masks = torch.nn.functional.one_hot(lab, K).permute(0, 3, 1, 2).float().to(dev)
frames = frames.to(dev)
opt = T.AdamW(model.named_parameters(), lr=1e-5)
for _ in range(2):
    T.train_step(model, opt, frames, masks, 0.5)
log = collections.Counter()
def where():
    st = [s for s in traceback.extract_stack(limit=8)[:-2] if 'v-floodnet_amd' in s.filename or 'vfloodnet_amd' in s.filename]
    return ' <- '.join('%s:%d' % (os.path.basename(s.filename), s.lineno) for s in st[-2:])
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        if self.is_cuda:
            log[(name, where())] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
for n in ('copy_', 'zero_', 'fill_', 'clone', 'contiguous', '__iadd__', '__add__', '__mul__', '__imul__', '__truediv__', '__sub__', 't', 'reshape'):
    wrap(n)
for fn in ('zeros', 'cat', 'stack', 'zeros_like', 'empty_like', 'argmax'):
    o = getattr(torch, fn)
    def g(*a, _o=o, _n=fn, **k):
        log[(_n, where())] += 1
        return _o(*a, **k)
    setattr(torch, fn, g)
T.train_step(model, opt, frames, masks, 0.5)
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -kv[1])[:60]:
    print(v, k)
