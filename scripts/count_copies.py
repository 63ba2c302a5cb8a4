"""Debug: which torch-level copies / fills one steady-state ClipRunner frame issues (each is its own blit launch)."""
import sys, os, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import video_seg, AFB_URR
from tools import synth
dev = torch.device('cuda', 0)
model = AFB_URR(dev, update_bank=True).to(dev).eval(); model.load_state_dict(synth.make_state_dict(20200212))
frames, m0 = synth.clip(1, 8, 480, 854)
fr = frames.to(dev)
r = video_seg.ClipRunner(model)
r.start(fr[0:1], synth.onehot(m0).unsqueeze(0).to(dev))
for t in range(1, 4):
    r.launch(fr[t:t + 1], fr[t + 1:t + 2]); r.collect()
log = collections.Counter()
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        st = traceback.extract_stack(limit=4)[:-1]
        log[(name, tuple(self.shape), ' <- '.join('%s:%d' % (os.path.basename(s.filename), s.lineno) for s in st[-2:]))] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
for n in ('copy_', 'zero_', 'fill_', 'clone', 'contiguous', 'to', 'float'):
    wrap(n)
for fn in ('empty', 'zeros', 'cat', 'stack'):
    o = getattr(torch, fn)
    def g(*a, _o=o, _n=fn, **k):
        st = traceback.extract_stack(limit=3)[:-1]
        log[(_n, '', ' <- '.join('%s:%d' % (os.path.basename(s.filename), s.lineno) for s in st[-2:]))] += 1
        return _o(*a, **k)
    setattr(torch, fn, g)
r.launch(fr[4:5], fr[5:6]); r.collect()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print(v, k)
