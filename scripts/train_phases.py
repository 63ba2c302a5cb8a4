"""Where the dependent chain of the training step goes (VFN_SIDE_DROP=1: weight gradients dropped, so the times are the main
stream's): CUDA events around the phases of train._forward_backward."""
import sys, os
os.environ.setdefault('VFN_SIDE_DROP', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, train as T, engine as E, backward as Bk, ops
from tools import synth
Tn, H, W, K = 6, 400, 400, 2
dev = torch.device('cuda', 0)
model = AFB_URR(dev, update_bank=False).to(dev); model.load_state_dict(synth.make_state_dict(20200212)); model.train()
frames, m0 = synth.clip(3, Tn, H, W)
lab = torch.stack([torch.roll(m0.long(), (2 * t, 5 * t), (0, 1)) for t in range(Tn)], 0)
masks = torch.nn.functional.one_hot(lab, K).permute(0, 3, 1, 2).float().to(dev)
frames = frames.to(dev)
opt = T.AdamW(model.named_parameters(), lr=1e-5)
for _ in range(3):
    T.train_step(model, opt, frames, masks, 0.5)
marks = []


def timed(owner, name, tag):
    orig = getattr(owner, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a, **k); e1.record()
        marks.append((tag, e0, e1))
        return r
    setattr(owner, name, f)


timed(E.Engine, 'memorize', 'memorize forward')
timed(E.Engine, 'query_batch', 'query encoder forward (5 frames)')
timed(E.Engine, 'segment', 'memory read + decoder forward (per sample)')
timed(ops, 'segment_loss', 'loss + dloss (per sample)')
timed(Bk.ModelBackward, 'segment_sample', 'decoder + memory read backward (per sample)')
timed(Bk.ModelBackward, 'finish_query', 'query encoder backward (5 frames)')
timed(Bk.ModelBackward, 'finish_memorize', 'memory encoder backward')
timed(T.AdamW, 'set_grads', 'gradients -> flat buffer')
timed(T.AdamW, 'step', 'AdamW')
timed(E.Engine, 'refresh', 'refresh of derived tensors')
tot = {}
for rep in range(3):
    marks.clear()
    T.train_step(model, opt, frames, masks, 0.5)
    torch.cuda.synchronize()
    for tag, e0, e1 in marks:
        tot.setdefault(tag, []).append(e0.elapsed_time(e1))
n = 3
s = 0.0
for tag, v in tot.items():
    ms = sum(v) / n
    s += ms
    print(f'{ms:6.2f} ms  ({len(v) // n} calls)  {tag}')
print(f'{s:6.2f} ms  sum')
