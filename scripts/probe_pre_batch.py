"""The frame-only side batched over n frames (QuerySet(nq=n).pre[n]), per frame, against the look-ahead's two-frame pass: C2 shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR
from tools import synth
dev = torch.device('cuda', 0)
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 854)
model = AFB_URR(dev, update_bank=True, precision=prec).to(dev).eval()
model.load_state_dict(synth.make_state_dict(20200212), strict=True)
eng = model.engine()
p = eng.plan(H, W, 2)
for n in (3, 4, 6):
    p.batch_set(n)
eng.autotune(H, W, 2, only_missing=True)


def timed(fn, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for n, lst in [(1, p.qsets[0].pre[1]), (2, p.qsets[0].pre[2])] + [(n, p.batch_set(n).pre[n]) for n in (3, 4, 6)]:
    for _ in range(3):
        p.graphs.run(lst)
    t = timed(lambda: p.graphs.run(lst))
    print(f'{prec} {H}x{W}: frame-only side over {n} frame(s): {t:8.1f} us = {t / n:7.1f} us per frame ({len(lst)} launches)', flush=True)
