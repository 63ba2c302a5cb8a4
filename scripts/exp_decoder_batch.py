"""What would batching the training step's decoder over the five samples buy?  The bank-dependent decoder list at 400 x 400 with
obj_n = 2 (one sample: what the step runs five times) against obj_n = 10 (the same convolutions with five times the images),
launch by launch, kernels alone on the device."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR
from tools import synth
dev = torch.device('cuda', 0)
model = AFB_URR(dev, update_bank=False).to(dev); model.load_state_dict(synth.make_state_dict(20200212)); model.eval()
eng = model.engine()
res = {}
for K in (2, 8):
    p = eng.plan(400, 400, K)
    p.dec_in.normal_()
    qs = p.qsets[0]
    for t in [qs.fm_q, qs.lq, qs.q['r1']] + qs.s8 + qs.s4:
        t.normal_()
    lst = qs.post[0]
    for _ in range(3):
        for l in lst:
            l()
    torch.cuda.synchronize()
    rows = []
    for l in lst:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            l()
        e1.record(); torch.cuda.synchronize()
        rows.append((l.name, e0.elapsed_time(e1) / 5 * 1e3))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        for l in lst:
            l()
    e1.record(); torch.cuda.synchronize()
    res[K] = (rows, e0.elapsed_time(e1) / 10)
for (n2, t2), (n10, t10) in zip(res[2][0], res[8][0]):
    print(f'{n2:36s} {t2:8.1f} us x4 = {4 * t2:8.1f}   batched {t10:8.1f} us   ratio {t10 / (4 * t2):.2f}')
print(f'list: obj_n=2 {res[2][1]:.3f} ms x4 = {4 * res[2][1]:.3f} ms; obj_n=8 {res[8][1]:.3f} ms')
