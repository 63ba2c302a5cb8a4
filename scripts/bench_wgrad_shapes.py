"""Per-shape time of the weight-gradient kernel inside one training step (events around every ops.conv_wgrad call, synchronised:
each launch alone on the device) and a sweep of the pixel-split factor on the live operands.
usage: bench_wgrad_shapes.py [T H W obj_n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, ops, backward, train as T
from tools import synth
a = [int(x) for x in sys.argv[1:]]
Tn, H, W, K = (a + [6, 400, 400, 2][len(a):])[:4]
dev = torch.device('cuda', 0)
model = AFB_URR(dev, update_bank=False).to(dev); model.load_state_dict(synth.make_state_dict(20200212)); model.train()
frames, m0 = synth.clip(3, Tn, H, W)
lab = torch.stack([torch.roll(m0.long(), (2 * t, 5 * t), (0, 1)) for t in range(Tn)], 0)
masks = torch.nn.functional.one_hot(lab, K).permute(0, 3, 1, 2).float().to(dev)
frames = frames.to(dev)
T.forward_backward(model, frames, masks, 0.5)
orig = ops.conv_wgrad
rows = {}


def timed(x, gy, k, stride, pad, **kw):
    N, Hh, Ww = (kw.get('N') or x.shape[0]), (kw.get('H') or x.shape[1]), (kw.get('W') or x.shape[2])
    cin = kw.get('cin') or x.shape[-1]
    cout = kw.get('cout') or gy.shape[-1]
    M = N * gy.shape[1] * gy.shape[2]
    key = (M, cout, cin, k, stride, bool(kw.get('relu')), kw.get('batch', 1))
    if key not in rows:
        # sweep the split on a scratch output (the real launch follows)
        sweep = {}
        kw2 = dict(kw); kw2['out'] = torch.empty(kw.get('batch', 1) * cout, k * k * cin, device=dev); kw2['accumulate'] = False
        for ks in (None, 1, 2, 4, 8, 16, 32, 64):
            if ks is not None and ks > max(1, M // 64):
                continue
            kw2['ksplit'] = ks
            orig(x, gy, k, stride, pad, **kw2)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                orig(x, gy, k, stride, pad, **kw2)
            e1.record(); torch.cuda.synchronize()
            sweep[ks] = e0.elapsed_time(e1) / 5 * 1e3
        rows[key] = [0, sweep]
    rows[key][0] += 1
    return orig(x, gy, k, stride, pad, **kw)


ops.conv_wgrad = timed
T.forward_backward(model, frames, masks, 0.5)
torch.cuda.synchronize()
ops.conv_wgrad = orig
tot = 0.0
print('     M  cout   cin k s relu calls | default us (TF/s) | best split us | sweep')
for key in sorted(rows):
    M, cout, cin, k, s, relu, nb = key
    calls, sweep = rows[key]
    fl = 2.0 * M * cout * cin * k * k * nb
    d = sweep[None]
    best = min((v, ks) for ks, v in sweep.items() if ks is not None)
    tot += calls * d
    print(f'{M:6d} {cout:5d} {cin:5d} {k} {s} {int(relu)} x{nb:<2d} {calls:5d} | {d:8.1f} ({fl / d / 1e6:6.1f}) | {best[0]:8.1f} @{best[1]:<3d}| ' +
          ' '.join(f'{ks}:{v:.0f}' for ks, v in sweep.items() if ks is not None))
print(f'total at the default split: {tot / 1e3:.2f} ms per step')
