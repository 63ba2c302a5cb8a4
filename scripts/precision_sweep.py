"""Per-layer precision sensitivity: which groups of layers tolerate plain bf16 operands on the C3 clip
(720x1280 -> 480p, every 5th frame memorised), measured as label mIoU against the f32 run of the same clip.

  1. every group alone in bf16, the rest bf16x3            -> how much each group costs
  2. every group alone in bf16x3, the rest bf16            -> how much each group buys back
  3. greedy: starting from all-bf16x3, move groups to bf16 in order of harmlessness while min mIoU stays >= 0.99
Writes gpurun_out/r04_precision_sweep.json.  usage: precision_sweep.py [C3|C5short]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfloodnet_amd
from vfloodnet_amd import AFB_URR
from vfloodnet_amd.video_seg import run_clip
from tools import synth

dev = torch.device('cuda', 0)
which = sys.argv[1] if len(sys.argv) > 1 else 'C3'
H, W, T, mem_every = (720, 1280, 100, 5) if which == 'C3' else (1080, 1920, 120, 1)
GROUPS = ['encoder_q', 'encoder_m', 'keyval', 'memread', 'bank_update', 'decoder.convFM', 'decoder.ResMM', 'decoder.RF3',
          'decoder.RF2', 'decoder.pred2', 'decoder.local']
sd = synth.make_state_dict(20200212)
frames, m0 = synth.clip_on_device(3 if which == 'C3' else 9, T, H, W, dev)
budget = 250000 if which == 'C3' else 2 * int(1.25 * 2 * (T + 2) * 1620) + 4


def miou(a, b):
    v = []
    for c in (0, 1):
        inter = ((a == c) & (b == c)).sum().item()
        union = ((a == c) | (b == c)).sum().item()
        v.append(1.0 if union == 0 else inter / union)
    return sum(v) / 2


def run(base, pmap):
    model = AFB_URR(dev, update_bank=True, precision=base).to(dev).eval()
    model.precision_map = dict(pmap)
    model.load_state_dict(sd, strict=True)
    run_clip(model, frames[:6], m0, budget=budget, size=480, mem_every=mem_every)       # warm (tables, plans)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_clip(model, frames, m0, budget=budget, size=480, mem_every=mem_every)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return out['labels'], (T - 1) / dt, out['bank_sizes'][-1]


ref, fps_ref, bank_ref = run('fp32', {})
print(f'fp32: {fps_ref:.1f} frames/s, bank {bank_ref}', flush=True)
res = {'workload': which, 'fp32_fps': fps_ref, 'runs': []}


def score(tag, base, pmap):
    lab, fps, bank = run(base, pmap)
    ious = [miou(lab[t], ref[t]) for t in range(1, T)]
    r = {'tag': tag, 'base': base, 'map': pmap, 'fps': round(fps, 1), 'miou_min': round(min(ious), 5), 'miou_mean': round(sum(ious) / len(ious), 5),
         'miou_first': round(ious[0], 5), 'bank': bank}
    res['runs'].append(r)
    print(r, flush=True)
    return r


score('all bf16x3', 'bf16x3', {})
score('all bf16', 'bf16', {})
cost = {}
for g in GROUPS:
    cost[g] = score(f'bf16x3 except {g} -> bf16', 'bf16x3', {g: 'bf16'})['miou_min']
for g in GROUPS:
    score(f'bf16 except {g} -> bf16x3', 'bf16', {g: 'bf16x3'})
# greedy
chosen = {}
for g in sorted(GROUPS, key=lambda g_: -cost[g_]):
    trial = dict(chosen, **{g: 'bf16'})
    r = score('greedy + ' + g, 'bf16x3', trial)
    if r['miou_min'] >= 0.99:
        chosen = trial
res['greedy_bf16_groups'] = sorted(chosen)
final = score('greedy result', 'bf16x3', chosen) if chosen else None
res['greedy_result'] = final
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open(f'gpurun_out/r04_precision_sweep_{which}.json', 'w'), indent=1)
print('greedy: groups in plain bf16 with min mIoU >= 0.99:', sorted(chosen))
