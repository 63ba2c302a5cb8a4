"""One-off validation beyond the 100-frame golden clip: the C2 clip continued into the regime where the bank sits at
its budget and LFU eviction fires on every update (from frame ~102 on), HIP path vs the CPU oracle (torch CPU, slow:
~1 frame/s).  usage: long_parity.py [frames]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR
from tools import synth
from vfloodnet_amd.video_seg import run_clip
from oracle import afb_urr_ref as O

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
gpu = torch.device('cuda', 0)
sd = synth.make_state_dict(20200212)
frames, m0 = synth.clip(1, T, 480, 854)
model = AFB_URR(gpu, update_bank=True).to(gpu).eval(); model.load_state_dict(sd)
out = run_clip(model, frames.to(gpu), m0)
torch.set_num_threads(16)
t0 = time.time()
ref = O.run_clip(sd, frames, m0)
print('oracle: %.1f s' % (time.time() - t0))

def miou(a, b):
    v = []
    for c in (0, 1):
        i = ((a == c) & (b == c)).sum().item(); u = ((a == c) | (b == c)).sum().item()
        v.append(1.0 if u == 0 else i / u)
    return sum(v) / 2
ious = [miou(out['labels'][t], ref['labels'][t]) for t in range(1, T)]
hs, rs = np.array(out['bank_sizes']), np.array(ref['bank_sizes'])
res = {'frames': T, 'miou_min': min(ious), 'miou_mean': float(np.mean(ious)), 'miou_min_after_100': min(ious[100:]) if T > 101 else None,
       'bank_size_max_abs_diff': int(np.abs(hs - rs).max()), 'final_bank_hip': hs[-1].tolist(), 'final_bank_oracle': rs[-1].tolist(),
       'replace_n_hip': out['fb'].replace_n.tolist(), 'replace_n_oracle': ref['fb'].replace_n.tolist(),
       'peak_n_hip': out['fb'].peak_n.tolist(), 'peak_n_oracle': ref['fb'].peak_n.tolist()}
print(json.dumps(res))
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/r01_long_parity_%d.json' % T, 'w'), indent=1)
