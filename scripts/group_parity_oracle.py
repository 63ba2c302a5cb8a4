"""The grouped loop (ClipRunner.launch_group, BASELINE config C3: 720p frames, key frame every 5th) against the CPU oracle over a
stretch of the clip -- labels BEFORE post-processing, bank sizes -- in f32 and bf16x3 on the synthetic weights, and beside it the
frame-by-frame HIP loop against the same oracle run.  usage: group_parity_oracle.py [frames]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR
from vfloodnet_amd.video_seg import ClipRunner
from tools import synth
from oracle import afb_urr_ref as O

T = int(sys.argv[1]) if len(sys.argv) > 1 else 41
n = 5
gpu = torch.device('cuda', 0)
sd = synth.make_state_dict(20200212)
frames, m0 = synth.clip(1, T, 720, 1280)
torch.set_num_threads(16)
t0 = time.time()
ref = O.run_clip(sd, frames, m0, mem_every=n)
print('oracle: %.1f s for %d frames' % (time.time() - t0, T), flush=True)
fr = frames.to(gpu)
onehot = synth.onehot(m0).unsqueeze(0).to(gpu)


def miou(a, b):
    v = []
    for c in (0, 1):
        i = ((a == c) & (b == c)).sum().item(); u = ((a == c) | (b == c)).sum().item()
        v.append(1.0 if u == 0 else i / u)
    return sum(v) / 2


def run(prec, grouped):
    model = AFB_URR(gpu, update_bank=True, precision=prec).to(gpu).eval(); model.load_state_dict(sd)
    r = ClipRunner(model, 2, 250000, mem_every=n)
    r.group_capture = n if grouped else 0
    r.start(fr[0:1], onehot)
    labs, t = [], 1
    while t < T:
        if grouped:
            g = min(n - (t - 1) % n, T - t)
            r.launch_group([fr[u:u + 1] for u in range(t, t + g)], next_frames=[fr[u:u + 1] for u in range(t + g, min(T, t + g + n))] or None)
            labs += [x.clone() for x in r.collect_group()]
            t += g
        else:
            labs.append(r.step(fr[t:t + 1], next_frames=[fr[u:u + 1] for u in range(t + 1, min(T, t + 4))]).clone())
            t += 1
    return labs, r.size_log


res = {}
for prec in ('fp32', 'bf16x3'):
    for grouped in (True, False):
        labs, sizes = run(prec, grouped)
        ious = [miou(labs[t - 1], ref['labels'][t]) for t in range(1, T)]
        hs, rs = np.array(sizes[1:]), np.array(ref['bank_sizes'])
        key = f'{prec} {"grouped" if grouped else "frame by frame"}'
        res[key] = {'miou_min': round(min(ious), 5), 'miou_mean': round(float(np.mean(ious)), 5), 'bank_size_max_abs_diff': int(np.abs(hs - rs).max()),
                    'final_bank': hs[-1].tolist(), 'final_bank_oracle': rs[-1].tolist()}
        print(key, json.dumps(res[key]), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump({'frames': T, 'mem_every': n, 'clip': '720x1280 synthetic, network at 480x853', 'vs': 'CPU oracle (torch f32)', 'runs': res},
          open('gpurun_out/r06_group_parity_oracle.json', 'w'), indent=1)
