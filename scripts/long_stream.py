"""BASELINE config C5 shape (fp32 here): a long 1920x1080 stream at reference semantics (bicubic resize to 853x480
on the device), bank budget sized so that nothing is evicted -> frames/s as a function of the bank size.
Frames are produced on the GPU by rolling frame 0 (a 2000-frame 1080p clip would be 50 GB of input)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR
from tools import synth
from vfloodnet_amd.video_seg import ClipRunner

T = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(20200212)
model = AFB_URR(dev, update_bank=True, precision=os.environ.get('VFN_PRECISION', 'fp32')).to(dev).eval(); model.load_state_dict(sd)
f0, m0 = synth.frame0(1, 1080, 1920)
f0 = f0.to(dev)
onehot = synth.onehot(m0).unsqueeze(0).to(dev)
budget = 2 * int(1.25 * 2 * T * 1620) + 4         # class_budget = 0.8 * budget/2 >= T*HW: no eviction
runner = ClipRunner(model, 2, budget)
runner.start(f0.unsqueeze(0), onehot)
curve = []
every = 25 if T <= 500 else 100
torch.cuda.synchronize(); t_prev = time.perf_counter(); n_prev = 0; t_start = t_prev
for t in range(1, T + 1):
    fr = torch.roll(f0, shifts=(2 * t, 5 * t), dims=(1, 2)).unsqueeze(0)
    runner.step(fr, want_label=False)
    if t % every == 0:
        torch.cuda.synchronize(); now = time.perf_counter()
        curve.append({'frame': t, 'bank_entries_per_object': max(runner.bank_sizes()), 'ms_per_frame': round(1e3 * (now - t_prev) / (t - n_prev), 3),
                      'hbm_allocated_gb': round(torch.cuda.memory_allocated() / 1e9, 2)})
        print(curve[-1], flush=True)
        t_prev, n_prev = now, t
total = time.perf_counter() - t_start
print('whole stream: %d frames in %.1f s = %.2f frames/s' % (T, total, T / total))
os.makedirs('gpurun_out', exist_ok=True)
json.dump({'config': f'C5 shape: {T} frames 1920x1080 -> 853x480 (bicubic on device), {os.environ.get("VFN_PRECISION", "fp32")}, no eviction (budget {budget})',
           'frames': T, 'seconds': round(total, 2), 'frames_per_s': round(T / total, 3), 'curve': curve},
          open('gpurun_out/r01_c5_long_stream_%s%s.json' % (os.environ.get('VFN_PRECISION', 'fp32'), '' if T == 400 else '_%d' % T), 'w'), indent=1)
