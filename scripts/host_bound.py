"""Is the loop host-bound?  Per-frame wall vs host time spent enqueueing (ClipRunner.launch) for a workload / precision."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR
from vfloodnet_amd.video_seg import ClipRunner
from tools import synth
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
H0, W0, mem_every = (720, 1280, 5) if (len(sys.argv) > 2 and sys.argv[2] == 'C3') else (480, 854, 1)
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(20200212)
model = AFB_URR(dev, update_bank=True, precision=prec).to(dev).eval(); model.load_state_dict(sd)
frames, m0 = synth.clip(1, 60, H0, W0); frames = frames.to(dev)
r = ClipRunner(model, 2, 250000, size=480, mem_every=mem_every, postprocess=True)
r.start(frames[0:1], synth.onehot(m0).unsqueeze(0).to(dev))
T = frames.shape[0]
host = 0.0
torch.cuda.synchronize(); t0 = time.perf_counter()
for rep in range(3):
    for t in range(1, T):
        h0 = time.perf_counter()
        r.launch(frames[t:t + 1], next_frames=[frames[u:u + 1] for u in range(t + 1, min(T, t + 4))])
        host += time.perf_counter() - h0
        if len(r._pending) == 2:
            r.collect()
    while r._pending:
        r.collect()
torch.cuda.synchronize(); t1 = time.perf_counter()
n = 3 * (T - 1)
print(f'{prec} {H0}x{W0}: wall {1e3 * (t1 - t0) / n:.3f} ms/frame, host enqueue {1e3 * host / n:.3f} ms/frame, graphs={os.environ.get("VFN_GRAPHS", "default")}')
