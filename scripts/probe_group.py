"""Where a group of G frames spends its time (events around the pieces of ClipRunner.launch_group), C3 shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, ops
from vfloodnet_amd.video_seg import ClipRunner
from vfloodnet_amd.engine import DV
from tools import synth
dev = torch.device('cuda', 0)
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
G = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H0, W0 = 720, 1280
model = AFB_URR(dev, update_bank=True, precision=prec).to(dev).eval()
model.load_state_dict(synth.make_state_dict(20200212), strict=True)
frames, m0 = synth.clip(3, 6 * G + 1, H0, W0)
frames = frames.to(dev)
r = ClipRunner(model, 2, 250000, mem_every=G, postprocess=True, capture_graphs=True, autotune=True)
r.group_capture = G
r.start(frames[0:1], synth.onehot(m0).unsqueeze(0).to(dev))
t = 1
for _ in range(4):
    r.step_group([frames[u:u + 1] for u in range(t, t + G)]); t += G
eng = model.engine()
nets = [r._net_cached(frames[u:u + 1]) for u in range(t, t + G)]
p = eng.plan(nets[0].shape[2], nets[0].shape[3], 2)
qs = p.batch_set(G); b = qs.dec_batch()
K = 2


def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


print(f'{prec}, G = {G}, bank {r.bank_sizes()}')
print(f'resize {G} frames to network size: {timed(lambda: [r._net_frame(frames[u:u + 1]) for u in range(t, t + G)]):8.1f} us')
print(f'frame copies:                     {timed(lambda: [qs.frames[i].copy_(nets[i][0]) for i in range(G)]):8.1f} us')
print(f'pre[{G}] ({len(qs.pre[G])} launches):          {timed(lambda: p.graphs.run(qs.pre[G])):8.1f} us')
print(f'pre[{G}] eager:                      {timed(lambda: [l() for l in qs.pre[G]]):8.1f} us')
print(f'memory read ({G} x HW queries):     {timed(lambda: eng._memory_read(b.mr, r.fb, False, qs.kv_q[0:G])):8.1f} us')
print(f'read-out regroup copy:            {timed(lambda: b.dec_in.view(G, K, p.HW, DV).copy_(b.mr.dec_in.view(K, G, p.HW, DV).permute(1, 0, 2, 3))):8.1f} us')
print(f'post ({len(b.post)} launches):             {timed(lambda: p.graphs.run(b.post)):8.1f} us')
print(f'post eager:                       {timed(lambda: [l() for l in b.post]):8.1f} us')
prob = torch.empty_like(b.score)
print(f'softmax x {G}:                      {timed(lambda: [ops.softmax_objects(b.score[g:g + 1], out=prob[g:g + 1]) for g in range(G)]):8.1f} us')
print(f'memorize list ({len(p.mem)} launches):      {timed(lambda: p.graphs.run(p.mem)):8.1f} us')
lab = torch.empty(H0, W0, dtype=torch.uint8, device=dev); post = torch.empty_like(lab)
def tail():
    for g in range(G):
        ops.resize_argmax(prob[g:g + 1], H0, W0, out=lab)
        ops.postprocess_pred_device(lab, post, r._ccl_scratch)
print(f'argmax + CCL x {G}:                 {timed(tail):8.1f} us')
qs1 = p.qsets[0]
print(f'(frame by frame: pre[2] {timed(lambda: p.graphs.run(qs1.pre[2])):.1f} us, pre[1] {timed(lambda: p.graphs.run(qs1.pre[1])):.1f} us, post {timed(lambda: p.graphs.run(qs1.post[0])):.1f} us, '
      f'memory read {timed(lambda: eng._memory_read(p, r.fb, False, qs1.kv_q[0:1])):.1f} us)')
from collections import Counter
import itertools
def by_name(lst):
    out = []
    for l in lst:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); l(); e1.record(); torch.cuda.synchronize()
        out.append((l.name, e0.elapsed_time(e1) * 1e3))
    return out
for nm, lst in (('pre', qs.pre[G]), ('post', b.post)):
    rows = by_name(lst)
    print(nm, 'sum of launches alone', round(sum(x for _, x in rows), 1))
    for name, us in sorted(rows, key=lambda x: -x[1])[:14]:
        print(f'    {name:50s} {us:8.1f} us')
