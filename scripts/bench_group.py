"""Round 6: frames between two memorize calls as ONE batched pass (ClipRunner.launch_group / AFB_URR.segment_group) against the
frame-by-frame loop, on BASELINE config C3's shape (720p clip, key frame every 5th, bf16 by default): frames/s of both loops
(alternating, same process) and the agreement of their labels.  Usage: bench_group.py [precision] [mem_every] [T] [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR
from vfloodnet_amd.video_seg import ClipRunner
from tools import synth
dev = torch.device('cuda', 0)
OVERLAP = os.environ.get('GROUP_OVERLAP', '1') == '1'
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
H0, W0 = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (720, 1280)
model = AFB_URR(dev, update_bank=True, precision=prec).to(dev).eval()
ck = os.environ.get('VFN_CKPT')
if ck:
    model.load_state_dict(torch.load(ck, map_location='cpu')['model'], strict=False)
else:
    model.load_state_dict(synth.make_state_dict(20200212), strict=True)
frames, m0 = synth.clip(3, T, H0, W0) if os.environ.get('GROUP_CLIP', 'easy') == 'easy' else synth.clip_hard(3, T, H0, W0)
frames = frames.to(dev)
onehot = synth.onehot(m0).unsqueeze(0).to(dev)


def seq(capture):
    r = ClipRunner(model, 2, 250000, mem_every=n, postprocess=True, capture_graphs=capture)
    r.start(frames[0:1], onehot)
    labs = torch.empty(T, H0, W0, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(1, T):
        r.launch(frames[t:t + 1], next_frames=[frames[u:u + 1] for u in range(t + 1, min(T, t + 4))])
        labs[t].copy_(r.label_device(), non_blocking=True)
        if len(r._pending) == 2:
            r.collect()
    while r._pending:
        r.collect()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return (T - 1) / dt, labs, r.size_log


def grp(capture):
    r = ClipRunner(model, 2, 250000, mem_every=n, postprocess=True, capture_graphs=capture, autotune=True)
    r.group_capture = n
    r.start(frames[0:1], onehot)
    labs = torch.empty(T, H0, W0, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = 1
    while t < T:
        g = min(n - (t - 1) % n, T - t)
        g2 = min(n, T - t - g)
        r.launch_group([frames[u:u + 1] for u in range(t, t + g)],
                       next_frames=[frames[u:u + 1] for u in range(t + g, t + g + g2)] if OVERLAP and g2 > 0 else None)
        for i, l in enumerate(r.group_labels_device()):
            labs[t + i].copy_(l, non_blocking=True)
        t += g
        if len(r._gpending) == 2:
            r.collect_group()
    while r._gpending:
        r.collect_group()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return (T - 1) / dt, labs, r.size_log


def miou(a, b):
    v = []
    for c in (0, 1):
        i = ((a == c) & (b == c)).flatten(1).sum(1).float()
        u = ((a == c) | (b == c)).flatten(1).sum(1).float()
        v.append(torch.where(u > 0, i / u.clamp(min=1), torch.ones_like(u)))
    return (v[0] + v[1]) / 2


seq(False); grp(False)                      # warm-up: plans, autotuned choices, allocator
res = {}
for rep in range(3):
    for name, fn in (('frame by frame', seq), ('grouped', grp)):
        fps, labs, sizes = fn(True)
        res.setdefault(name, []).append(fps)
        res[name + '/labs'], res[name + '/sizes'] = labs, sizes
print(f'{prec}, {T}-frame {H0}x{W0} clip ({os.environ.get("GROUP_CLIP", "easy")} frames, weights: {os.path.basename(ck) if ck else "synthetic"}), key frame every {n}th:')
for name in ('frame by frame', 'grouped'):
    print(f'  {name:15s} ' + ' / '.join(f'{x:.1f}' for x in res[name]) + ' frames/s')
m = miou(res['frame by frame/labs'][1:], res['grouped/labs'][1:])
same = (res['frame by frame/labs'][1:] == res['grouped/labs'][1:]).flatten(1).all(1).float().mean()
print(f'  labels grouped vs frame by frame: mIoU min {float(m.min()):.5f} mean {float(m.mean()):.5f}; identical frames {float(same):.3f}')
if os.environ.get('GROUP_DETAIL'):
    first = next((i for i, (x, y) in enumerate(zip(res['frame by frame/sizes'], res['grouped/sizes'])) if x != y), None)
    print(f'  first frame whose bank sizes differ: {first}; frames with mIoU < 0.999: ' + ' '.join(f'{i + 1}:{float(v):.4f}' for i, v in enumerate(m) if v < 0.999)[:600])
    f2, l2, s2 = grp(True)
    same_g = bool((l2[1:] == res['grouped/labs'][1:]).all())
    f3, l3, s3 = seq(True)
    same_s = bool((l3[1:] == res['frame by frame/labs'][1:]).all())
    print(f'  a second grouped run equals the first bit for bit: {same_g} (sizes {s2 == res["grouped/sizes"]}); a second frame-by-frame run equals the first: {same_s} (sizes {s3 == res["frame by frame/sizes"]})')
sa, sb = res['frame by frame/sizes'], res['grouped/sizes']
print(f'  bank sizes at the end: {sa[-1]} vs {sb[-1]}; size vectors equal: {sa == sb}')
