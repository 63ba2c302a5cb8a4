"""Frame-by-frame comparison of the HIP loop against the oracle loop (debug aid, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.nn import functional as F
import vfloodnet_amd
from vfloodnet_amd import AFB_URR, FeatureBank, ops
from tools import synth
from vfloodnet_amd.video_seg import ClipRunner
from oracle import afb_urr_ref as O

gpu = torch.device('cuda', 0)
sd = synth.make_state_dict(20200212)
model = AFB_URR(gpu, update_bank=True).to(gpu).eval()
model.load_state_dict(sd)
frames, m0 = synth.clip(3, 6, 64, 96)
size = 128
m = (m0 > 0).to(torch.uint8)
onehot = torch.stack([1 - m, m], 0).unsqueeze(0)
fb_ref = O.FeatureBankRef(2, 250000)
f0 = O.tf_resize(frames[0:1], size, 'bicubic'); mm0 = O.tf_resize(onehot, size, 'nearest')
k, v = O.memorize(sd, f0, mm0); fb_ref.init_bank(k, v)
run = ClipRunner(model, 2, 250000, size=size)
run.start(frames[0:1].to(gpu), onehot.to(gpu))
def rep(name, a, b):
    d = (a - b).abs(); print(f'   {name:10s} max|d| {d.max().item():.3e}  ref max {b.abs().max().item():.3e}')
for i in range(2):
    rep(f'bank k{i}', run.fb.keys[i].cpu(), fb_ref.keys[i])
for t in range(1, 6):
    print('frame', t)
    fr = O.tf_resize(frames[t:t+1], size, 'bicubic')
    score_ref, _ = O.segment(sd, fr, fb_ref)
    pm_ref = F.softmax(score_ref, dim=1)
    # HIP, step by step
    f = run._net_frame(frames[t:t+1].to(gpu))
    rep('frame', f.cpu(), fr)
    score, _ = model.segment(f, run.fb)
    pm = ops.softmax_objects(score)
    rep('prob', pm.cpu(), pm_ref)
    for i in range(2):
        rep(f'info{i}', run.fb.info[i].cpu(), fb_ref.info[i])
    # memorize with the ORACLE's soft mask on both sides to isolate
    k_ref, v_ref = O.memorize(sd, fr, pm_ref)
    kh, vh = model.memorize(f, pm_ref.to(gpu))
    rep('mem k', kh[1].cpu(), k_ref[1]); rep('mem v', vh[0].cpu(), v_ref[0])
    kh, vh = model.memorize(f, pm)
    fb_ref.update(k_ref, v_ref, t)
    run.fb.update(kh, vh, t)
    for i in range(2):
        print('   sizes', run.fb.keys[i].shape, fb_ref.keys[i].shape)
        n = min(run.fb.keys[i].shape[1], fb_ref.keys[i].shape[1])
        rep(f'bank k{i}', run.fb.keys[i].cpu()[:, :n], fb_ref.keys[i][:, :n])
        rep(f'bank v{i}', run.fb.values[i].cpu()[:, :n], fb_ref.values[i][:, :n])
