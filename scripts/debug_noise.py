"""fp32 oracle and HIP path against an fp64 run of the oracle: is the HIP error at the fp32 noise floor?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.nn import functional as F
import vfloodnet_amd
from tools import synth
from oracle import afb_urr_ref as O

H, W = 96, 160
sd = synth.make_state_dict(20200212)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
frames, m0 = synth.clip(1, 2, H, W)
oh = synth.onehot(m0).unsqueeze(0)
def run(sd_, dt):
    k, v = O.memorize(sd_, frames[0:1].to(dt), oh.to(dt))
    fb = O.FeatureBankRef(2, 250000); fb.init_bank(k, v)
    fb.info = [i.to(dt) for i in fb.info]
    s, _ = O.segment(sd_, frames[1:2].to(dt), fb)
    return k, v, s
k64, v64, s64 = run(sd64, torch.float64)
k32, v32, s32 = run(sd, torch.float32)
p64 = torch.sigmoid(s64)
def rep(n, a, b): print(f'{n:22s} max|d| {(a.double()-b).abs().max().item():.3e}')
rep('oracle32 key', k32[0], k64[0]); rep('oracle32 logit(|s|<8)', s32[s64.abs() < 8], s64[s64.abs() < 8]); rep('oracle32 prob', torch.sigmoid(s32), p64)
if torch.cuda.is_available():
    from vfloodnet_amd import AFB_URR, FeatureBank
    gpu = torch.device('cuda', 0)
    model = AFB_URR(gpu, update_bank=True).to(gpu).eval(); model.load_state_dict(sd)
    k, v = model.memorize(frames[0:1].to(gpu), oh.to(gpu))
    fb = FeatureBank(2, 250000, gpu); fb.init_bank(k, v)
    s, _ = model.segment(frames[1:2].to(gpu), fb)
    s = s.cpu()
    rep('hip key', k[0].cpu(), k64[0]); rep('hip logit(|s|<8)', s[s64.abs() < 8], s64[s64.abs() < 8]); rep('hip prob', torch.sigmoid(s), p64)
