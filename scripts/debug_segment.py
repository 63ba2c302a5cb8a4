"""Stage-by-stage comparison of the HIP segment path against the oracle (debug aid, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.nn import functional as F
import vfloodnet_amd
from vfloodnet_amd import AFB_URR, FeatureBank
from tools import synth
from oracle import afb_urr_ref as O

H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 96, int(sys.argv[2]) if len(sys.argv) > 2 else 160
gpu = torch.device('cuda', 0)
sd = synth.make_state_dict(20200212)
model = AFB_URR(gpu, update_bank=True).to(gpu).eval()
model.load_state_dict(sd)
frames, m0 = synth.clip(1, 2, H, W)
oh = synth.onehot(m0).unsqueeze(0)
k_ref, v_ref = O.memorize(sd, frames[0:1], oh)
fb_ref = O.FeatureBankRef(2, 250000); fb_ref.init_bank(k_ref, v_ref)
k, v = model.memorize(frames[0:1].to(gpu), oh.to(gpu))
fb = FeatureBank(2, 250000, gpu); fb.init_bank(k, v)
score, _ = model.segment(frames[1:2].to(gpu), fb)
p = model.engine().plan(H, W, 2)

def nchw(x): return x.permute(0, 3, 1, 2).cpu()
def rep(name, a, b):
    d = (a - b).abs()
    i = d.flatten().argmax().item()
    print(f'{name:12s} max|d| {d.max().item():.3e} at {tuple(int(x) for x in torch.unravel_index(torch.tensor(i), d.shape))} ref max {b.abs().max().item():.3e}')

[fr], pad = O.pad_divide_by([frames[1:2]], 16, (H, W))
r4, r3, r2, r1 = O.encoder_q(sd, fr)
rep('r1', nchw(p.q['r1']), r1); rep('r2', nchw(p.q['res2']['out']), r2)
rep('r3', nchw(p.q['res3']['out']), r3); rep('r4', nchw(p.q['res4']['out']), r4)
k4, v4 = O.keyval(sd, r4)
rep('k4', p.kv_q[0, :, :128].t().cpu(), k4[0]); rep('v4', p.kv_q[0, :, 128:].t().cpu(), v4[0])
res = O.matcher(fb_ref, k4, v4, True)
gh, gw = r4.shape[2:]
res = res.reshape(2, 1024, gh, gw)
rep('dec_in', nchw(p.dec_in), res[:, :512])
r3e, r2e, r1e = r3.expand(2, -1, -1, -1), r2.expand(2, -1, -1, -1), r1.expand(2, -1, -1, -1)
out, parts = O.decoder(sd, res, r3e, r2e, r1e, (1, 2, r1.shape[2], r1.shape[3]), return_parts=True)
D = 'decoder'
pp0 = O._resblock(sd, D + '.ResMM', O._conv3(sd, D + '.convFM', res))
rep('d16', nchw(p.d16[2]), pp0)
pp1 = O._refine(sd, D + '.RF3', r3e, pp0); rep('d8', nchw(p.d8[2]), pp1)
pp2 = O._refine(sd, D + '.RF2', r2e, pp1); rep('d4', nchw(p.d4[2]), pp2)
pr = O._conv3(sd, D + '.pred2', F.relu(pp2)); rep('pred2', nchw(p.pp), pr)
rep('p_up', nchw(p.p_up), parts['p_up'])
rep('rough', p.rough.cpu(), parts['rough'][:, 0])
rep('unc', p.unc.cpu(), parts['unc'][0, 0])
rep('r1_local', nchw(p.lm), parts['r1_local'])
rep('conf', p.conf.cpu(), parts['r1_conf'][:, 0])
qraw = parts['q'] / parts['r1_conf']
rep('q*conf', nchw(p.qq) * p.conf.cpu().unsqueeze(1), parts['q'])
score_ref, _ = O.segment(sd, frames[1:2], fb_ref)
rep('score', score.cpu(), score_ref)
