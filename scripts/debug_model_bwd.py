import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, FeatureBank, ops, _lib, backward
from tools import synth
gpu = torch.device('cuda', 0)
orig_check = _lib.check
calls = []
def chk(status, what):
    torch.cuda.synchronize()
    calls.append(what)
    if status != 0:
        print('FAILED at', what, 'after', calls[-6:])
    orig_check(status, what)
for mod in (_lib, ops, backward):
    mod.check = chk
H, W, K = 96, 160, 2
sd = synth.make_state_dict(20200212)
model = AFB_URR(gpu, update_bank=False).to(gpu); model.load_state_dict(sd); model.train()
frames, m0 = synth.clip(6, 2, H, W); oh = synth.onehot(m0).unsqueeze(0)
k, v = model.memorize(frames[0:1].to(gpu), oh.to(gpu))
fb = FeatureBank(K, 250000, gpu); fb.init_bank(k, v)
scores, unc = model.segment(frames[1:2].to(gpu), fb)
label = torch.randint(0, K, (1, H, W))
stats, ds = ops.segment_loss(scores.contiguous(), label.to(gpu), 0.5)
mb = backward.ModelBackward(model.engine())
orig_launch = ops.conv2d_launch
def launch(d, cfg, mode=0):
    torch.cuda.synchronize()
    try:
        orig_launch(d, cfg, mode)
        torch.cuda.synchronize()
    except Exception as e:
        print('conv launch failed', e, dict(M=d.M, Cout=d.Cout, Cin=d.Cin, KH=d.KH, ks=d.ksplit, cfg=cfg, out_ld=d.out_ld, cout_pad=d.cout_pad))
        raise
ops.conv2d_launch = launch
g_bk, g_bv = mb.segment_sample(fb, ds[0])
print('segment_sample ok')
mb.finish_memorize(frames[0:1].to(gpu), oh.to(gpu), g_bk, g_bv)
print('finish ok', len(mb.grads))
