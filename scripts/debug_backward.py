import sys, os
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/tests') else os.getcwd())
import torch, torch.nn.functional as F
import vfloodnet_amd
from vfloodnet_amd import AFB_URR, FeatureBank
from vfloodnet_amd.backward import DecoderBackward
from tools import synth
gpu = torch.device('cuda', 0)
sd = synth.make_state_dict(20200212)
model = AFB_URR(gpu, update_bank=False).to(gpu).eval(); model.load_state_dict(sd)
H, W = 96, 160
frames, m0 = synth.clip(4, 2, H, W)
oh = synth.onehot(m0).unsqueeze(0)
k, v = model.memorize(frames[0:1].to(gpu), oh.to(gpu))
fb = FeatureBank(2, 250000, gpu); fb.init_bank(k, v)
model.segment(frames[1:2].to(gpu), fb)
eng = model.engine(); plan, qs, slot = eng.last_query
B = DecoderBackward(eng)
K, h2, w2 = 2, plan.h2, plan.w2
g = torch.randn(K, h2, w2, 32, device=gpu)
l2 = plan.l2
w2c = sd['decoder.local_ResMM.conv2.weight'].to(gpu)
w1c = sd['decoder.local_ResMM.conv1.weight'].to(gpu)
# check forward consistency: l2[1] == conv1(relu(l2[0])) + b ?
x = l2[0].permute(0, 3, 1, 2)
r_ref = F.conv2d(F.relu(x), w1c, sd['decoder.local_ResMM.conv1.bias'].to(gpu), padding=1).permute(0, 2, 3, 1)
print('forward l2[1] vs conv1(relu(l2[0])):', (l2[1] - r_ref).abs().max().item(), r_ref.abs().max().item())
out_ref = l2[0] + F.conv2d(F.relu(r_ref.permute(0, 3, 1, 2)), w2c, sd['decoder.local_ResMM.conv2.bias'].to(gpu), padding=1).permute(0, 2, 3, 1)
print('forward l2[2]:', (l2[2] - out_ref).abs().max().item())
g_r = B.dgrad(plan, 'local_ResMM.conv2', g, K, h2, w2, mask=l2[1])
ref = F.conv_transpose2d(g.permute(0, 3, 1, 2), w2c, padding=1).permute(0, 2, 3, 1) * (l2[1] > 0)
d = (g_r - ref).abs()
print('g_r err', d.max().item(), 'ref max', ref.abs().max().item(), 'bad frac', (d > 1e-4).float().mean().item())
bad = (d > 1e-4).nonzero()
print(bad[:20].tolist())
print('---- real chain')
torch.manual_seed(0)
G = torch.randn(K, H, W, device=gpu)
L = vfloodnet_amd._lib.lib()
from vfloodnet_amd._lib import ptr, stream, check
p = plan
g_o = torch.zeros(K, 2 * h2, 2 * w2, 4, device=gpu)
check(L.vfn_tail_grad_o_f32(ptr(G), ptr(p.p_up), ptr(p.unc), ptr(p.conf), ptr(p.qq), ptr(g_o), K, h2, w2, p.pad[2], p.pad[0], p.H0, p.W0, stream()), 'a')
g_p2 = torch.empty(K, h2, w2, 4, device=gpu)
check(L.vfn_upsample2x_add_backward_f32(ptr(g_o), None, ptr(g_p2), K, 2 * h2, 2 * w2, 4, 0, stream()), 'b')
g_q = torch.zeros(K, h2, w2, 32, device=gpu)
g_cf = torch.empty(K, h2, w2, device=gpu); g_u = torch.empty(h2, w2, device=gpu)
check(L.vfn_tail_split_f32(ptr(g_p2), ptr(p.unc), ptr(p.conf), ptr(p.qq), ptr(g_q), ptr(g_cf), ptr(g_u), K, h2 * w2, stream()), 'c')
wp2 = sd['decoder.local_pred2.weight'].to(gpu)
gy = B.dgrad(p, 'local_pred2', g_q, K, h2, w2, mask=l2[2])
ref_gy = F.conv_transpose2d(g_q[..., :2].permute(0, 3, 1, 2), wp2, padding=1).permute(0, 2, 3, 1) * (l2[2] > 0)
print('gy err', (gy - ref_gy).abs().max().item(), ref_gy.abs().max().item())
g_r = B.dgrad(p, 'local_ResMM.conv2', gy, K, h2, w2, mask=l2[1])
ref_gr = F.conv_transpose2d(ref_gy.permute(0, 3, 1, 2), w2c, padding=1).permute(0, 2, 3, 1) * (l2[1] > 0)
print('g_r err', (g_r - ref_gr).abs().max().item(), ref_gr.abs().max().item())
dw, db = B.wgrad(p, l2[0], g_r, True)
xr = F.relu(l2[0]).permute(0, 3, 1, 2)
ref_dw = torch.nn.grad.conv2d_weight(xr, w1c.shape, ref_gr.permute(0, 3, 1, 2), padding=1)
print('dw1 err', (dw - ref_dw).abs().max().item(), ref_dw.abs().max().item(), 'db err', (db - ref_gr.sum((0, 1, 2))).abs().max().item(), db.abs().max().item())
g_x = B.dgrad(p, 'local_ResMM.conv1', g_r, K, h2, w2, mask=l2[0], res=gy)
ref_gx = F.conv_transpose2d(ref_gr.permute(0, 3, 1, 2), w1c, padding=1).permute(0, 2, 3, 1) * (l2[0] > 0) + ref_gy
print('g_x err', (g_x - ref_gx).abs().max().item(), ref_gx.abs().max().item())
print('---- oracle forward vs HIP buffers')
from oracle import afb_urr_ref as O
nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu().double()
sd64 = {n: t.double() for n, t in sd.items() if n.startswith('decoder.') and t.is_floating_point()}
mem = nchw(plan.dec_in); q_out = qs.kv_q[slot, :, 128:].t().reshape(1, 512, plan.h16, plan.w16).cpu().double()
r3 = nchw(qs.q['res3']['out'][slot:slot + 1]); r2 = nchw(qs.q['res2']['out'][slot:slot + 1]); r1 = nchw(qs.q['r1'][slot:slot + 1])
e = lambda t: t.expand(K, -1, -1, -1)
out, parts = O.decoder(sd64, torch.cat([mem, e(q_out)], 1), e(r3), e(r2), e(r1), (1, K, h2, w2), return_parts=True)
print('rough', (parts['rough'][:, 0] - plan.rough.cpu().double()).abs().max().item())
print('r1_local', (parts['r1_local'] - nchw(plan.lm)).abs().max().item(), parts['r1_local'].abs().max().item())
print('conf', (parts['r1_conf'][:, 0] - plan.conf.cpu().double()).abs().max().item())
lmm = torch.cat([e(r1), parts['r1_local']], 1)
x_or = O._conv3(sd64, 'decoder.local_convFM', lmm)
print('x', (x_or - nchw(l2[0])).abs().max().item(), x_or.abs().max().item(), 'sign flips', ((x_or > 0) != (nchw(l2[0]) > 0)).float().mean().item())
r_or = O._conv3(sd64, 'decoder.local_ResMM.conv1', F.relu(x_or))
print('r', (r_or - nchw(l2[1])).abs().max().item(), 'sign flips', ((r_or > 0) != (nchw(l2[1]) > 0)).float().mean().item())
print('unc', (parts['unc'][0, 0] - plan.unc.cpu().double()).abs().max().item())
print('---- full run_tail vs oracle autograd, same G')
G = torch.randn(K, H, W, generator=torch.Generator().manual_seed(H + W)).to(gpu)
grads, gin = B.run_tail(plan, G, qs, slot)
torch.cuda.synchronize()
sd64 = {n: t.double().clone().requires_grad_() for n, t in sd.items() if n.startswith('decoder.') and t.is_floating_point()}
mem.requires_grad_(); q_out.requires_grad_(); r3.requires_grad_(); r2.requires_grad_(); r1.requires_grad_()
acts = {}
orig_conv3 = O._conv3
def hooked(sd_, p_, x_):
    y = orig_conv3(sd_, p_, x_)
    if p_.startswith('decoder.local'):
        y.retain_grad(); acts[p_] = (x_, y)
    return y
O._conv3 = hooked
out = O.decoder(sd64, torch.cat([mem, e(q_out)], 1), e(r3), e(r2), e(r1), (1, K, h2, w2))
sc = torch.clamp(out, 1e-7, 1 - 1e-7); sc = torch.log(sc / (1 - sc))
(sc * G.cpu().double()).sum().backward()
O._conv3 = orig_conv3
rel = lambda a, b: (a.double() - b.double()).abs().max().item() / b.double().abs().max().item()
for n in ('decoder.local_pred2', 'decoder.local_ResMM.conv2', 'decoder.local_ResMM.conv1', 'decoder.local_convFM'):
    print(n, 'weight rel', rel(grads[n + '.weight'].cpu(), sd64[n + '.weight'].grad), 'grad wrt output max', acts[n][1].grad.abs().max().item())
# oracle's gradient w.r.t. conv2 output (= gy) and conv1 output (g_r before mask is applied downstream)
gy_or = acts['decoder.local_ResMM.conv2'][1].grad      # dL/d(conv2 out)
gr_or = acts['decoder.local_ResMM.conv1'][1].grad      # dL/d(conv1 out) = g_r
gy_h = B.dgrad(p, 'local_pred2', None, K, h2, w2) if False else None
print('---- chain with the failing G vs the oracle grads of the intermediate tensors')
g_o = torch.zeros(K, 2 * h2, 2 * w2, 4, device=gpu)
check(L.vfn_tail_grad_o_f32(ptr(G), ptr(p.p_up), ptr(p.unc), ptr(p.conf), ptr(p.qq), ptr(g_o), K, h2, w2, p.pad[2], p.pad[0], p.H0, p.W0, stream()), 'a')
g_p2 = torch.empty(K, h2, w2, 4, device=gpu)
check(L.vfn_upsample2x_add_backward_f32(ptr(g_o), None, ptr(g_p2), K, 2 * h2, 2 * w2, 4, 0, stream()), 'b')
g_q = torch.zeros(K, h2, w2, 32, device=gpu)
check(L.vfn_tail_split_f32(ptr(g_p2), ptr(p.unc), ptr(p.conf), ptr(p.qq), ptr(g_q), ptr(g_cf), ptr(g_u), K, h2 * w2, stream()), 'c')
gq_or = acts['decoder.local_pred2'][1].grad
print('g_q', rel(nchw(g_q[..., :2]), gq_or))
gy = B.dgrad(p, 'local_pred2', g_q, K, h2, w2, mask=l2[2])
print('gy (dL/d l2[2])', rel(nchw(gy), gy_or))
g_r = B.dgrad(p, 'local_ResMM.conv2', gy, K, h2, w2, mask=l2[1])
print('g_r', rel(nchw(g_r), gr_or))
d = (nchw(g_r) - gr_or).abs()
print('bad frac', (d > 1e-5).float().mean().item(), 'where oracle nonzero & hip zero', ((gr_or != 0) & (nchw(g_r) == 0)).float().mean().item(), 'hip nonzero & oracle zero', ((gr_or == 0) & (nchw(g_r) != 0)).float().mean().item())
