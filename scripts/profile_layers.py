"""Per-launch timing of one frame's plan (HIP events, each launch repeated) -> layer table."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfloodnet_amd
from vfloodnet_amd import AFB_URR, FeatureBank, ops
from tools import synth

dev = torch.device('cuda', 0)
H0, W0 = 480, 854
sd = synth.make_state_dict(20200212)
model = AFB_URR(dev, update_bank=True).to(dev).eval(); model.load_state_dict(sd)
eng = model.engine()
if '--tune' in sys.argv:
    eng.autotune(H0, W0, 2)
p = eng.plan(H0, W0, 2)
frames, m0 = synth.clip(1, 2, H0, W0)
oh = synth.onehot(m0).unsqueeze(0).to(dev)
k, v = model.memorize(frames[0:1].to(dev), oh)
fb = FeatureBank(2, 250000, dev); fb.init_bank(k, v)
model.segment(frames[1:2].to(dev), fb)
tiles = ops.conv_cfg_tiles()
reps = 5
tot = {}
rows = []
eng.autotune(H0, W0, 2, only_missing=True)
lists = {'seg_pre': p.seg_pre, 'seg_post': p.seg_post, 'mem': p.mem, 'seg_pre_x2': p.qsets[0].pre[2]}
for lname, lst in lists.items():
    for l in lst:
        l(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): l()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        cfg = ''
        if l.fn is ops.conv2d_launch:
            if l.args[1] >= ops.WINO_GEMM_CFG0:          # the persistent transform-domain GEMM: tile / request depth / workgroups
                c_ = l.args[1] - (ops.PCONV_CFG0 if l.args[1] >= ops.PCONV_CFG0 else ops.WINO_GEMM_CFG0)
                cfg = 'p%dx%d/pd%d/%d' % (ops.WINO_GEMM_TILES[c_ & 3][0], ops.WINO_GEMM_TILES[c_ & 3][1], 1 + ((c_ & 7) >> 2), (c_ >> 3) * 128)
            else:
                cfg = '%dx%d' % tiles[l.args[1]] + ('/k%d' % l.args[0].ksplit if l.args[0].ksplit > 1 else '') + ('/wk%d' % ops.conv_cfg_wk(l.args[1]) if ops.conv_cfg_wk(l.args[1]) > 1 else '')
        tf = l.flops / us / 1e6 if l.flops else 0
        rows.append((lname, l.name, cfg, us, tf))
        t = tot.setdefault(lname, [0.0, 0.0]); t[0] += us; t[1] += l.flops
for r in rows:
    print('%-10s %-48s %-12s %8.1f us %7.1f TF' % r)
for k_, (us, fl) in tot.items():
    print(k_, 'total %.1f us, %.1f GFLOP, %.1f TF' % (us, fl / 1e9, fl / us / 1e6 if us else 0))

# whole lists back to back (as a frame runs them): the per-launch table above repeats each launch in isolation
for lname, lst in lists.items():
    for l in lst: l()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for l in lst: l()
    e1.record(); torch.cuda.synchronize()
    print(lname, 'list back to back: %.1f us' % (e0.elapsed_time(e1) * 1e3 / reps))
