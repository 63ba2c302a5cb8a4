"""Round 6: the software-pipelined plain-bf16 apply kernel (memread_apply_pipe_kernel) against memread_apply_shw_kernel<false>:
bit-identity of the read-out and of the hit counts over bank sizes that exercise one chunk, partial last chunks and empty
slices, then the apply launch alone at C5-size banks, alternating the two kernels in one process (VFN_APPLY_PIPE is read at
every call).  Usage: bench_apply_bf16.py [entries ...]"""
import os, sys, types, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd.feature_bank import FeatureBank, pick_nsplit
from vfloodnet_amd.engine import Engine
from vfloodnet_amd import _lib
dev = torch.device('cuda', 0)
HW = 1620


def setup(B, hw=HW, budget=None):
    fb = FeatureBank(2, budget or max(250000, int(2.6 * B)), dev, precision='bf16')
    fb._hw = hw
    fb._alloc(hw, B)
    g = torch.Generator(device=dev).manual_seed(B)
    fb._kbuf[:, :B].copy_(torch.randn(2, B, 128, device=dev, generator=g))
    fb._vbuf[:, :B].copy_(torch.randn(2, B, 512, device=dev, generator=g))
    fb._set_lengths([B, max(1, B - 37)])            # two objects of different length: a partial last chunk on one of them
    kvq = torch.randn(1, hw, 640, device=dev, generator=g)
    kvq[..., :128] *= 2.0
    plan = types.SimpleNamespace(HW=hw, kv_q=kvq, ml=torch.empty(2, hw, 2, device=dev),
                                 ml_part=torch.empty(2, 256, hw, 2, device=dev), work=torch.zeros(4, dtype=torch.int32, device=dev),
                                 o_part=torch.empty(2, 20, hw, 512, device=dev), dec_in=torch.empty(2, hw, 512, device=dev))
    return fb, plan


def read(fb, plan, pipe, update=True):
    os.environ['VFN_APPLY_PIPE'] = '1' if pipe else '0'
    fb._cnt.zero_()
    info0 = fb._ibuf.clone()
    Engine._memory_read(types.SimpleNamespace(mode=1), plan, fb, update)
    torch.cuda.synchronize()
    out = plan.dec_in.clone()
    info = fb._ibuf.clone()
    fb._ibuf.copy_(info0)
    return out, info


ok = True
for B, hw in [(60, 60), (64, 150), (65, 150), (127, 60), (128, 1620), (1000, 150), (5000, 1620), (25037, 1620)]:
    fb, plan = setup(B, hw)
    o1, i1 = read(fb, plan, True)
    o0, i0 = read(fb, plan, False)
    same = torch.equal(o1, o0) and torch.equal(i1, i0)
    ok &= same
    print(f'B={B:6d} HW={hw:5d}: pipe == shw bit for bit: {same}   max|d|={float((o1 - o0).abs().max()):.3e}  finite={bool(torch.isfinite(o1).all())}', flush=True)
print('IDENTICAL' if ok else 'MISMATCH', flush=True)

L = _lib.lib()
orig = L.vfn_memread_apply
rec = {}


def timed(desc, s):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = orig(desc, s)
    e1.record()
    rec.setdefault(os.environ['VFN_APPLY_PIPE'], []).append((e0, e1))
    return rc


for B in [int(x) for x in (sys.argv[1:] or ['56000', '330000', '660000', '1000000'])]:
    fb, plan = setup(B)
    for w in range(2):
        read(fb, plan, True); read(fb, plan, False)
    L.vfn_memread_apply = timed
    rec.clear()
    for r in range(6):
        read(fb, plan, True, update=True); read(fb, plan, False, update=True)
    L.vfn_memread_apply = orig
    torch.cuda.synchronize()
    ent = 2 * B - 37
    fl = 1280.0 * ent * HW
    res = {}
    for k, v in rec.items():
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in v)
        res[k] = (ts[len(ts) // 2], ts[0])
    print(f'B={B:8d} entries/object  nsplit={pick_nsplit(HW, 2, B, 128, 20) if False else "-"}: '
          f'shw<false> median {res["0"][0]:9.1f} us ({fl / res["0"][0] / 1e6:7.1f} TF = {fl / res["0"][0] / 1e6 / 2500:.3f})   '
          f'pipe median {res["1"][0]:9.1f} us ({fl / res["1"][0] / 1e6:7.1f} TF = {fl / res["1"][0] / 1e6 / 2500:.3f})   x{res["0"][0] / res["1"][0]:.3f}', flush=True)
