"""A/B of the reduced-precision apply kernels on one bank: prints the apply time and a checksum of O / hit counts.
usage: VFN_APPLY_PP=0|1 PREC=2 python scripts/ab_apply_pp.py B [out.pt]   (run once per setting; compare the dumps)"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd.feature_bank import FeatureBank, pick_nsplit, pick_scan_slices, MAX_SPLIT, MAX_SPLIT_SCAN, QT_SCAN, DK, DV
from vfloodnet_amd import _lib
from vfloodnet_amd._lib import ptr, stream, check, MemReadDesc, BankScanDesc
C = _lib.C
dev = torch.device('cuda', 0)
HW, K = 1620, 2
PREC = int(os.environ.get('PREC', 2))
L = _lib.lib()
B = int(sys.argv[1])
torch.manual_seed(5)
fb = FeatureBank(K, max(250000, int(1.3 * B)), dev, precision={1: 'bf16', 2: 'bf16x3'}[PREC])
fb._hw = HW
fb._alloc(HW, B)
fb._kbuf.normal_(); fb._vbuf.normal_()
fb._set_lengths([B, B - 37])
cap = fb._cap
klp, vlp = fb.lp_image()
kv_q = torch.randn(1, HW, 640, device=dev)
ml = torch.empty(K, HW, 2, device=dev); ml_part = torch.empty(K, MAX_SPLIT_SCAN, HW, 2, device=dev)
o_part = torch.zeros(K, MAX_SPLIT, HW, DV, device=dev); dec_in = torch.empty(K, HW, DV, device=dev)
nsplit_scan = pick_scan_slices(HW, K, B)
work = torch.zeros(4, dtype=torch.int32, device=dev)
nsplit = pick_nsplit(HW, K, B, QT_SCAN, MAX_SPLIT)
scale = 1.0 / math.sqrt(DK)
d = BankScanDesc()
d.q, d.bank_k, d.bank_len, d.rowscale, d.part = ptr(kv_q), ptr(fb._kbuf), ptr(fb._len_dev), None, ptr(ml_part)
d.stride_q, d.stride_k, d.stride_rs = 0, cap * DK, 0
d.scale = scale
d.work_counter = ptr(work)
d.bank_k_lp = ptr(klp)
d.ldq, d.q_per_obj, d.HW, d.obj_n, d.nsplit, d.mode, d.precision = DK + DV, 0, HW, K, nsplit_scan, 0, PREC
check(L.vfn_bank_scan(C.byref(d), stream()), 'scan')
check(L.vfn_bank_scan_finish(ptr(ml_part), nsplit_scan, HW, K, 0, ptr(ml), None, None, None, stream()), 'fin')
m = MemReadDesc()
m.q, m.qv = ptr(kv_q), None
m.bank_k, m.bank_v, m.bank_len, m.ml, m.o_part = ptr(fb._kbuf), ptr(fb._vbuf), ptr(fb._len_dev), ptr(ml), ptr(o_part)
fb._cnt.zero_()
m.cnt, m.info, m.out = ptr(fb._cnt), ptr(fb._ibuf), ptr(dec_in)
m.stride_k, m.stride_v, m.stride_cnt, m.stride_info = cap * DK, cap * DV, cap, cap * 2
m.scale, m.thres = scale, 1e-3
m.ldq, m.ldqv, m.ld_out, m.HW, m.obj_n, m.nsplit, m.precision = DK + DV, DK + DV, DV, HW, K, nsplit, PREC
m.bank_k_lp, m.bank_v_lp = ptr(klp), ptr(vlp)
check(L.vfn_memread_apply(C.byref(m), stream()), 'apply')
torch.cuda.synchronize()
o = o_part[:, :nsplit].sum(1)
cnt = fb._cnt.clone()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    check(L.vfn_memread_apply(C.byref(m), stream()), 'apply')
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 5
print(f'PP={os.environ.get("VFN_APPLY_PP", "default")} PREC={PREC} B={B}: apply {us:.1f} us ({2.0 * 640 * (2 * B - 37) * HW / us / 1e6:.1f} TF executed) nsplit {nsplit} '
      f'sum|O| {o.abs().sum().item():.6e} cnt {int(cnt.sum())} nan {int(torch.isnan(o).sum())}')
if len(sys.argv) > 2:
    torch.save({'o_part': o_part[:, :nsplit].cpu(), 'cnt': cnt.cpu()}, sys.argv[2])
