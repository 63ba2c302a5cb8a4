"""Each bank kernel on its own (HIP events) at given bank sizes: scan mode 0 (softmax statistics), apply (P^T V + hit
counts), finish, scan mode 1 (cosine arg-max).  usage: bench_bank_kernels.py [B ...]   (VFN_LIB_PATH selects a build)"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd.feature_bank import FeatureBank, pick_nsplit, pick_scan_slices, MAX_SPLIT, MAX_SPLIT_SCAN, QT_SCAN, DK, DV
from vfloodnet_amd import _lib
from vfloodnet_amd._lib import ptr, stream, check, MemReadDesc, BankScanDesc
C = _lib.C
dev = torch.device('cuda', 0)
HW, K = 1620, 2
PREC = int(os.environ.get('PREC', 0))
L = _lib.lib()


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for B in [int(x) for x in (sys.argv[1:] or ['56000'])]:
    fb = FeatureBank(K, max(250000, int(2.6 * B)), dev)
    fb._hw = HW
    fb._alloc(HW, B)
    fb._kbuf.normal_(); fb._vbuf.normal_()
    fb._set_lengths([B] * K)
    cap = fb._cap
    klp, vlp = (fb.lp_image() if PREC and os.environ.get('LP', '1') != '0' else (None, None))
    kv_q = torch.randn(1, HW, 640, device=dev)
    ml = torch.empty(K, HW, 2, device=dev); ml_part = torch.empty(K, MAX_SPLIT_SCAN, HW, 2, device=dev)
    o_part = torch.empty(K, MAX_SPLIT, HW, DV, device=dev); dec_in = torch.empty(K, HW, DV, device=dev)
    nsplit_scan = pick_scan_slices(HW, K, B)
    work = torch.zeros(4, dtype=torch.int32, device=dev)
    nsplit = min(int(os.environ.get('NSPLIT', 0)), MAX_SPLIT) or pick_nsplit(HW, K, B, QT_SCAN, MAX_SPLIT)   # o_part holds MAX_SPLIT slabs
    scale = 1.0 / math.sqrt(DK)
    d = BankScanDesc()
    d.q, d.bank_k, d.bank_len, d.rowscale, d.part = ptr(kv_q), ptr(fb._kbuf), ptr(fb._len_dev), None, ptr(ml_part)
    d.stride_q, d.stride_k, d.stride_rs = 0, cap * DK, 0
    d.scale = scale
    d.work_counter = ptr(work)
    d.bank_k_lp = ptr(klp)
    d.ldq, d.q_per_obj, d.HW, d.obj_n, d.nsplit, d.mode, d.precision = DK + DV, 0, HW, K, nsplit_scan, 0, PREC
    scores = None
    if PREC == 0 and os.environ.get('SCORES', '1') != '0':
        per_obj = ((cap + 63) // 64) * ((HW + 127) // 128) * 8192
        if K * per_obj * 4 < 48 * 2 ** 30:
            scores = torch.empty(K, per_obj, device=dev)
            d.scores, d.stride_scores = ptr(scores), per_obj
    t_scan0 = timeit(lambda: check(L.vfn_bank_scan(C.byref(d), stream()), 'scan'))
    check(L.vfn_bank_scan_finish(ptr(ml_part), nsplit_scan, HW, K, 0, ptr(ml), None, None, None, stream()), 'fin')
    m = MemReadDesc()
    m.q, m.qv = ptr(kv_q), None
    m.bank_k, m.bank_v, m.bank_len, m.ml, m.o_part = ptr(fb._kbuf), ptr(fb._vbuf), ptr(fb._len_dev), ptr(ml), ptr(o_part)
    m.cnt, m.info, m.out = ptr(fb._cnt), ptr(fb._ibuf), ptr(dec_in)
    m.stride_k, m.stride_v, m.stride_cnt, m.stride_info = cap * DK, cap * DV, cap, cap * 2
    m.scale, m.thres = scale, 1e-3
    m.ldq, m.ldqv, m.ld_out, m.HW, m.obj_n, m.nsplit, m.precision = DK + DV, DK + DV, DV, HW, K, nsplit, PREC
    m.bank_k_lp, m.bank_v_lp = ptr(klp), ptr(vlp)
    if scores is not None:
        m.scores, m.stride_scores = ptr(scores), scores.shape[1]
    t_apply = timeit(lambda: check(L.vfn_memread_apply(C.byref(m), stream()), 'apply'))
    t_fin = timeit(lambda: check(L.vfn_memread_finish(C.byref(m), stream()), 'finish'))
    d1 = BankScanDesc()
    new = torch.randn(K, HW, 640, device=dev)
    d1.q, d1.bank_k, d1.bank_len, d1.rowscale, d1.part = ptr(new), ptr(fb._kbuf), ptr(fb._len_dev), ptr(fb._kinv.fill_(1.0)), ptr(ml_part)
    d1.stride_q, d1.stride_k, d1.stride_rs = HW * 640, cap * DK, cap
    d1.scale = 1.0
    d1.work_counter = ptr(work)
    d1.bank_k_lp = ptr(klp)
    d1.ldq, d1.q_per_obj, d1.HW, d1.obj_n, d1.nsplit, d1.mode, d1.precision = 640, 1, HW, K, nsplit_scan, 1, PREC
    t_scan1 = timeit(lambda: check(L.vfn_bank_scan(C.byref(d1), stream()), 'scan1'))
    gf_scan = 2.0 * 128 * B * HW * K / 1e9
    gf_apply = 2.0 * 640 * B * HW * K / 1e9
    print(f'B={B}: scan0 {t_scan0:7.1f} us ({gf_scan / t_scan0 * 1e3:5.1f} TF, nsplit {nsplit_scan})  apply {t_apply:7.1f} us '
          f'({gf_apply / t_apply * 1e3:5.1f} TF executed, nsplit {nsplit})  finish {t_fin:5.1f} us  scan1 {t_scan1:7.1f} us ({gf_scan / t_scan1 * 1e3:5.1f} TF)')
