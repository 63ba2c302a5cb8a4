"""Debug: where the waves of one f32 wide memread_apply launch spend their cycles (needs a -DVFN_CENSUS build of the
library: VFN_LIB_PATH=.../libvfn_census.so).  Per wave and chunk, shader-clock cycles of: score GEMM, softmax + P^T
write, wait at the P^T barrier, P^T V GEMM, wait at the end-of-chunk barrier."""
import sys, os, math, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vfloodnet_amd
from vfloodnet_amd.feature_bank import FeatureBank, pick_nsplit, pick_scan_slices, MAX_SPLIT, MAX_SPLIT_SCAN, QT_SCAN, DK, DV
from vfloodnet_amd import _lib
from vfloodnet_amd._lib import ptr, stream, check, BankScanDesc, MemReadDesc
C = _lib.C
dev = torch.device('cuda', 0); HW, K = 1620, 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 56000
L = _lib.lib()
fb = FeatureBank(K, max(250000, int(2.6 * B)), dev); fb._hw = HW; fb._alloc(HW, B)
fb._kbuf.normal_(); fb._vbuf.normal_(); fb._set_lengths([B] * K)
cap = fb._cap
kv_q = torch.randn(1, HW, 640, device=dev)
ml = torch.empty(K, HW, 2, device=dev); ml_part = torch.empty(K, MAX_SPLIT_SCAN, HW, 2, device=dev)
o_part = torch.empty(K, MAX_SPLIT, HW, DV, device=dev); dec_in = torch.empty(K, HW, DV, device=dev)
ns = pick_scan_slices(HW, K, B); work = torch.zeros(4, dtype=torch.int32, device=dev)
d = BankScanDesc()
d.q, d.bank_k, d.bank_len, d.rowscale, d.part = ptr(kv_q), ptr(fb._kbuf), ptr(fb._len_dev), None, ptr(ml_part)
d.stride_q, d.stride_k, d.stride_rs = 0, cap * DK, 0
d.scale = 1 / math.sqrt(DK); d.work_counter = ptr(work)
d.ldq, d.q_per_obj, d.HW, d.obj_n, d.nsplit, d.mode, d.precision = DK + DV, 0, HW, K, ns, 0, 0
check(L.vfn_bank_scan(C.byref(d), stream()), 'scan')
check(L.vfn_bank_scan_finish(ptr(ml_part), ns, HW, K, 0, ptr(ml), None, None, None, stream()), 'fin')
nsplit = pick_nsplit(HW, K, B, QT_SCAN, MAX_SPLIT)
m = MemReadDesc()
m.q, m.qv = ptr(kv_q), None
m.bank_k, m.bank_v, m.bank_len, m.ml, m.o_part = ptr(fb._kbuf), ptr(fb._vbuf), ptr(fb._len_dev), ptr(ml), ptr(o_part)
m.cnt, m.info, m.out = ptr(fb._cnt), ptr(fb._ibuf), ptr(dec_in)
m.stride_k, m.stride_v, m.stride_cnt, m.stride_info = cap * DK, cap * DV, cap, cap * 2
m.scale, m.thres = 1 / math.sqrt(DK), 1e-3
m.ldq, m.ldqv, m.ld_out, m.HW, m.obj_n, m.nsplit, m.precision = DK + DV, DK + DV, DV, HW, K, nsplit, 0
for _ in range(3):
    check(L.vfn_memread_apply(C.byref(m), stream()), 'apply')
torch.cuda.synchronize()
buf = np.zeros(4096 * 4, np.uint64)
L.vfn_debug_census.argtypes = [ctypes.c_void_p]
assert L.vfn_debug_census(buf.ctypes.data_as(ctypes.c_void_p)) == 0
c = buf.reshape(2048, 8).astype(np.float64)
c = c[c[:, 5] > 0]
per = c[:, :5] / c[:, 5:6]
names = ['score GEMM', 'softmax + P^T write', 'wait P^T barrier', 'P^T V GEMM', 'wait end barrier']
print('waves sampled', len(c), 'chunks per wave', c[:, 5].mean(), ' nsplit', nsplit)
tot = per.sum(1).mean()
for k, n in enumerate(names):
    print('  %-22s %8.0f cycles/chunk (%4.1f %%)  min %.0f max %.0f' % (n, per[:, k].mean(), 100 * per[:, k].mean() / tot, per[:, k].min(), per[:, k].max()))
print('  total %.0f cycles/chunk; MFMA issue alone: score 64 x 64 + P^T V 256 x 64 = 20480 per wave, two waves per SIMD' % tot)
