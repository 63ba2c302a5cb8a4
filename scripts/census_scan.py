"""Debug: which CU every workgroup of one bank_scan launch ran on and when (needs a -DVFN_CENSUS build, VFN_LIB_PATH)."""
import sys, os, math, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vfloodnet_amd
from vfloodnet_amd.feature_bank import FeatureBank, pick_nsplit, MAX_SPLIT_SCAN, QT_SCAN, DK, DV
from vfloodnet_amd import _lib
from vfloodnet_amd._lib import ptr, stream, check, BankScanDesc
C = _lib.C
dev = torch.device('cuda', 0); HW, K, B = 1620, 2, 56000
L = _lib.lib()
fb = FeatureBank(K, 250000, dev); fb._hw = HW; fb._alloc(HW, B); fb._kbuf.normal_(); fb._set_lengths([B] * K)
kv_q = torch.randn(1, HW, 640, device=dev); ml_part = torch.empty(K, MAX_SPLIT_SCAN, HW, 2, device=dev)
ns = int(os.environ.get('NS', 78))
work = torch.zeros(4, dtype=torch.int32, device=dev)
d = BankScanDesc()
d.q, d.bank_k, d.bank_len, d.rowscale, d.part = ptr(kv_q), ptr(fb._kbuf), ptr(fb._len_dev), None, ptr(ml_part)
d.stride_q, d.stride_k, d.stride_rs = 0, fb._cap * DK, 0
d.scale = 1 / math.sqrt(DK)
d.work_counter = ptr(work)
d.ldq, d.q_per_obj, d.HW, d.obj_n, d.nsplit, d.mode, d.precision = DK + DV, 0, HW, K, ns, 0, 0
for _ in range(3):
    check(L.vfn_bank_scan(C.byref(d), stream()), 'scan')
torch.cuda.synchronize()
buf = np.zeros(4096 * 4, np.uint64)
L.vfn_debug_census.argtypes = [ctypes.c_void_p]
assert L.vfn_debug_census(buf.ctypes.data_as(ctypes.c_void_p)) == 0
n = min(512, 13 * ns * K)
c = buf.reshape(4096, 4)[:n]
t0 = c[:, 0].min()
start, end = (c[:, 0] - t0) / 100.0, (c[:, 1] - t0) / 100.0          # us (100 MHz)
hw = c[:, 2]; xcc = c[:, 3] & 0xf
cu = ((hw >> 8) & 0xf); sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7       # gfx9 HW_ID: cu_id[11:8] sh_id[12] se_id[15:13]
key = xcc * 10000 + se * 100 + sh * 50 + cu
print('workgroups', n, 'distinct CU keys', len(set(key.tolist())), 'kernel span %.1f us' % end.max())
print('per-WG duration us: median %.1f min %.1f max %.1f' % (np.median(end - start), (end - start).min(), (end - start).max()))
# concurrency per CU: max number of overlapping intervals
mx = 0
for k_ in set(key.tolist()):
    iv = sorted([(s, 1) for s in start[key == k_]] + [(e, -1) for e in end[key == k_]])
    cur = 0
    for _, dlt in iv:
        cur += dlt; mx = max(mx, cur)
print('max concurrent workgroups on one CU:', mx)
late = (start > 5).sum()
print('workgroups that started more than 5 us after the first:', int(late))
