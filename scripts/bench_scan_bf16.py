"""Round 6: the register-staged plain-bf16 scans (bank_scan_pipe_kernel<0|1>) against bank_scan_kernel<MODE, 1>: bit-identity of
the per-slice partials over bank sizes incl. one chunk / partial last chunk / empty slices, then the launch alone at C5-size
banks, alternating in one process (VFN_SCAN_PIPE is read at every call).  Usage: bench_scan_bf16.py [entries ...]"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd.feature_bank import FeatureBank, pick_scan_slices, MAX_SPLIT_SCAN, DK, DV
from vfloodnet_amd import _lib
from vfloodnet_amd._lib import ptr, stream, check, BankScanDesc
dev = torch.device('cuda', 0)
L = _lib.lib()


def setup(B, hw):
    fb = FeatureBank(2, max(250000, int(2.6 * B)), dev, precision='bf16')
    fb._hw = hw
    fb._alloc(hw, B)
    g = torch.Generator(device=dev).manual_seed(B + 1)
    fb._kbuf[:, :B].copy_(torch.randn(2, B, 128, device=dev, generator=g))
    fb._vbuf[:, :B].copy_(torch.randn(2, B, 512, device=dev, generator=g))
    fb._set_lengths([B, max(1, B - 37)])
    kvq = torch.randn(2, hw, 640, device=dev, generator=g)
    rs = torch.rand(2, fb._cap, device=dev, generator=g) + 0.5
    return fb, kvq, rs


def scan(fb, kvq, rs, hw, mode, pipe, part, work):
    os.environ['VFN_SCAN_PIPE'] = '1' if pipe else '0'
    klp, _ = fb.lp_image()
    nsplit = pick_scan_slices(hw, 2, fb.len_upper())
    d = BankScanDesc()
    d.q, d.bank_k, d.bank_len, d.part = ptr(kvq), ptr(fb._kbuf), ptr(fb._len_dev), ptr(part)
    d.rowscale = ptr(rs) if mode == 1 else None
    d.stride_q, d.stride_k, d.stride_rs = (hw * 640 if mode == 1 else 0), fb._cap * DK, (fb._cap if mode == 1 else 0)
    d.scale = 1.0 / math.sqrt(DK)
    d.ldq, d.q_per_obj, d.HW, d.obj_n, d.nsplit, d.mode = 640, (1 if mode == 1 else 0), hw, 2, nsplit, mode
    d.precision = 1
    d.work_counter = ptr(work)
    d.bank_k_lp = ptr(klp)
    check(L.vfn_bank_scan(_lib.C.byref(d), stream()), 'vfn_bank_scan')
    return nsplit


ok = True
for B, hw in [(60, 60), (64, 150), (65, 150), (127, 60), (128, 1620), (1000, 150), (5000, 1620), (25037, 1620)]:
    fb, kvq, rs = setup(B, hw)
    work = torch.zeros(4, dtype=torch.int32, device=dev)
    for mode in (0, 1):
        parts = []
        for pipe in (True, False):
            part = torch.full((2, MAX_SPLIT_SCAN, hw, 2), float('nan'), device=dev)
            ns = scan(fb, kvq, rs, hw, mode, pipe, part, work)
            torch.cuda.synchronize()
            parts.append(part[:, :ns].clone())
        same = torch.equal(parts[0].view(torch.int32), parts[1].view(torch.int32))
        ok &= same
        print(f'B={B:6d} HW={hw:5d} mode {mode}: pipe == dma bit for bit: {same}', flush=True)
print('IDENTICAL' if ok else 'MISMATCH', flush=True)

HW = 1620
for B in [int(x) for x in (sys.argv[1:] or ['56000', '330000', '660000', '1000000'])]:
    fb, kvq, rs = setup(B, HW)
    work = torch.zeros(4, dtype=torch.int32, device=dev)
    part = torch.empty(2, MAX_SPLIT_SCAN, HW, 2, device=dev)
    for mode in (0, 1):
        res = {}
        for rep in range(8):
            for pipe in (True, False):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                scan(fb, kvq, rs, HW, mode, pipe, part, work)
                e1.record()
                torch.cuda.synchronize()
                if rep >= 2:
                    res.setdefault(pipe, []).append(e0.elapsed_time(e1) * 1e3)
        fl = 256.0 * (2 * B - 37) * HW
        m = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        print(f'B={B:8d} mode {mode}: dma {m[False]:9.1f} us ({fl / m[False] / 1e6:7.1f} TF = {fl / m[False] / 1e6 / 2500:.3f})   '
              f'pipe {m[True]:9.1f} us ({fl / m[True] / 1e6:7.1f} TF = {fl / m[True] / 1e6 / 2500:.3f})   x{m[False] / m[True]:.3f}', flush=True)
