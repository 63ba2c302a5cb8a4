"""Micro-benchmark of the bank kernels (scan mode 0/1, apply) at a given bank size."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd.feature_bank import FeatureBank, pick_nsplit
from vfloodnet_amd.engine import Engine
from vfloodnet_amd import _lib
dev = torch.device('cuda', 0)
HW = 1620
MODE = int(os.environ.get('MODE', '0'))
eng = types.SimpleNamespace(mode=MODE)
for B in [int(x) for x in (sys.argv[1:] or ['1620', '25000', '100000'])]:
    fb = FeatureBank(2, 250000, dev)
    fb._hw = HW
    fb._alloc(HW, B)
    fb._kbuf.normal_(); fb._vbuf.normal_()
    fb._set_lengths([B, B])
    plan = types.SimpleNamespace(HW=HW, kv_q=torch.randn(1, HW, 640, device=dev), ml=torch.empty(2, HW, 2, device=dev),
                                 ml_part=torch.empty(2, 256, HW, 2, device=dev), work=torch.zeros(4, dtype=torch.int32, device=dev), o_part=torch.empty(2, 20, HW, 512, device=dev),
                                 dec_in=torch.empty(2, HW, 512, device=dev))
    for _ in range(2):
        Engine._memory_read(eng, plan, fb, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        Engine._memory_read(eng, plan, fb, True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 200
    fl = 2 * (2 * 2 * 128 + 2 * 512) * B * HW
    newk = [torch.randn(128, HW, device=dev) for _ in range(2)]; newv = [torch.randn(512, HW, device=dev) for _ in range(2)]
    print(f'mode {MODE} B={B:6d} nsplit={pick_nsplit(HW, 2, B)}: memory read (scan+apply+finish) {us:8.1f} us  {fl / us / 1e6:6.1f} TF')
