"""Micro-benchmark: every tile config of the implicit-GEMM conv on chosen layer shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfloodnet_amd
from vfloodnet_amd import ops

dev = torch.device('cuda', 0)
# (N, H, W, Cin, Cout, k, stride)
SHAPES = [(2, 120, 216, 256, 256, 3, 1), (1, 120, 216, 256, 256, 3, 1), (2, 60, 108, 256, 256, 3, 1),
          (1, 30, 54, 256, 256, 3, 1), (2, 30, 54, 1024, 256, 3, 1), (1, 30, 54, 1024, 640, 3, 1),
          (1, 30, 54, 256, 1024, 1, 1), (1, 30, 54, 1024, 256, 1, 1), (1, 120, 216, 64, 256, 1, 1),
          (2, 240, 432, 128, 32, 3, 1)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
tiles = ops.conv_cfg_tiles()
for (N, H, W, Cin, Cout, k, s) in SHAPES:
    x = torch.randn(N, H, W, Cin, device=dev)
    wp = ops.pad_rows(torch.randn(Cout, k * k * Cin, device=dev) * 0.05)
    sc = torch.ones(Cout, device=dev); sh = torch.zeros(Cout, device=dev)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    out = torch.empty(N, Ho, Wo, Cout, device=dev)
    d = ops.make_conv_desc(x, wp, Cout, k, k, s, k // 2, out, sc, sh, None, True, False)
    fl = 2.0 * d.M * Cout * k * k * Cin
    line = f'M={d.M:6d} Cout={Cout:4d} K={k*k*Cin:5d} | '
    for c, (bm, bn) in enumerate(tiles):
        if (bn > 64 and Cout <= 32) or (bn > 128 and Cout < 256):
            line += f'{bm}x{bn}: --   '
            continue
        for _ in range(2):
            ops.conv2d_launch(d, c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d_launch(d, c)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        line += f'{bm}x{bn}: {fl / us / 1e6:5.1f}  '
    print(line)
    ws = torch.empty(64 * 1024 * 1024, device=dev)
    for c in (8, 10, 17, 18, 19):
        bm, bn = tiles[c]
        line = f'     tail-split cfg{c} {bm}x{bn}: '
        for (full, ks, rows) in ops.tail_split_options(d, bm, bn, 8):
            if ks * rows * Cout > ws.numel():
                continue
            ops.set_splitk(d, ks, ws, full, rows)
            for _ in range(2):
                ops.conv2d_launch(d, c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv2d_launch(d, c)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            line += f'f{full}/s{ks}: {fl / us / 1e6:5.1f}  '
        ops.set_splitk(d, 1, None)
        print(line)
    if d.M <= 8192:
        ws = torch.empty(16 * d.M * Cout, device=dev)
        for c in (0, 1, 2, 3):
            line = f'     split-K {tiles[c][0]}x{tiles[c][1]}: '
            for ks in ops.valid_splits(d, 16)[1:]:
                ops.set_splitk(d, ks, ws)
                for _ in range(2):
                    ops.conv2d_launch(d, c)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.conv2d_launch(d, c)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 100
                line += f's{ks}: {fl / us / 1e6:5.1f}  '
            ops.set_splitk(d, 1, None)
            print(line)
