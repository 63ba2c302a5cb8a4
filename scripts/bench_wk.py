"""Micro-benchmark of the small-M layers: best plain configuration (any tile, any split-K over workgroups, reduce launch
included) against the in-workgroup split-K configurations (cfg >= 26).  Times in us per layer, back to back launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import ops
dev = torch.device('cuda', 0)
# (N, H, W, Cin, Cout, k, stride): res4 / res3 layers of both encoders at 480x854, keyval-side 1x1s
SHAPES = [(1, 30, 54, 1024, 256, 1, 1), (1, 30, 54, 256, 256, 3, 1), (1, 30, 54, 256, 1024, 1, 1), (2, 30, 54, 1024, 256, 1, 1),
          (2, 30, 54, 256, 256, 3, 1), (1, 60, 108, 512, 128, 1, 1), (1, 60, 108, 128, 128, 3, 1), (1, 60, 108, 128, 512, 1, 1),
          (2, 60, 108, 512, 128, 1, 1), (2, 60, 108, 128, 128, 3, 1), (1, 120, 216, 256, 64, 1, 1), (1, 120, 216, 64, 64, 3, 1)]
tiles = ops.conv_cfg_tiles()
ws = torch.empty(64 * 1024 * 1024, device=dev)
def t(d, c):
    for _ in range(3): ops.conv2d_launch(d, c)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv2d_launch(d, c)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 50)
    return best
for (N, H, W, Cin, Cout, k, s) in SHAPES:
    x = torch.randn(N, H, W, Cin, device=dev)
    wp = ops.pad_rows(torch.randn(Cout, k * k * Cin, device=dev) * 0.05)
    sc = torch.ones(Cout, device=dev); sh = torch.zeros(Cout, device=dev)
    out = torch.empty(N, H, W, Cout, device=dev)
    d = ops.make_conv_desc(x, wp, Cout, k, k, s, k // 2, out, sc, sh, None, True, False)
    fl = 2.0 * d.M * Cout * k * k * Cin
    best = (1e9, None)
    for c, (bm, bn) in enumerate(tiles):
        if c >= 26 or d.cout_pad < ((Cout + bn - 1) // bn) * bn or (bn > 128 and Cout < 256): continue
        for ks in ops.valid_splits(d, 16):
            if ks * d.M * Cout > ws.numel(): continue
            ops.set_splitk(d, ks, ws if ks > 1 else None)
            us = t(d, c)
            if us < best[0]: best = (us, (c, ks))
    ops.set_splitk(d, 1, None)
    line = f'M={d.M:5d} Cout={Cout:4d} K={k*k*Cin:5d}: plain best {best[0]:6.1f} us cfg{best[1][0]} {tiles[best[1][0]]} ksplit {best[1][1]} ({fl/best[0]/1e6:5.1f} TF) | in-WG split:'
    for c in range(26, len(tiles)):
        bm, bn = tiles[c]
        if d.cout_pad < ((Cout + bn - 1) // bn) * bn: continue
        us = t(d, c)
        line += f'  {bm}x{bn}/{ops.conv_cfg_wk(c)}{"t" if ops.conv_cfg_tpb(c) > 1 else ""}: {us:5.1f}'
    print(line, flush=True)
