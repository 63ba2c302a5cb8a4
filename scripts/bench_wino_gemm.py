"""Probe: the Winograd-domain GEMMs of a 256 -> 256 3x3 layer at 1/4 resolution as ONE batched-filter launch
(vfn_conv_desc.w_batch_rows): 36 components x [tiles x 256] x [256 x 256], every tile configuration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfloodnet_amd
from vfloodnet_amd import ops, engine
dev = torch.device('cuda', 0)
tiles = ops.conv_cfg_tiles()
ws = torch.empty(engine.WS_FLOATS, device=dev)
cnt = torch.zeros(ops.SK_MAX_TILES, dtype=torch.int32, device=dev)
for (ntile, C, Cout, comps) in [(3240, 256, 256, 36), (1620, 256, 256, 36), (810, 256, 256, 36), (3240, 256, 256, 16)]:
    rows = (ntile + 255) // 256 * 256
    M = comps * rows
    V = torch.randn(1, 1, M, C, device=dev)
    U = torch.randn(comps * 256, C, device=dev) * 0.05            # [comps][cout_pad = 256][C]
    out = torch.empty(1, 1, M, Cout, device=dev)
    fl = 2.0 * comps * ntile * C * Cout
    res = []
    ref = None
    for c in range(len(tiles)):
        bm, bn = tiles[c]
        if rows % bm or bn > 256 or ops.conv_cfg_kind(c) == 2:
            continue
        d = ops.make_conv_desc(V, U, Cout, 1, 1, 1, 0, out, None, None, None, False, False, N=1, H=1, W=M)
        d.cout_pad = 256
        d.w_batch_rows = rows
        d.k_rot = int(os.environ.get('VFN_KROT', '1'))
        try:
            for _ in range(2):
                ops.conv2d_launch(d, c)
        except RuntimeError:
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = (V.view(comps, rows, C) @ U.view(comps, 256, C)[:, :Cout].transpose(1, 2)).view(-1, Cout)
        err = (out.view(-1, Cout) - ref).abs().max().item() / ref.abs().max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for _ in range(3):
            e0.record()
            for _ in range(5):
                ops.conv2d_launch(d, c)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 200
            best = us if best is None else min(best, us)
        res.append((best, c, err))
    res.sort()
    print(f'tiles {ntile} x {comps} comps, {C}->{Cout}: ' + '  '.join(f'cfg{c} {tiles[c][0]}x{tiles[c][1]}: {t:.0f} us {fl / t / 1e6:.0f} TF (err {e:.0e})' for t, c, e in res[:6]), flush=True)
