"""Per-layer A/B: the tuned LDS-tiled choice (tuned_gfx950.json) against every wave-autonomous configuration
(conv_direct.hip) with its split-K options, on the convolution shapes of one C2 frame."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfloodnet_amd
from vfloodnet_amd import ops, engine

dev = torch.device('cuda', 0)
# (N, H, W, Cin, Cout, k, stride, relu_in, residual)
SHAPES = [
    (2, 120, 216, 64, 64, 1, 1, 0, 0), (2, 120, 216, 64, 64, 3, 1, 0, 0), (2, 120, 216, 64, 256, 1, 1, 0, 1),
    (2, 120, 216, 256, 64, 1, 1, 0, 0), (2, 120, 216, 256, 128, 1, 1, 0, 0), (2, 120, 216, 128, 128, 3, 2, 0, 0),
    (2, 60, 108, 128, 512, 1, 1, 0, 1), (2, 120, 216, 256, 512, 1, 2, 0, 0), (2, 60, 108, 512, 128, 1, 1, 0, 0),
    (2, 60, 108, 128, 128, 3, 1, 0, 0), (2, 60, 108, 512, 256, 1, 1, 0, 0), (2, 60, 108, 256, 256, 3, 2, 0, 0),
    (2, 30, 54, 256, 1024, 1, 1, 0, 1), (2, 60, 108, 512, 1024, 1, 2, 0, 0), (2, 30, 54, 1024, 256, 1, 1, 0, 0),
    (2, 30, 54, 256, 256, 3, 1, 0, 0), (2, 30, 54, 1024, 640, 3, 1, 0, 0),
    (2, 30, 54, 512, 256, 3, 1, 0, 1), (2, 30, 54, 256, 256, 3, 1, 1, 1), (2, 60, 108, 256, 256, 3, 1, 1, 1),
    (2, 120, 216, 256, 256, 3, 1, 1, 1), (2, 60, 108, 512, 256, 3, 1, 0, 0), (2, 240, 432, 64, 32, 3, 1, 0, 0),
    (2, 240, 432, 32, 32, 3, 1, 1, 1),
]
if len(sys.argv) > 1 and sys.argv[1] != 'all':
    SHAPES = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
tiles = ops.conv_cfg_tiles()
D0 = 38
ws = torch.empty(engine.WS_FLOATS, device=dev)
cnt = torch.zeros(ops.SK_MAX_TILES, dtype=torch.int32, device=dev)


def timeit(d, c, iters=10):
    for _ in range(2):
        ops.conv2d_launch(d, c)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.conv2d_launch(d, c)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        best = us if best is None else min(best, us)
    return best


tot_old = tot_new = 0.0
for (N, H, W, Cin, Cout, k, s, relu_in, use_res) in SHAPES:
    x = torch.randn(N, H, W, Cin, device=dev)
    wp = ops.pad_rows(torch.randn(Cout, k * k * Cin, device=dev) * 0.05)
    sc = torch.ones(Cout, device=dev); sh = torch.zeros(Cout, device=dev)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    out = torch.empty(N, Ho, Wo, Cout, device=dev)
    res = torch.randn(N, Ho, Wo, Cout, device=dev) if use_res else None
    d = ops.make_conv_desc(x, wp, Cout, k, k, s, k // 2, out, sc, sh, res, bool(relu_in), False)
    K = k * k * Cin
    fl = 2.0 * d.M * Cout * K
    choice = engine.choose_cfg(d.M, Cout, K, 0)
    cfg = engine.apply_choice(d, choice, ws, cnt)
    t_old = timeit(d, cfg)
    ref = out.clone()
    results = []
    for c in range(D0, len(tiles)):
        bm, bn = tiles[c]
        if wp.shape[0] < ((Cout + bn - 1) // bn) * bn or (bn > 64 and Cout <= 32) or (bn >= 128 and Cout < 128 and Cout % bn):
            continue
        blocks = ((d.M + bm - 1) // bm) * ((Cout + bn - 1) // bn)
        options = [(c, 1, 0)]
        if ops.conv_cfg_kind(c) == 2:
            pass
        elif blocks < 256:
            options += [(c, k_, 0) for k_ in ops.valid_splits(d, 16)[1:] if k_ * d.M * Cout <= engine.WS_FLOATS]
        else:
            options += [(c, k_, full) for (full, k_, rows) in ops.tail_split_options(d, bm, bn, 8)
                        if k_ * rows * Cout <= engine.WS_FLOATS and k_ in (2, 3, 4, 5, 6, 8)]
        for opt in options:
            engine.apply_choice(d, opt, ws, cnt)
            out.zero_()
            t = timeit(d, c)
            err = (out - ref).abs().max().item() / max(1e-6, ref.abs().max().item())
            results.append((t, opt, err))
    results.sort(key=lambda r: r[0])
    t_new, opt, err = results[0]
    tot_old += t_old; tot_new += min(t_old, t_new)
    def nm(ch):
        return '%dx%d' % tiles[ch[0]] + (f'/wk{ops.conv_cfg_wk(ch[0])}' if ops.conv_cfg_wk(ch[0]) > 1 else '') + ('/sk' if ops.conv_cfg_kind(ch[0]) == 2 else '') + (f'/k{ch[1]}@{ch[2]}' if ch[1] > 1 else '')
    print(f'M={d.M:6d} Cout={Cout:4d} K={K:5d} r{relu_in}{use_res} | tuned cfg{choice[0]:2d} {nm(choice):16s} {t_old:7.1f} us {fl / t_old / 1e6:6.1f} TF | '
          f'direct cfg{opt[0]:2d} {nm(opt):16s} {t_new:7.1f} us {fl / t_new / 1e6:6.1f} TF  relerr {err:.1e} | '
          + '  '.join(f'{nm(o)}:{t:.1f}' for t, o, _ in results[1:4]), flush=True)
print(f'sum over shapes: tuned {tot_old:.1f} us, best-of {tot_new:.1f} us')
