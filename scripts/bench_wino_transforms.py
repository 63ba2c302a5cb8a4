"""The HBM-bound kernels around the Winograd GEMMs at the C2 frame's shapes: input / output transform, upsample + add,
max-pool -- microseconds and effective TB/s (algorithmic bytes: each operand once).  usage: bench_wino_transforms.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import ops
dev = torch.device('cuda', 0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


flush = torch.empty(64 * 1024 * 1024, device=dev)          # 256 MB: evicts L2 / Infinity Cache between timed groups (not inside)
for (N, H, W, C) in [(2, 120, 216, 256), (1, 120, 216, 256), (2, 60, 108, 256), (2, 30, 54, 256), (2, 30, 54, 1024), (2, 60, 108, 128)]:
    x = torch.randn(N, H, W, C, device=dev)
    rows = ops.winograd_rows(N, H, W)
    V = torch.zeros(36 * rows, C, device=dev)
    Mb = torch.randn(36 * rows, C, device=dev)
    out = torch.empty(N, H, W, C, device=dev)
    res = torch.randn(N, H, W, C, device=dev)
    sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    act = N * H * W * C * 4
    t_in = timeit(lambda: ops.winograd_input(x, V, rows, True))
    t_out = timeit(lambda: ops.winograd_output(Mb, rows, out, N, H, W, C, sc, sh, res, C, 0, False))
    t_out0 = timeit(lambda: ops.winograd_output(Mb, rows, out, N, H, W, C, sc, sh, None, 0, 0, True))
    tiles = ops.vfn_winograd_tiles(N, H, W)
    b_in, b_out = act + 36 * tiles * C * 4, 36 * tiles * C * 4 + 2 * act
    print(f'[{N}x{H}x{W}x{C}] wino_in {t_in:6.1f} us {b_in / t_in / 1e6:5.2f} TB/s | wino_out+res {t_out:6.1f} us {b_out / t_out / 1e6:5.2f} TB/s | '
          f'wino_out {t_out0:6.1f} us {(b_out - act) / t_out0 / 1e6:5.2f} TB/s', flush=True)
for (N, h, w, C) in [(2, 120, 216, 256), (2, 60, 108, 256)]:
    s = torch.randn(1, h, w, C, device=dev)
    pm = torch.randn(N, h // 2, w // 2, C, device=dev)
    o = torch.empty(N, h, w, C, device=dev)
    t = timeit(lambda: ops.upsample2x_add(s, pm, o, True))
    b = (N * h * w * C + h * w * C + N * h * w * C // 4) * 4
    print(f'upsample2x_add [{N}x{h}x{w}x{C}] {t:6.1f} us {b / t / 1e6:5.2f} TB/s')
for (N, H, W, C) in [(2, 240, 432, 64), (1, 240, 432, 64)]:
    x = torch.randn(N, H, W, C, device=dev)
    o = torch.empty(N, H // 2, W // 2, C, device=dev)
    t = timeit(lambda: ops.maxpool3x3s2(x, o))
    b = (N * H * W * C + N * H * W * C // 4) * 4
    print(f'maxpool3x3s2 [{N}x{H}x{W}x{C}] {t:6.1f} us {b / t / 1e6:5.2f} TB/s')
