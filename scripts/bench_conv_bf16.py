"""bf16-operand conv: TFLOP/s per tile config (and split-K) on the layer shapes of the path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfloodnet_amd
from vfloodnet_amd import ops

dev = torch.device('cuda', 0)
MODE = int(os.environ.get('MODE', '1'))
SHAPES = [(2, 120, 216, 64, 64, 3, 1), (2, 120, 216, 256, 64, 1, 1), (2, 120, 216, 64, 64, 1, 1), (1, 120, 216, 64, 64, 3, 1), (4, 120, 216, 64, 64, 3, 1),
          (2, 120, 216, 256, 256, 3, 1), (1, 120, 216, 256, 256, 3, 1), (2, 60, 108, 256, 256, 3, 1),
          (1, 60, 108, 128, 128, 3, 1), (1, 30, 54, 256, 256, 3, 1), (2, 30, 54, 1024, 256, 3, 1), (1, 30, 54, 1024, 640, 3, 1),
          (1, 30, 54, 256, 1024, 1, 1), (1, 30, 54, 1024, 256, 1, 1), (1, 120, 216, 64, 256, 1, 1), (1, 120, 216, 256, 64, 1, 1),
          (2, 240, 432, 128, 32, 3, 1)]
tiles = ops.conv_cfg_tiles()

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for (N, H, W, Cin, Cout, k, s) in SHAPES:
    x = torch.randn(N, H, W, Cin, device=dev)
    wp = ops.pad_rows(torch.randn(Cout, k * k * Cin, device=dev) * 0.05)
    sc = torch.ones(Cout, device=dev); sh = torch.zeros(Cout, device=dev)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    out = torch.empty(N, Ho, Wo, Cout, device=dev)
    d = ops.make_conv_desc(x, wp, Cout, k, k, s, k // 2, out, sc, sh, None, True, False)
    fl = 2.0 * d.M * Cout * k * k * Cin
    best32 = min(timeit(lambda: ops.conv2d_launch(d, c)) for c in (0, 2, 3, 8, 9, 10) if not (tiles[c][1] > 64 and Cout <= 32))
    line = f'M={d.M:6d} Cout={Cout:4d} K={k*k*Cin:5d} f32 best {fl/best32/1e6:5.1f} ({best32:6.1f}us) | mode {MODE}: '
    for c in ops.BF16_CFGS:
        bm, bn = tiles[c]
        if (bn > 64 and Cout <= 32) or (bn > 128 and Cout < 256):
            continue
        us = timeit(lambda: ops.conv2d_launch(d, c, mode=MODE))
        line += f'c{c}:{fl / us / 1e6:5.0f} '
    print(line, flush=True)
    if d.M <= 8192:
        ws = torch.empty(16 * d.M * Cout, device=dev)
        for c in (0, 2, 3, 10):
            line = f'     split-K cfg{c}: '
            for ks in ops.valid_splits(d, 16, mode=MODE)[1:]:
                ops.set_splitk(d, ks, ws)
                us = timeit(lambda: ops.conv2d_launch(d, c, mode=MODE))
                line += f's{ks}:{fl / us / 1e6:5.0f} '
            ops.set_splitk(d, 1, None)
            print(line, flush=True)
