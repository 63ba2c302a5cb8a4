"""Weight gradient of the decoder's 3x3 layers: the direct implicit GEMM over the pixels (ops.conv_wgrad) against the Winograd-domain
form (ops.conv_wgrad_winograd), per shape of the training step (N x H x W, Cin -> Cout)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import ops
dev = torch.device('cuda', 0)


def t_us(fn, iters=10):
    fn(); torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        best = us if best is None else min(best, us)
    return best


for (N, H, W, cin, cout) in [(2, 100, 100, 256, 256), (1, 100, 100, 256, 256), (5, 100, 100, 256, 256), (2, 50, 50, 256, 256), (1, 50, 50, 256, 256),
                             (5, 50, 50, 256, 256), (1, 50, 50, 512, 256), (5, 50, 50, 512, 256), (2, 25, 25, 256, 256), (2, 25, 25, 512, 256),
                             (5, 25, 25, 1024, 640), (2, 200, 200, 32, 32), (5, 100, 100, 64, 64), (5, 50, 50, 128, 128)]:
    x = torch.randn(N, H, W, cin, device=dev)
    gy = torch.randn(N, H, W, cout, device=dev)
    out = torch.empty(cout, 9 * cin, device=dev)
    d = t_us(lambda: ops.conv_wgrad(x, gy, 3, 1, 1, relu=True, out=out))
    w = t_us(lambda: ops.conv_wgrad_winograd(x, gy, relu=True, out=out))
    parts = [t_us(lambda: ops.winograd_input(x, torch.empty(36 * ops.winograd_rows(N, H, W) * cin, device=dev), ops.winograd_rows(N, H, W), True))]
    print(f'M {N * H * W:6d} {cin:4d} -> {cout:4d}: direct {d:7.1f} us   winograd {w:7.1f} us   ({d / w:4.2f}x)', flush=True)
