"""The persistent transform-domain GEMM (vfn_winograd_gemm_f32) against the batched-filter launch of the convolution kernels
(vfn_conv2d_nhwc_f32 with w_batch_rows, the tuned choice of the shape) on the GEMM shapes of a C2 frame: bit-identity and time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import ops, engine
dev = torch.device('cuda', 0)
ws = torch.empty(engine.WS_FLOATS, device=dev)
cnt = torch.zeros(ops.SK_MAX_TILES, dtype=torch.int32, device=dev)


def timeit(fn, n=10):
    for _ in range(2):
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


SHAPES = [(3240, 256, 256), (1620, 256, 256), (810, 256, 256), (810, 512, 256), (405, 256, 256), (405, 512, 256), (224, 256, 256), (112, 256, 256),
          (224, 512, 256), (224, 1024, 640), (112, 1024, 640), (810, 128, 128), (405, 128, 128)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
for (ntile, C, Cout) in SHAPES:
    rows = (ntile + 255) // 256 * 256
    cp = (Cout + 255) // 256 * 256
    V = torch.randn(36 * rows, C, device=dev)
    U = torch.randn(36 * cp, C, device=dev) * 0.05
    ref = torch.empty(36 * rows, Cout, device=dev)
    out = torch.empty(36 * rows, Cout, device=dev)
    fl = 2.0 * 36 * ntile * C * Cout
    d = ops.make_winograd_gemm_desc(V, U, ref, rows, C, Cout)
    choice = engine.choose_cfg(d.M, Cout, C, 0)
    cfg = engine.apply_choice(d, choice, ws, cnt)
    t_ref = timeit(lambda: ops.conv2d_launch(d, cfg, 0))
    # an un-split launch of the same tile family for the bit-identity check
    d1 = ops.make_winograd_gemm_desc(V, U, ref, rows, C, Cout)
    ops.conv2d_launch(d1, 9 if rows % 128 == 0 else 3, 0)
    torch.cuda.synchronize()
    res = []
    for c in range(8):
        for wgs in (256, 512, 768):
            out.zero_()
            try:
                ops.winograd_gemm(V, U, out, rows, C, Cout, cfg=c, wgs=wgs)
            except RuntimeError:
                continue
            torch.cuda.synchronize()
            same = torch.equal(out, ref)
            err = (out - ref).abs().max().item()
            t = timeit(lambda: ops.winograd_gemm(V, U, out, rows, C, Cout, cfg=c, wgs=wgs))
            res.append((t, c, wgs, same, err))
    res.sort()
    print(f'tiles {ntile} C {C} Cout {Cout}: tuned batched launch cfg{choice} {t_ref:.1f} us {fl / t_ref / 1e6:.0f} TF | persistent: ' +
          '  '.join(f'cfg{c}/{w}: {t:.1f} us {fl / t / 1e6:.0f} TF{"" if s_ else " DIFF %.1e" % e}' for t, c, w, s_, e in res[:6]), flush=True)
