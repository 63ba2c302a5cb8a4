"""Where the microseconds of a Winograd-domain GEMM launch go (36 x [tiles x C] x [C x Cout] as ONE batched-filter launch):
per-workgroup timestamps of the census build (make -C v-floodnet_amd/csrc census; VFN_LIB_PATH=.../libvfn_census.so):
entry, first K tile staged, K loop done, last store left.  usage: census_wino_gemm.py [tiles C Cout cfg ...]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vfloodnet_amd
from vfloodnet_amd import ops, _lib
dev = torch.device('cuda', 0)
a = [int(x) for x in sys.argv[1:]]
ntile, C, Cout = (a + [3240, 256, 256][len(a):])[:3]
cfgs = a[3:] or [9, 10, 22]
tiles = ops.conv_cfg_tiles()
rows = (ntile + 255) // 256 * 256
M = 36 * rows
V = torch.randn(1, 1, M, C, device=dev)
cp = (Cout + 255) // 256 * 256
U = torch.randn(36 * cp, C, device=dev) * 0.05
out = torch.empty(1, 1, M, Cout, device=dev)
L = _lib.lib()
L.vfn_debug_conv_census.argtypes = [ctypes.c_void_p]
for c in cfgs:
    d = ops.make_conv_desc(V, U, Cout, 1, 1, 1, 0, out, None, None, None, False, False, N=1, H=1, W=M)
    d.cout_pad = cp
    d.w_batch_rows = rows
    d.k_rot = int(os.environ.get('VFN_KROT', '1'))
    for _ in range(3):
        ops.conv2d_launch(d, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv2d_launch(d, c)
    e1.record(); torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, np.uint64)
    assert L.vfn_debug_conv_census(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(4096, 8)[:, :4].astype(np.float64)
    t = t[t[:, 0] > 0]
    t = (t - t[:, 0].min()) / 100.0
    n_wg = (M // tiles[c][0]) * ((Cout + tiles[c][1] - 1) // tiles[c][1])
    print(f'tiles {ntile} C {C} Cout {Cout} cfg{c} {tiles[c]}: {e0.elapsed_time(e1) * 100:.1f} us per launch, {n_wg} workgroups ({len(t)} in the census: the first 4096)')
    print('  start    : median %.2f  p90 %.2f  last %.2f us' % (np.median(t[:, 0]), np.percentile(t[:, 0], 90), t[:, 0].max()))
    for nm, i, j in (('prologue', 0, 1), ('K loop', 1, 2), ('epilogue', 2, 3), ('whole', 0, 3)):
        x = t[:, j] - t[:, i]
        print('  %-9s: median %.2f  p10 %.2f  p90 %.2f  max %.2f us' % (nm, np.median(x), np.percentile(x, 10), np.percentile(x, 90), x.max()))
    # workgroups alive over time (how many rounds; how synchronised the phases are)
    order = np.argsort(t[:, 0])
    first = t[order[:256], :]
    print('  first 256 workgroups: start spread %.2f us, K-loop start spread %.2f us, end spread %.2f us' % (
        first[:, 0].max() - first[:, 0].min(), first[:, 1].max() - first[:, 1].min(), first[:, 3].max() - first[:, 3].min()))
