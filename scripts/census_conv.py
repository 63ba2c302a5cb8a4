"""Debug: where a small conv layer's microseconds go (needs the census build: make -C v-floodnet_amd/csrc census; VFN_LIB_PATH=.../libvfn_census.so).
Per workgroup, 100 MHz timestamps at kernel entry, after the first K tile is staged, after the K loop, after the last
store has left.  usage: census_conv.py N,H,W,Cin,Cout,k cfg [ksplit [mode]]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vfloodnet_amd
from vfloodnet_amd import ops, _lib
dev = torch.device('cuda', 0)
N, H, W, Cin, Cout, k = (int(x) for x in sys.argv[1].split(','))
cfg = int(sys.argv[2]); ks = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # 0 f32, 1 bf16, 2 bf16x3 (packed filters, f32 activations)
x = torch.randn(N, H, W, Cin, device=dev)
wp = ops.pad_rows(torch.randn(Cout, k * k * Cin, device=dev) * 0.05)
sc = torch.ones(Cout, device=dev); sh = torch.zeros(Cout, device=dev)
out = torch.empty(N, H, W, Cout, device=dev)
d = ops.make_conv_desc(x, wp, Cout, k, k, 1, k // 2, out, sc, sh, None, True, False)
ws = torch.empty(32 * 1024 * 1024, device=dev)
ops.set_splitk(d, ks, ws if ks > 1 else None)
if mode:
    w_lp = ops.pack_weights_lp(wp, mode)
    ops.use_packed_weights(d, w_lp)
for _ in range(5): ops.conv2d_launch(d, cfg, mode)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.conv2d_launch(d, cfg, mode)
e1.record(); torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(4096 * 8, np.uint64)
L.vfn_debug_conv_census.argtypes = [ctypes.c_void_p]
assert L.vfn_debug_conv_census(buf.ctypes.data_as(ctypes.c_void_p)) == 0
c = buf.reshape(4096, 8)[:, :4].astype(np.float64)
c = c[c[:, 0] > 0]
t0 = c[:, 0].min()
c = (c - t0) / 100.0      # us
tiles = ops.conv_cfg_tiles()[cfg]
print(f'M={d.M} Cout={Cout} K={k*k*Cin} cfg{cfg} {tiles} wk{ops.conv_cfg_wk(cfg)} ksplit {ks}: {e0.elapsed_time(e1)*50:.1f} us per launch (incl. reduce), {len(c)} workgroups')
print('  start      : first %.2f  median %.2f  last %.2f us' % (c[:, 0].min(), np.median(c[:, 0]), c[:, 0].max()))
print('  prologue   : median %.2f  max %.2f us   (entry -> first K tile staged)' % (np.median(c[:, 1] - c[:, 0]), (c[:, 1] - c[:, 0]).max()))
print('  K loop     : median %.2f  max %.2f us' % (np.median(c[:, 2] - c[:, 1]), (c[:, 2] - c[:, 1]).max()))
print('  epilogue   : median %.2f  max %.2f us   (loop end -> last store left)' % (np.median(c[:, 3] - c[:, 2]), (c[:, 3] - c[:, 2]).max()))
print('  last end   : %.2f us after the first start' % c[:, 3].max())
