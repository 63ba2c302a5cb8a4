import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libvfn_hip_ablate.so')
from vfloodnet_amd import ops
dev = torch.device('cuda', 0)
N, H, W, Cin, Cout, k = 1, 256, 256, 256, 256, 3
x = torch.randn(N, H, W, Cin, device=dev); wp = ops.pad_rows(torch.randn(Cout, k*k*Cin, device=dev)*0.05)
out = torch.empty(N, H, W, Cout, device=dev)
d = ops.make_conv_desc(x, wp, Cout, k, k, 1, 1, out, None, None, None, False, False)
fl = 2.0 * d.M * Cout * k*k*Cin
tiles = ops.conv_cfg_tiles()
for mode, name in [(0, 'full'), (256, 'no loads/LDS writes'), (512, 'no barriers (racy)'), (768, 'MFMA + ds_read only')]:
    d.relu_out = mode
    line = f'{name:24s}'
    for c in (0, 2, 3, 7, 8, 10):
        for _ in range(2): ops.conv2d_launch(d, c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv2d_launch(d, c)
        e1.record(); torch.cuda.synchronize()
        line += f' cfg{c} {tiles[c][0]}x{tiles[c][1]}: {fl / (e0.elapsed_time(e1)*100) / 1e6:6.1f}'
    print(line)
