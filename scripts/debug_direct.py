"""Debug: which channel of B does the direct kernel pair with channel c of A; which rows / cols land where."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfloodnet_amd
from vfloodnet_amd import ops
dev = torch.device('cuda', 0)
N, H, W, Cin, Cout, k = 1, 8, 8, 64, 64, 1
sc = torch.ones(Cout, device=dev); sh = torch.zeros(Cout, device=dev)
wt = (torch.arange(Cin, device=dev).float() + 1).repeat(Cout, 1)          # w[n, c] = c + 1
wp = ops.pad_rows(wt)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 38
pair = []
for c0 in range(Cin):
    x = torch.zeros(N, H, W, Cin, device=dev); x[..., c0] = 1
    y = ops.conv2d_nhwc(x, wp, Cout, k, k, 1, 0, sc, sh, None, False, False, cfg=cfg).reshape(-1, Cout)
    pair.append(y[0, 0].item())
print('A channel c -> value (expect c+1):', pair)
# rows / cols: x[m, 0] = m + 1, w[n, 0] = 1000 * (n + 1) -> y[m, n] = (m + 1) * 1000 * (n + 1)
x = torch.zeros(N, H, W, Cin, device=dev); x.view(-1, Cin)[:, 0] = torch.arange(64, device=dev).float() + 1
wt2 = torch.zeros(Cout, Cin, device=dev); wt2[:, 0] = (torch.arange(Cout, device=dev).float() + 1) * 1000
y = ops.conv2d_nhwc(x, ops.pad_rows(wt2), Cout, k, k, 1, 0, sc, sh, None, False, False, cfg=cfg).reshape(-1, Cout)
print('rows (y[:,0]/1000):', (y[:, 0] / 1000).tolist())
print('cols (y[0,:]/1000):', (y[0, :] / 1000).tolist())
