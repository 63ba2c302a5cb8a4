import sys, os
sys.path.insert(0, os.getcwd())
import torch, vfloodnet_amd
import torch.nn.functional as F
from vfloodnet_amd import _lib
from vfloodnet_amd._lib import ptr, stream
L = _lib.lib()
gpu = torch.device('cuda', 0)
N, H, W, C = 2, 7, 9, 8
x = torch.randn(N, H, W, C, device=gpu).relu()
Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
g = torch.randn(N, Ho, Wo, C, device=gpu)
add = torch.randn(N, H, W, C, device=gpu)
xr = x.permute(0, 3, 1, 2).double().requires_grad_()
(F.max_pool2d(xr, 3, 2, 1) * g.permute(0, 3, 1, 2).double()).sum().backward()
for use_add in (0, 1):
    for mask in (0, 1):
        gx = torch.full_like(x, 777.0)
        torch.cuda.synchronize()
        rc = L.vfn_maxpool3x3s2_backward_f32(ptr(x), ptr(g), ptr(gx), N, H, W, C, ptr(add) if use_add else None, mask, stream())
        torch.cuda.synchronize()
        ref = xr.grad + (add.permute(0, 3, 1, 2).double() if use_add else 0)
        if mask:
            ref = ref * (xr > 0)
        print('add', use_add, 'mask', mask, 'rc', rc, 'untouched', (gx == 777.0).float().mean().item(), 'err', (gx.permute(0, 3, 1, 2).double() - ref).abs().max().item())
