"""What does bf16 operand rounding (fp32 accumulate) do to the masks?  Oracle-side emulation, CPU."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'oracle'))
import torch, torch.nn.functional as F
import vfloodnet_amd
from tools import synth
import afb_urr_ref as ref

H, W, T, size = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
what = sys.argv[5] if len(sys.argv) > 5 else 'conv,mm'
torch.set_num_threads(8)
sd = synth.make_state_dict(20200212)
frames, m0 = synth.clip(1, T, H, W)
r0 = ref.run_clip(sd, frames, m0, size=size); base, sizes0 = r0['labels'], r0['bank_sizes']

rb = lambda x: x.bfloat16().float()
conv0, mm0 = F.conv2d, torch.matmul
class FP:                       # patched namespaces
    pass
def split(x):
    h = rb(x); return h, rb(x - h)
if 'x3' in what:
    def conv_x3(x, w, b=None, **kw):
        if x.shape[1] <= 5: return conv0(x, w, b, **kw)
        xh, xl = split(x); wh, wl = split(w)
        return conv0(xh + xl, wh + wl, b, **kw) - conv0(xl, wl, None, **kw)
    ref.F.conv2d = conv_x3
    def mm_x3(a, b):
        ah, al = split(a); bh, bl = split(b)
        return mm0(ah + al, bh + bl) - mm0(al, bl)
    ref.torch.matmul = mm_x3
elif 'conv' in what:
    def conv_bf(x, w, b=None, **kw):
        if x.shape[1] <= 5: return conv0(x, w, b, **kw)       # stems stay f32
        return conv0(rb(x), rb(w), b, **kw)
    ref.F.conv2d = conv_bf
if 'mm' in what and 'x3' not in what:
    ref.torch.matmul = lambda a, b: mm0(rb(a), rb(b))
try:
    r1 = ref.run_clip(sd, frames, m0, size=size); out, sizes1 = r1['labels'], r1['bank_sizes']
finally:
    ref.F.conv2d = conv0; ref.torch.matmul = mm0
ious = []
for t in range(1, T):
    a, b = base[t] > 0, out[t] > 0
    ious.append(((a & b).sum().item() + 1e-9) / ((a | b).sum().item() + 1e-9))
print(what, 'IoU(water) per frame min %.4f mean %.4f' % (min(ious), sum(ious) / len(ious)), 'bank', sizes0[-1], sizes1[-1])
print(' '.join('%.3f' % i for i in ious))
