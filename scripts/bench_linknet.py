"""First-frame bootstrap model (linknet.LinknetB4) at the reference's 416 x 416: time per predict, HIP path vs the torch CPU
restatement (oracle/linknet_ref.py) on the box's host cores."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd.linknet import LinknetB4
from oracle import linknet_ref as R
from tools import synth_linknet as S
dev = torch.device('cuda', 0)
sd = S.make_state_dict()
x = S.frame(3, 416, 416)
m = LinknetB4.from_checkpoint(sd, dev)
xd = x.to(dev)
for _ in range(3):
    p = m.predict(xd)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    p = m.predict(xd)
torch.cuda.synchronize()
t_gpu = (time.perf_counter() - t0) / n
with torch.no_grad():
    R.forward(sd, x)
    t0 = time.perf_counter()
    for _ in range(3):
        ref = R.forward(sd, x)
    t_cpu = (time.perf_counter() - t0) / 3

print(f'LinknetB4 416x416: HIP {1e3 * t_gpu:.2f} ms per predict (~230 small launches, replayed as one HIP graph from the third call on), torch CPU restatement {1e3 * t_cpu:.0f} ms '
      f'({torch.get_num_threads()} threads); max |dprob| {(p.cpu() - ref).abs().max().item():.1e}')
