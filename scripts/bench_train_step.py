"""Wall time of one training step (train.train_step) at the reference's training shape: Water_Image_Train_DS gives
output_size 400, clip_n 6 (train_video_seg.py:93,44-46): frames [6,3,400,400] -> 1 memorize + 5 segmented samples.
usage: python scripts/bench_train_step.py [T H W obj_n steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, train as T
from tools import synth
a = [int(x) for x in sys.argv[1:]]
Tn, H, W, K, steps = (a + [6, 400, 400, 2, 5][len(a):])[:5]
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(20200212)
model = AFB_URR(dev, update_bank=False).to(dev); model.load_state_dict(sd); model.train()
frames, m0 = synth.clip(3, Tn, H, W)
lab = torch.stack([torch.roll(m0.long(), (2 * t, 5 * t), (0, 1)) for t in range(Tn)], 0)
masks = torch.nn.functional.one_hot(lab, K).permute(0, 3, 1, 2).float()
frames, masks = frames.to(dev), masks.to(dev)
opt = T.AdamW(model.named_parameters(), lr=1e-5)
times, losses = [], []
for s in range(steps + 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, unc = T.train_step(model, opt, frames, masks, 0.5)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    times.append(t1 - t0); losses.append(loss); enq = T.last_enqueue_s
# the same steps back to back, as train_model runs them (train_step returns when the loss has arrived, the optimizer's launches may
# still be running; nothing synchronises between steps)
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(steps):
    T.train_step(model, opt, frames, masks, 0.5)
torch.cuda.synchronize(); b2b = (time.perf_counter() - t0) / steps
# phases of one more step
torch.cuda.synchronize(); t0 = time.perf_counter()
model.engine(); torch.cuda.synchronize(); t1 = time.perf_counter()
l, u, g = T.forward_backward(model, frames, masks, 0.5); torch.cuda.synchronize(); t2 = time.perf_counter()
opt.zero_grad(); opt.set_grads(g); opt.step(); torch.cuda.synchronize(); t3 = time.perf_counter()
print(f'train step {Tn}x{H}x{W}, {K} objects: {1e3 * min(times[2:]):.1f} ms/step, {1e3 * b2b:.1f} ms/step back to back (median {1e3 * sorted(times[2:])[len(times[2:]) // 2]:.1f}, max {1e3 * max(times[2:]):.1f}, first {1e3 * times[0]:.0f}; host enqueue of forward+backward {1e3 * enq:.1f} ms); engine rebuild {1e3 * (t1 - t0):.1f}, '
      f'forward+backward {1e3 * (t2 - t1):.1f}, optimizer {1e3 * (t3 - t2):.1f} ms; losses {[round(x, 4) for x in losses]}')
