"""Measure the convolution shapes of the TRAINING step (train_video_seg.py's 6 x 400 x 400 sample: forward plans with kept
activations + every data-gradient convolution of the backward pass) that the shipped table lacks, and write the completed table
to gpurun_out/tuned_gfx950.json (copy into v-floodnet_amd/).  The forward shapes are tuned on the plan; the backward's
convolutions are tuned where they are launched (engine.tune_desc on the live descriptor, first time a shape is seen).
usage: tune_train.py [T H W obj_n]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, engine, ops, backward, train as T
from tools import synth
a = [int(x) for x in sys.argv[1:]]
Tn, H, W, K = (a + [6, 400, 400, 2][len(a):])[:4]
dev = torch.device('cuda', 0)
model = AFB_URR(dev, update_bank=False).to(dev); model.load_state_dict(synth.make_state_dict(20200212)); model.train()
frames, m0 = synth.clip(3, Tn, H, W)
lab = torch.stack([torch.roll(m0.long(), (2 * t, 5 * t), (0, 1)) for t in range(Tn)], 0)
masks = torch.nn.functional.one_hot(lab, K).permute(0, 3, 1, 2).float().to(dev)
frames = frames.to(dev)
before = set(engine._TABLES[0])
t0 = time.time()
model.engine().plan(H, W, K).batch_set(Tn - 1).dec_batch()     # (the query encoder and the decoder over all frames of the sample: Engine.query_batch / segment_batch)
model.engine().autotune(H, W, K, iters=8, only_missing=True)
print('forward shapes tuned:', len(set(engine._TABLES[0]) - before), f'{time.time() - t0:.0f} s', flush=True)

ws = torch.empty(engine.WS_FLOATS, device=dev)
cnt = torch.zeros(ops.SK_MAX_TILES, dtype=torch.int32, device=dev)
seen = {}


def tuned_launch(self, d, plan):
    key = (d.M, d.Cout, d.KH * d.KW * d.Cin)
    if key not in engine._TABLES[0]:
        ok = not (d.out_ld % 4 or (d.res and d.res_ld % 4) or (d.mask and d.mask_ld % 4))
        engine._TABLES[0][key] = engine.tune_desc(d, 0, ws, cnt, iters=8, allow_split=ok)
        seen[key] = engine._TABLES[0][key]
    choice = engine.choose_cfg(*key, 0)
    if choice[1] > 1 and (d.out_ld % 4 or (d.res and d.res_ld % 4) or (d.mask and d.mask_ld % 4)):
        choice = (choice[0], 1, 0)
    ops.conv2d_launch(d, engine.apply_choice(d, choice, plan.ws, plan.cnt), 0)


orig = backward.DecoderBackward._launch
backward.DecoderBackward._launch = tuned_launch
t0 = time.time()
T.forward_backward(model, frames, masks, 0.5)
torch.cuda.synchronize()
backward.DecoderBackward._launch = orig
print('backward shapes tuned:', len(seen), f'{time.time() - t0:.0f} s')
for k in sorted(seen):
    print(' ', k, seen[k])
os.makedirs('gpurun_out', exist_ok=True)
engine.save_tuned('gpurun_out/' + os.path.basename(engine._TABLE_PATHS[0]), 0)
# before / after on this box
opt = T.AdamW(model.named_parameters(), lr=1e-5)
for tag in ('tuned',):
    ts = []
    for s in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        T.train_step(model, opt, frames, masks, 0.5)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f'{tag}: {1e3 * min(ts[2:]):.1f} ms/step')
