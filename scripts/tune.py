"""Measure the best (tile, split-K) per conv shape on the GPU and write v-floodnet_amd/tuned_gfx950.json
(into gpurun_out/ on the GPU box; copy it into the package afterwards)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['VFN_IGNORE_TUNED'] = '1'
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, synth, engine
dev = torch.device('cuda', 0)
model = AFB_URR(dev, update_bank=True).to(dev).eval()
for (h, w) in [(480, 854), (480, 853), (480, 800)]:
    model.engine().autotune(h, w, 2, iters=5)
os.makedirs('gpurun_out', exist_ok=True)
engine.save_tuned('gpurun_out/tuned_gfx950.json')
print(len(engine._TUNED), 'shapes tuned')
