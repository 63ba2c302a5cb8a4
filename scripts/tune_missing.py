"""Measure the conv shapes of the C2 / C3 / C5 plans that the shipped tables lack (e.g. after a plan gained a layer) and
write the completed tables to gpurun_out/ (copy them into v-floodnet_amd/).  usage: tune_missing.py [fp32|bf16|bf16x3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, engine, ops
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
dev = torch.device('cuda', 0)
model = AFB_URR(dev, update_bank=True, precision=prec).to(dev).eval()
mode = ops.MODES[prec]
before = set(engine._TABLES[mode])
for (h, w) in [(480, 854), (480, 853), (480, 800)]:
    model.engine().autotune(h, w, 2, iters=12, only_missing=True)
new = set(engine._TABLES[mode]) - before
print('new shapes:', {k: engine._TABLES[mode][k] for k in sorted(new)})
os.makedirs('gpurun_out', exist_ok=True)
engine.save_tuned('gpurun_out/' + os.path.basename(engine._TABLE_PATHS[mode]), mode)
