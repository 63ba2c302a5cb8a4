"""Measure the best (tile, split-K) per conv shape on the GPU and write the tuned table of a precision mode
(into gpurun_out/ on the GPU box; copy it into v-floodnet_amd/ afterwards).  usage: tune.py [fp32|bf16|bf16x3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['VFN_IGNORE_TUNED'] = '1'
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, engine, ops
from tools import synth
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
dev = torch.device('cuda', 0)
model = AFB_URR(dev, update_bank=True, precision=prec).to(dev).eval()
for (h, w) in [(480, 854), (480, 853), (480, 800)]:
    model.engine().autotune(h, w, 2, iters=int(os.environ.get('VFN_TUNE_ITERS', 12)))
os.makedirs('gpurun_out', exist_ok=True)
# (the bf16 engine runs its 32-channel layers in bf16x3: those shapes live in the bf16x3 table, written by the bf16x3 run)
mode = ops.MODES[prec]
name = os.path.basename(engine._TABLE_PATHS[mode])
engine.save_tuned('gpurun_out/' + name, mode)
print(name, len(engine._TABLES[mode]), 'shapes tuned')
