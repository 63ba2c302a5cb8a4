"""Re-tune the 64-filter layer shapes of the reduced-precision tables against the tall tiles (configurations 22 / 23: 128 x 64 and
256 x 64, 8 waves) that round 3 enabled in those modes; candidates: the shipped choice's tile and 1 / 3 / 22 / 23 / 27 / 35.
usage: tune_thin.py bf16x3|bf16  -> gpurun_out/tuned_gfx950_<mode>.json"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, engine, ops
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
dev = torch.device('cuda', 0)
model = AFB_URR(dev, update_bank=True, precision=prec).to(dev).eval()
mode = ops.MODES[prec]
before = dict(engine._TABLES[mode])
cands = (1, 3, 22, 23, 27, 35)
for (h, w) in [(480, 854), (480, 853), (480, 800)]:
    model.engine().autotune(h, w, 2, iters=20, shape_filter=lambda k: k[1] == 64,
                            cfg_filter=lambda k, c: c in cands or (k in before and c == before[k][0]))
changed = {k: (before.get(k), v) for k, v in engine._TABLES[mode].items() if before.get(k) != v}
print('changed:', changed)
os.makedirs('gpurun_out', exist_ok=True)
engine.save_tuned('gpurun_out/' + os.path.basename(engine._TABLE_PATHS[mode]), mode)
