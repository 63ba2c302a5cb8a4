"""Re-measure the kernel choice of every Winograd-domain GEMM (default) or of every 1x1 / stride-1 convolution (argument `pconv`) of the
C2 / C3 / C5 plans (and of the training plan) now that the persistent kernel (vfn_winograd_gemm_f32, configuration ids >= 1000;
vfn_conv1x1_persistent_f32, ids >= 2000) competes with one-workgroup-per-tile launches: the table entry of a shape changes hands only
when the new choice is at least MARGIN faster than the old one, re-timed alternately.
Writes gpurun_out/tuned_gfx950.json + gpurun_out/r05_tune_{wino_gemm,pconv}_report.json."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, engine, ops
MARGIN = float(os.environ.get('VFN_TUNE_MARGIN', 0.97))
WHICH = sys.argv[1] if len(sys.argv) > 1 else 'wino'
dev = torch.device('cuda', 0)
model = AFB_URR(dev, update_bank=True).to(dev).eval()
eng = model.engine()
table = engine._TABLES[0]
report = []


def timeit(d, c, iters=12):
    for _ in range(2):
        ops.conv2d_launch(d, c, 0)
    torch.cuda.synchronize()
    best = None
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.conv2d_launch(d, c, 0)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        best = ms if best is None else min(best, ms)
    return best * 1e3


seen = {}
plans = [eng.plan(h, w, 2) for (h, w) in [(480, 854), (480, 853)]]
if os.environ.get('VFN_TUNE_TRAIN', '1') == '1':
    try:
        tp = eng.plan(400, 400, 2, keep_acts=True)
        tp.batch_set(5)
        plans.append(tp)
    except Exception as e:                                # (the inference shapes are what matters here)
        print('training plan skipped:', repr(e))
for p in plans:
    for lst in p.all_lists():
        for l in lst:
            if l.fn is not ops.conv2d_launch or int(l.args[2]) != 0:
                continue
            d = l.args[0]
            if (WHICH == 'wino' and 'wino_gemm' in (l.name or '')) or (WHICH == 'pconv' and ops.pconv_eligible(d, 0)):
                seen.setdefault((d.M, d.Cout, d.KH * d.KW * d.Cin), (d, p))
for key, (d, p) in sorted(seen.items()):
    old = table.get(key) or engine.choose_cfg(*key, 0)
    cand = []
    for c in (ops.wino_gemm_cfg_options(d.w_batch_rows, d.Cout) if WHICH == 'wino' else ops.pconv_cfg_options(d.Cout)):
        engine.apply_choice(d, (c, 1, 0), p.ws, p.cnt)
        try:
            cand.append((timeit(d, c, 6), c))
        except RuntimeError:
            pass
    cand.sort()
    engine.apply_choice(d, tuple(old), p.ws, p.cnt)
    t_old = timeit(d, old[0])
    best = None
    for t_, c in cand[:3]:                               # the three fastest once more, alternately with the old choice
        engine.apply_choice(d, (c, 1, 0), p.ws, p.cnt)
        t_new = timeit(d, c)
        if best is None or t_new < best[0]:
            best = (t_new, c)
    engine.apply_choice(d, tuple(old), p.ws, p.cnt)
    t_old = min(t_old, timeit(d, old[0]))
    take = best is not None and best[0] < MARGIN * t_old
    fl = 2.0 * key[0] * key[1] * key[2]
    report.append({'shape': list(key), 'old': list(old), 'old_us': round(t_old, 1), 'new': [best[1], 1, 0] if best else None,
                   'new_us': round(best[0], 1) if best else None, 'taken': bool(take), 'old_tf': round(fl / t_old / 1e6, 1),
                   'new_tf': round(fl / best[0] / 1e6, 1) if best else None})
    print(report[-1], flush=True)
    if take:
        table[key] = (best[1], 1, 0) + (tuple(old[:3]) if WHICH == 'pconv' else ())      # (pconv: the old choice stays as the fallback)
os.makedirs('gpurun_out', exist_ok=True)
engine.save_tuned('gpurun_out/tuned_gfx950.json', 0)
json.dump(report, open('gpurun_out/r05_tune_%s_report.json' % ('wino_gemm' if WHICH == 'wino' else 'pconv'), 'w'), indent=1)
print('shapes', len(report), 'moved', sum(r['taken'] for r in report), 'old total %.1f us, new total %.1f us' % (
    sum(r['old_us'] for r in report), sum(r['new_us'] if r['taken'] else r['old_us'] for r in report)))
