"""Per layer shape: the direct implicit-GEMM convolution (tuned choice) against Winograd F(4x4, 3x3) (input transform + the 36
batched-filter GEMMs with their best tile configuration + output transform), every 3x3 / stride-1 layer of the C2 / C3 / C5 plans.
usage: tune_winograd.py [HxW ...] (default: the bench sizes; 400x400 = the training sample).  VFN_WINO_GROUP=n: also the shapes of groups of n frames.
Writes gpurun_out/wino_gfx950.json ("M,cin,cout" -> 1 where Winograd is at least 5 % faster) and the GEMM shapes' entries
into gpurun_out/tuned_gfx950.json."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['VFN_WINOGRAD'] = '0'
PREC = os.environ.get('VFN_WINO_PREC', 'fp32')            # 'bf16': the plain-bf16 mode's table (bf16 V / U, persistent bf16 GEMM)
import torch, vfloodnet_amd
from vfloodnet_amd import AFB_URR, engine, ops
dev = torch.device('cuda', 0)
model = AFB_URR(dev, update_bank=True, precision=PREC).to(dev).eval()
MODE = ops.MODES[PREC]
eng = model.engine()
tiles = ops.conv_cfg_tiles()


def time_list(lst, iters=10):
    for l in lst: l()
    torch.cuda.synchronize()
    best = None
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            for l in lst: l()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        best = us if best is None else min(best, us)
    return best


RETUNE = os.environ.get('VFN_WINO_RETUNE', '0') == '1'    # measure the shapes of the shipped table again (new kernels)
SHIPPED = engine._WINO_TABLE if MODE == 0 else engine._WINO_TABLE_BF16
table, report = ({} if RETUNE else dict(SHIPPED)), []          # (otherwise shapes measured before stay as they are)
sizes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(480, 854), (480, 853), (480, 800)]
layers = []          # every ConvLayer of the engine, by identity of its packed filters
def walk(o):
    if isinstance(o, engine.ConvLayer): layers.append(o)
    elif isinstance(o, dict):
        for v in o.values(): walk(v)
    elif isinstance(o, list):
        for v in o: walk(v)
for o in (eng.enc_q, eng.enc_m, eng.dec, eng.keyval): walk(o)
by_w = {l.w.data_ptr(): l for l in layers}
if MODE:                                                  # (reduced-precision descriptors point at the packed operand image)
    for l_ in layers:
        if getattr(l_, 'k', 0) == 3 and l_.cin % 64 == 0:
            by_w[l_.w_lp(MODE).data_ptr()] = l_
GROUP = int(os.environ.get('VFN_WINO_GROUP', '0'))         # also the batched lists of Engine.segment_group for groups of this many frames
for (h, w) in sizes:
    p = eng.plan(h, w, 2)
    if GROUP > 1:
        p.batch_set(GROUP).dec_batch()
    for lst in p.all_lists():
        for l in lst:
            if l.fn is not ops.conv2d_launch or int(l.args[2]) != MODE:
                continue
            d = l.args[0]
            layer = by_w.get(int(d.w or 0))
            if layer is None or d.KH != 3 or d.stride != 1 or d.in_ld != d.Cin or d.Cin % (64 if MODE else 32) or d.Cout < 32 or d.Cout % 4:
                continue
            key = (d.M, d.Cin, d.Cout)
            if key in table:
                continue
            t_dir = time_list([l])
            # the Winograd sequence for the same tensors
            x = torch.randn(d.N, d.H, d.W, d.Cin, device=dev)
            out = torch.empty(d.N, d.H, d.W, d.Cout, device=dev)
            seq = []
            p._ws_cur, p._cnt_cur = p.ws, p.cnt
            p._conv_winograd(seq, layer, x, out, d.N, d.H, d.W, None, bool(d.relu_in), bool(d.relu_out), 'probe', None, 0, MODE)
            dg = seq[1].args[0]
            gkey = (dg.M, dg.Cout, dg.KH * dg.KW * dg.Cin)
            best, t_best = None, None
            for c in ops.wino_gemm_cfg_options(dg.w_batch_rows, dg.Cout):        # the persistent GEMM (round 5)
                try:
                    engine.apply_choice(dg, (c, 1, 0), p.ws, p.cnt)
                    t = time_list([engine.Launch(ops.conv2d_launch, (dg, c, MODE), 'g')], iters=5)
                except RuntimeError:
                    continue
                if t_best is None or t < t_best:
                    best, t_best = (c, 1, 0), t
            for c, (bm, bn) in enumerate(tiles if MODE == 0 else []):
                if dg.w_batch_rows % bm or dg.cout_pad < ((dg.Cout + bn - 1) // bn) * bn or (bn > 128 and dg.Cout < 256):
                    continue
                if ops.conv_cfg_wk(c) > 1 or ops.conv_cfg_tpb(c) > 1:
                    continue
                opts = [(c, 1, 0)]
                blocks = ((dg.M + bm - 1) // bm) * ((dg.Cout + bn - 1) // bn)
                if ops.conv_cfg_kind(c) == 0 and blocks >= 256:
                    opts += [(c, k_, full) for (full, k_, rows) in ops.tail_split_options(dg, bm, bn, 4, 0) if k_ * rows * dg.Cout <= engine.WS_FLOATS and k_ in (2, 4)]
                for opt in opts:
                    try:
                        engine.apply_choice(dg, opt, p.ws, p.cnt)
                        t = time_list([engine.Launch(ops.conv2d_launch, (dg, c, 0), 'g')], iters=5)
                    except RuntimeError:
                        continue
                    if t_best is None or t < t_best:
                        best, t_best = opt, t
            engine.apply_choice(dg, best, p.ws, p.cnt)
            seq[1].args = (dg, best[0], MODE)
            engine._TABLES[MODE][gkey] = best
            t_w = time_list(seq)
            t_in, t_out = time_list([seq[0]]), time_list([seq[2]])
            table[key] = int(t_w < 0.95 * t_dir)
            report.append(dict(layer=l.name, key=key, direct_us=round(t_dir, 1), wino_us=round(t_w, 1), input_us=round(t_in, 1), gemm_us=round(t_best, 1),
                               output_us=round(t_out, 1), gemm_cfg=list(best), use=table[key]))
            print(report[-1], flush=True)
for k_, v_ in SHIPPED.items():
    table.setdefault(k_, v_)                           # (shapes of other frame sizes keep their entries)
os.makedirs('gpurun_out', exist_ok=True)
json.dump({','.join(str(x) for x in k): v for k, v in sorted(table.items())}, open('gpurun_out/wino_gfx950%s.json' % ('' if MODE == 0 else '_bf16'), 'w'), indent=0)
json.dump(report, open('gpurun_out/r05_tune_winograd_report%s.json' % ('' if MODE == 0 else '_bf16'), 'w'), indent=1)
engine.save_tuned('gpurun_out/' + os.path.basename(engine._TABLE_PATHS[MODE]), MODE)
print(sum(table.values()), 'of', len(table), 'shapes use Winograd')
