"""Benchmark of the V-FloodNet video-segmentation hot loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

One *step* = one frame of ``test_video_seg.py:105-115``: bicubic resize (identity at 480p) ->
``segment`` -> object softmax -> ``memorize`` -> ``FeatureBank.update`` -> resize+argmax -> largest-component filter -> label D2H, on a synthetic 480x854 clip (BASELINE.json config C2, fp32) that is resident in HBM when
the timed region starts.  For N > 1 every rank runs its own clip (seed = rank + 1, weak
scaling, no collective on the data path) and the per-clip label masks are exchanged with one
RCCL all-gather inside the timed bracket (BASELINE.json config C4).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline      the dominant kernel (f32-MFMA implicit-GEMM conv): algorithmic FLOP / HIP-event time
                measured on sampled frames of the timed region, against the 157.3 TFLOP/s f32 matrix peak
  memory_read   the fused memory read of the sampled frames: algorithmic FLOP (scores counted once) / HIP-event time
  cpu_baseline  the CPU oracle (oracle/afb_urr_ref.py, torch CPU, all host cores) on the first
                frames of the same clip -- a reported baseline, not the target
  parity        mIoU / max |dprob| of the HIP labels against that oracle run on the same frames.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3        # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_BF16_MATRIX_TFLOPS = 2500.0      # dense bf16 (v_mfma_f32_32x32x16_bf16: 32 cycles per 32x32x16 on 1024 SIMDs at 2.4 GHz)
# useful-FLOP peak per precision mode: bf16x3 spends three bf16 MFMAs per product
PEAKS = {'fp32': PEAK_F32_MATRIX_TFLOPS, 'bf16': PEAK_BF16_MATRIX_TFLOPS, 'bf16x3': PEAK_BF16_MATRIX_TFLOPS / 3}
DTYPES = {'fp32': 'f32', 'bf16': 'bf16 operands, f32 accumulate/storage', 'bf16x3': 'bf16x3 (split-bf16 operands, 3 MFMAs per product), f32 accumulate/storage'}
WAVES = [(2, 2), (2, 2), (2, 2), (2, 2), (1, 2), (2, 1), (4, 1), (4, 2), (2, 4), (4, 2), (2, 4),
         (2, 4), (2, 4), (2, 2), (4, 2), (4, 1), (2, 2), (2, 4), (2, 4), (2, 4)]   # per conv cfg
WORKLOADS = {'C2': (480, 854, 1), 'C3': (720, 1280, 5), 'C5': (1080, 1920, 1)}      # H0, W0, memorize every n-th frame


def miou(a, b):
    v = []
    for c in (0, 1):
        inter = ((a == c) & (b == c)).sum().item()
        union = ((a == c) | (b == c)).sum().item()
        v.append(1.0 if union == 0 else inter / union)
    return sum(v) / 2


class ConvTimer:
    """HIP events around every implicit-GEMM launch of sampled frames (events live on torch's current
    stream, which is the stream every kernel of this package is launched on)."""

    def __init__(self):
        from vfloodnet_amd import ops
        self.ops = ops
        self.orig = ops.conv2d_launch
        self.records = []        # (cfg, flops, ev0, ev1)
        self.active = False

    def install(self):
        ops = self.ops

        def timed(desc, cfg, mode=0):
            if not self.active:
                return self.orig(desc, cfg, mode)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self.orig(desc, cfg, mode)
            e1.record()
            self.records.append((cfg, 2.0 * desc.M * desc.Cout * desc.KH * desc.KW * desc.Cin, e0, e1))
        ops.conv2d_launch = timed
        # launches were bound at plan-build time: rebind
        return timed

    def summary(self):
        per = {}
        for cfg, fl, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            d = per.setdefault(cfg, [0.0, 0.0, 0])
            d[0] += fl
            d[1] += ms
            d[2] += 1
        return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=99)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--budget', type=int, default=250000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-frames', type=int, default=10, help='frames of the CPU baseline sample (after a 2-frame warm-up)')
    ap.add_argument('--autotune', action='store_true',
                    help='re-measure the tile / split-K choice of every conv shape before the run instead of using the shipped '
                         'tables (tuned_gfx950*.json, from scripts/tune.py); shapes the tables lack are always measured')
    ap.add_argument('--no-autotune', action='store_true', help='(default behaviour; kept for old command lines)')
    ap.add_argument('--no-overlap', action='store_true',
                    help="do not run the next frame's query encoder on a side stream under memorize/update")
    ap.add_argument('--sample-every', type=int, default=16, help='time the conv launches on every n-th frame')
    ap.add_argument('--precision', choices=sorted(PEAKS), default='fp32',
                    help='fp32 = BASELINE config C2 (the headline, exact f32); bf16x3 / bf16 = the reduced-precision configs')
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='C2',
                    help='C2: 480x854 clip, every frame memorised; C3: 720x1280 clip (resized to 480p on the device as '
                         'test_video_seg.py:88,107 does), bank grows with every 5th frame; C5: 1920x1080 stream, every frame '
                         'memorised, bank budget sized so that nothing is evicted (use --steps 2000 for the full config)')
    ap.add_argument('--native', action='store_true',
                    help='run the network at the input resolution instead of the reference semantics (resize to a 480-pixel '
                         'short edge, test_video_seg.py:46,107); only meaningful with --workload C3')
    args = ap.parse_args()
    H0, W0, mem_every = WORKLOADS[args.workload]
    if args.workload == 'C5':                      # class_budget = 0.8 * budget / 2 >= steps * HW: the bank only grows
        args.budget = max(args.budget, 2 * int(1.25 * 2 * (args.steps + args.warmup + 2) * 1620) + 4)
    net_size = H0 if args.native else 480
    peak = PEAKS[args.precision]

    import vfloodnet_amd
    from vfloodnet_amd import AFB_URR, synth, ops, dist as vdist
    from vfloodnet_amd.video_seg import ClipRunner
    import torch.distributed as dist

    rank, local_rank, world = vdist.init()
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs a GPU: the hot path is HIP-only')
    # VFN_SINGLE_DEVICE=1 (+ VFN_DIST_BACKEND=gloo): smoke-run the N > 1 code path on a 1-GPU box
    dev = torch.device('cuda', 0 if os.environ.get('VFN_SINGLE_DEVICE') == '1' else local_rank)
    torch.cuda.set_device(dev)

    K, Wm = args.steps, args.warmup
    args.sample_every = max(1, min(args.sample_every, K))       # at least one frame is sampled for the roofline
    if world > 1:       # N ranks share the host: keep the CPU-side weight synthesis of each from waking every core
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 8) // (2 * world))))
    sd = synth.make_state_dict(20200212)
    model = AFB_URR(dev, update_bank=True, precision=args.precision).to(dev).eval()
    model.load_state_dict(sd, strict=True)

    # ---- inputs resident in HBM
    seed = rank + 1
    if args.workload == 'C5':
        n_frames = K + 1                           # a stream never repeats: all frames resident (2001 x 1080p = 50 GB of HBM)
        frames, m0 = synth.clip_on_device(seed, n_frames, H0, W0, dev)
    else:
        n_frames = min(K + 1, 400)                 # (longer runs cycle through the frames)
        frames, m0 = synth.clip(seed, n_frames, H0, W0)
        frames = frames.to(dev)
    onehot = synth.onehot(m0).unsqueeze(0).to(dev)

    timer = ConvTimer()
    timed_launch = timer.install()
    # the memory read (bank scan + apply + finish) of the sampled frames, timed the same way
    from vfloodnet_amd.engine import Engine
    mem_records = []                               # (bank entries summed over objects, HW, ev0, ev1)
    orig_memread = Engine._memory_read

    def timed_memread(self_, p_, fb_, update_bank_):
        if not timer.active:
            return orig_memread(self_, p_, fb_, update_bank_)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_memread(self_, p_, fb_, update_bank_)
        e1.record()
        mem_records.append((sum(fb_._len_host), p_.HW, e0, e1))
    Engine._memory_read = timed_memread
    eng = model.engine()
    from vfloodnet_amd.video_seg import resized_hw
    Hn, Wn = resized_hw(H0, W0, net_size)            # reference semantics: the network always sees the 480p frame
    eng.autotune(Hn, Wn, 2, only_missing=not args.autotune)      # the shipped tables cover C2 / C3 / C5 at reference semantics
    plan = eng.plan(Hn, Wn, 2)
    for lst in (plan.seg_pre, plan.seg_post, plan.mem):
        for l in lst:
            if l.fn is timer.orig:
                l.fn = timed_launch

    # ---- warm-up on a throw-away bank
    warm = ClipRunner(model, 2, args.budget, size=net_size, mem_every=mem_every)
    warm.start(frames[0:1], onehot)
    for t in range(1, Wm + 1):
        warm.step(frames[(t % (n_frames - 1)) + 1:(t % (n_frames - 1)) + 2])
    del warm

    # ---- timed region: exactly K steps
    runner = ClipRunner(model, 2, args.budget, size=net_size, mem_every=mem_every, postprocess=True)     # largest-blob filter (:116) on the device too
    runner.start(frames[0:1], onehot)
    labels = torch.empty(K + 1, H0, W0, dtype=torch.uint8, device=dev)
    labels[0] = m0.to(dev)
    if world > 1:
        # the collective of the timed region once, untimed: RCCL sets up its channels / buffers at the first call of a
        # given collective and size, which would otherwise be charged to the clip
        labels[1:].zero_()
        vdist.gather_masks(labels.unsqueeze(0), world, rank, world)
        torch.cuda.synchronize()
    bank_sum = 0
    bank_sizes = []
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for t in range(1, K + 1):
        idx = ((t - 1) % (n_frames - 1)) + 1
        timer.active = (t % args.sample_every == 0)
        # no prefetch into / out of a sampled frame: its kernels are timed alone on the device
        sampled_next = ((t + 1) % args.sample_every == 0)
        nxt = ((t % (n_frames - 1)) + 1) if (t < K and not args.no_overlap and not timer.active and not sampled_next) else None
        runner.step(frames[idx:idx + 1], want_label=False,
                    next_frame=frames[nxt:nxt + 1] if nxt is not None else None)
        timer.active = False
        labels[t].copy_(runner._label_dev, non_blocking=True)
        bank_sum += sum(runner.bank_sizes())
        bank_sizes.append(runner.bank_sizes())
    if world > 1:
        all_labels = vdist.gather_masks(labels.unsqueeze(0), world, rank, world)   # one RCCL all-gather
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev if (world == 1 or dist.get_backend() == 'nccl') else 'cpu')
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel
    per = timer.summary()
    roof = None
    if per:
        tiles = ops.conv_cfg_tiles()
        dom = max(per, key=lambda c: per[c][1])
        tot_fl = sum(v[0] for v in per.values())
        tot_ms = sum(v[1] for v in per.values())
        fl, ms, n = per[dom]
        ach = fl / (ms * 1e-3) / 1e12
        kname = f'conv_igemm_kernel<{tiles[dom][0]}, {tiles[dom][1]}, {WAVES[dom][0]}, {WAVES[dom][1]}, {ops.MODES[args.precision]}>'
        traffic = None                      # HBM bytes per launch of this kernel from the committed PMC passes
        tname = 'r01_pmc_traffic.json' if args.precision == 'fp32' else f'r01_pmc_traffic_{args.precision}.json'
        tpath = os.path.join(ROOT, 'profiles', tname)
        if os.path.isfile(tpath):
            for k_, v_ in json.load(open(tpath)).get('kernels', {}).items():
                if kname in k_ and args.workload == 'C2':
                    traffic = round(v_['hbm_bytes_per_launch'])
        roof = {'bound': 'mfma', 'kernel': kname,
                'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                'frac': round(ach / peak, 4), 'traffic': traffic,
                'traffic_source': f'profiles/{tname} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH doubled per MI355X_MICROARCH.md)',
                'launches_timed': n, 'avg_launch_us': round(ms * 1e3 / n, 2),
                'timing': 'HIP events around every launch of frames that take no part in the side-stream overlap (kernel alone '
                          'on the device); rocprofv3 counterpart: profiles/r01_kernel_stats_no_overlap.csv (--no-overlap run); '
                          'profiles/r01_kernel_stats.csv is the default command, where overlapped launches run longer',
                'all_conv_achieved': round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                'all_conv_frac': round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4)}

    memread = None
    if mem_records:
        ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in mem_records)
        alg = sum(1280.0 * b * hw for b, hw, _, _ in mem_records)          # 2*(128+512) FLOP per (entry, query): scores once
        done = sum(1536.0 * b * hw for b, hw, _, _ in mem_records)         # as executed: the scores are formed in both passes
        memread = {'kernels': 'bank_scan_kernel<0> + memread_apply kernel + finish', 'frames_timed': len(mem_records),
                   'ms_per_frame': round(ms / len(mem_records), 3),
                   'achieved_algorithmic': round(alg / (ms * 1e-3) / 1e12, 2), 'achieved_executed': round(done / (ms * 1e-3) / 1e12, 2),
                   'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac_algorithmic': round(alg / (ms * 1e-3) / 1e12 / peak, 4),
                   'mean_bank_entries_per_object': round(sum(b for b, _, _, _ in mem_records) / (2.0 * len(mem_records)), 1)}

    # ---- whole-frame roofline: F_min(B) = 538.48 GFLOP + 3072*B*HW
    b_mean = bank_sum / (2.0 * K)
    fps = world * K / elapsed
    HWn = ((Hn + 15) // 16) * ((Wn + 15) // 16)
    f_min = 538.48e9 + 3072.0 * b_mean * 1620
    frame_frac = (fps / world) * f_min / (peak * 1e12)
    f_ref = 666.56e9 + 3072.0 * b_mean * 1620          # op-for-op reference FLOPs (per-object duplicate convs counted)
    frame_frac_ref = (fps / world) * f_ref / (peak * 1e12)

    # ---- CPU baseline + parity on the first frames of the same clip
    cpu = None
    parity = None
    if not args.no_cpu_baseline and world == 1 and args.workload == 'C2':
        from oracle import afb_urr_ref as O
        nthr = min(16, os.cpu_count() or 1)     # fastest setting measured on the GPU box's 256-core host (8/16/32/64/128 tried)
        torch.set_num_threads(nthr)
        n_cpu = min(args.cpu_frames, K)
        fr_cpu = frames[:n_cpu + 1].cpu()
        O.run_clip(sd, fr_cpu[:2], m0, budget=args.budget)                 # warm the CPU kernels
        c0 = time.perf_counter()
        ref = O.run_clip(sd, fr_cpu, m0, budget=args.budget, return_scores=True)
        c1 = time.perf_counter()
        cpu = {'value': round(n_cpu / (c1 - c0), 4), 'unit': 'frames/s', 'cores': nthr, 'kind': 'port',
               'sample': f'first {n_cpu} frames of the same 480x854 clip (incl. first-frame memorize), torch CPU oracle'}
        lab = labels[:n_cpu + 1].cpu()
        parity = {'frames': n_cpu,
                  'miou_vs_oracle': round(min(miou(lab[t], ref['labels'][t]) for t in range(1, n_cpu + 1)), 5),
                  'bank_sizes_equal': bank_sizes[:n_cpu] == ref['bank_sizes']}

    gpath = os.path.join(ROOT, 'tests', 'golden', 'c2_480x854_100.npz')
    if world == 1 and K == 99 and args.workload == 'C2' and os.path.isfile(gpath):
        import numpy as np
        g = np.load(gpath)
        refl = torch.from_numpy(np.unpackbits(g['labels'], axis=-1)[..., :W0])
        lab = labels.cpu()
        ious = [miou(lab[t], refl[t]) for t in range(1, K + 1)]
        parity = dict(parity or {})
        parity.update({'full_clip_frames': K, 'full_clip_miou_min': round(min(ious), 5),
                       'full_clip_miou_mean': round(sum(ious) / len(ious), 5),
                       'full_clip_reference': 'tests/golden/c2_480x854_100.npz (reference model + FeatureBank on CPU, '
                                              'oracle/gen_c2_golden.py)'})

    out = {'metric': 'segmented frames/sec at 480p' if args.workload == 'C2' else f'segmented frames/sec at {H0}p', 'value': round(fps, 3), 'unit': 'frames/s', 'n_gpus': world,
           'steps': K, 'warmup': Wm, 'ms_per_step': round(1e3 * elapsed / K, 3), 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPES[args.precision], 'data': 'synthetic',
           'config': {'workload': f'{args.workload}: {K + 1}-frame {H0}x{W0} synthetic clip per GPU through the test_video_seg.py loop '
                                  f'(' + ('bicubic resize to 480p+' if (Hn, Wn) != (H0, W0) else '') +
                                  f'segment+softmax+memorize' + (f' every {mem_every}th frame' if mem_every > 1 else '') +
                                  f'+bank update+argmax+CCL), {args.precision}, budget {args.budget}',
                      'mean_bank_entries_per_object': round(b_mean, 1),
                      'network_resolution': f'{Hn}x{Wn} ' + ('(native)' if args.native and (Hn, Wn) == (H0, W0) else '(reference semantics: 480-pixel short edge)'),
                      'frame_mfma_frac_Fmin': round(frame_frac, 4) if mem_every == 1 else None,
                      'frame_mfma_frac_Fref_reference_equivalent': round(frame_frac_ref, 4) if mem_every == 1 else None},
           'roofline': roof, 'memory_read': memread, 'cpu_baseline': cpu, 'parity': parity}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
