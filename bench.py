"""Benchmark of the V-FloodNet video-segmentation hot loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

One *step* = one frame of ``test_video_seg.py:105-116``: bicubic resize (identity at 480p) -> ``segment`` ->
object softmax -> ``memorize`` -> ``FeatureBank.update`` -> resize+argmax -> largest-component filter -> label D2H,
on a synthetic clip that is resident in HBM when the timed region starts.

Workload (BASELINE.json config C2, fp32): the **100-frame 480x854 clip**, whatever ``--steps`` says.
  * ``--steps 99`` (default): all 99 loop iterations of the clip are timed.
  * ``--steps K < 99``: the clip still runs from its first frame; frames ``1 .. s-1`` are an untimed pre-roll
    (they build the bank exactly as the reference loop would), the K frames ``s .. s+K-1`` are timed -- ``s`` is chosen
    so that the mean bank size of the timed frames equals the full-clip mean (the memory read and the bank update
    cost grows with the bank) -- and the clip is then finished untimed.  ``full_clip_fps`` reports all 99 iterations.
``--warmup W`` steps on a throw-away bank precede everything (the contract's warm-up; repeated until the device has
been busy for ``--min-warm-s`` so the timed region never starts on an idle-clocked GPU).

``--gpus N``: every rank runs its own clip (seed = rank + 1, weak scaling, no collective on the data path) and the
per-clip label masks are exchanged with one RCCL all-gather inside the timed bracket (config C4).  Under
``torch.distributed.run`` the ranks come from the environment; without it (``WORLD_SIZE`` unset) this script starts
the N ranks itself as child processes *before anything touches the GPU* and rank 0 prints the line.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline      the dominant kernel (f32-MFMA implicit-GEMM conv): algorithmic FLOP / HIP-event time
                measured on sampled frames of the timed region, against the 157.3 TFLOP/s f32 matrix peak
  memory_read   the fused memory read of the sampled frames: algorithmic FLOP (scores counted once) / HIP-event time
  cpu_baseline  the CPU oracle (oracle/afb_urr_ref.py, torch CPU) on the first frames of the same clip -- a
                reported baseline, not the target
  parity        mIoU of the HIP labels against that oracle run and against the committed reference-generated clip.
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
WORKLOADS = {'C2': (480, 854, 1), 'C3': (720, 1280, 5), 'C5': (1080, 1920, 1)}      # H0, W0, memorize every n-th frame
CLIP_FRAMES = 100                     # BASELINE.json configs[1]: "100-frame 480p synthetic clip"
PROFILE_ROUND = 'r06'


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
        self.records = []        # (kernel key, flops, ev0, ev1)
        self.active = False
        self.light = False       # this frame: only the memory read's apply kernel is bracketed (behind an idle side stream)

    def install(self):
        def timed(desc, cfg, mode=0):
            if not self.active:
                return self.orig(desc, cfg, mode)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self.orig(desc, cfg, mode)
            e1.record()
            self.records.append((cfg, 2.0 * desc.M * desc.Cout * desc.KH * desc.KW * desc.Cin, e0, e1))
        self.ops.conv2d_launch = timed
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=99)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--min-warm-s', type=float, default=1.0,
                    help='keep repeating the warm-up steps until the device has been busy this long (clock ramp after the '
                         'CPU-only weight synthesis); 0 = exactly --warmup steps')
    ap.add_argument('--min-timed-s', type=float, default=2.0,
                    help='after the clip, keep cycling through its frames (untimed for `value`) until the GPU has run the loop '
                         'for this long in total; reported as `sustained` (bank at its budget: the eviction regime). 0 = off')
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
    ap.add_argument('--apply-sample-every', type=int, default=None,
                    help='additionally bracket ONLY the dominant kernel (the memory read\'s apply launch) with HIP events on every '
                         'n-th timed frame -- the main stream first waits for the side stream to go idle, so the kernel is alone on '
                         'the device, and the rest of the frame keeps its overlap (a fully sampled frame costs about 2 ms of the '
                         'timed region, such a frame about 0.2).  Default: 5 when --steps < 48 (the driver\'s 20 steps then give '
                         'roofline.launches_timed = 5), else 0 = off')
    ap.add_argument('--precision', choices=sorted(PEAKS), default='fp32',
                    help='fp32 = BASELINE config C2 (the headline, exact f32); bf16x3 / bf16 = the reduced-precision configs')
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='C2',
                    help='C2: 100-frame 480x854 clip, every frame memorised; C3: 100-frame 720x1280 clip (resized to 480p on '
                         'the device as test_video_seg.py:88,107 does), bank grows with every 5th frame; C5: 1920x1080 stream of '
                         '--steps frames, every frame memorised, bank budget sized so that nothing is evicted (--steps 2000 for '
                         'the full config)')
    ap.add_argument('--native', action='store_true',
                    help='run the network at the input resolution instead of the reference semantics (resize to a 480-pixel '
                         'short edge, test_video_seg.py:46,107); only meaningful with --workload C3')
    ap.add_argument('--group', action='store_true',
                    help='workloads that memorise only every n-th frame (C3: n = 5): the frames between two memorize calls as ONE batched pass '
                         '(ClipRunner.launch_group / AFB_URR.segment_group); same frames, same bank updates, labels equal up to summation order')
    ap.add_argument('--main-loop', type=int, default=0, metavar='FRAMES',
                    help='also run the files-to-files loop (vfloodnet_amd.video_seg.main: JPEG frames on disk -> mask + overlay PNG files on '
                         'disk) on a clip of this many 480p frames in a CHILD process after the measurement and report its frames/s under '
                         '`extra` (scripts/main_throughput.py; about 40 s; rank 0 at N = 1 only)')
    ap.add_argument('--clip', choices=['easy', 'hard'], default='easy',
                    help="easy (default, every published line): tools/synth.clip -- tinted, textured water; hard: tools/synth.clip_hard -- water "
                         "that differs from land by texture only (the frames a checkpoint from scripts/train_ckpt.py hard was trained on; a "
                         "hard-task checkpoint on the easy clip is out of its distribution: its f32 margins collapse and so does any parity number)")
    ap.add_argument('--checkpoint', default=None,
                    help='weights from a checkpoint file in the reference\'s schema ({"model": state_dict, ...}, train_video_seg.py:159-177) '
                         'instead of the synthetic recipe (tools/synth.make_state_dict) -- e.g. the checkpoint scripts/bf16_trained_margins.py '
                         'trains on the GPU box with the HIP training step; the line says so in `data`')
    ap.add_argument('--launch-check', action='store_true',
                    help='bring the N ranks up, run the mask all-gather on a dummy clip and print the line skeleton without '
                         'touching the GPU (tests/test_bench_launch.py: the launch logic on a CPU-only machine)')
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launch
def self_launch(args, argv):
    """``python bench.py --gpus N`` without torchrun: start N ranks as *child processes* (this parent never touches
    the GPU: no exec of an initialised process, see the harness notes), one per device, rank 0 prints the line.
    All children are polled: the first one that exits non-zero takes the others down (``dist.spawn_ranks``) instead of
    leaving them in a barrier until the RCCL timeout.  The seam in the reference is the sequential loop of
    ``scripts/batch_test_video_seg.py:40-47``."""
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import dist as vdist
    return vdist.spawn_ranks([sys.executable, os.path.abspath(__file__)] + list(argv), args.gpus)


def launch_check(args):
    """The distributed bring-up + the one collective of the run, on host tensors (``gloo``)."""
    from vfloodnet_amd import dist as vdist
    import torch.distributed as dist
    rank, local_rank, world = vdist.init(backend='gloo' if not torch.cuda.is_available() else None)
    lab = torch.full((1, 4, 6, 8), rank + 1, dtype=torch.uint8)
    allm = vdist.gather_masks(lab, world, rank, world)
    ok = all(int(allm[c].min()) == c + 1 and int(allm[c].max()) == c + 1 for c in range(world))
    if vdist.active(world):
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'launch_check': True, 'n_gpus': world, 'gather_ok': bool(ok), 'gpus_arg': args.gpus}))
    return 0 if ok else 1


# ------------------------------------------------------------------------------------------------ timed window
def pick_window(K, n_iter, golden_sizes=None):
    """First timed iteration ``s`` (1-based) of a K-frame window inside a clip of ``n_iter`` iterations whose mean bank
    size is closest to the full-clip mean.  ``golden_sizes`` = per-iteration bank sizes of the reference's own run of the
    same clip (tests/golden/c2_480x854_100.npz) when available; the bank grows almost linearly, so without it the
    centred window is used."""
    if K >= n_iter:
        return 1
    if golden_sizes is not None and len(golden_sizes) == n_iter:
        tot = [float(sum(x)) for x in golden_sizes]
        full = sum(tot) / n_iter
        best, best_d = 1, None
        for s in range(1, n_iter - K + 2):
            d = abs(sum(tot[s - 1:s - 1 + K]) / K - full)
            if best_d is None or d < best_d:
                best, best_d = s, d
        return best
    return 1 + (n_iter - K) // 2


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        return self_launch(args, argv)
    if env_world is not None and int(env_world) != args.gpus and '--gpus' in ' '.join(argv):
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks')
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import dist as vdist
    # each rank keeps to its own slice of the host's CPUs (before any GPU call / thread pool): N ranks on one host otherwise
    # wake every core with every torch CPU op and starve each other's launch threads
    pinned = vdist.pin_rank_threads(int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('LOCAL_WORLD_SIZE', env_world or 1)))
    if os.environ.get('VFN_BENCH_FAIL_RANK') == os.environ.get('RANK', '0') and env_world is not None:
        return 7                                    # (tests/test_batch_gloo.py: a rank that dies at bring-up)
    if args.launch_check:
        return launch_check(args)

    H0, W0, mem_every = WORKLOADS[args.workload]
    stream_mode = args.workload == 'C5'
    if stream_mode:                                # class_budget = 0.8 * budget / 2 >= steps * HW: the bank only grows
        args.budget = max(args.budget, 2 * int(1.25 * 2 * (args.steps + args.warmup + 2) * 1620) + 4)
    net_size = H0 if args.native else 480
    peak = PEAKS[args.precision]

    from vfloodnet_amd import AFB_URR, ops
    from vfloodnet_amd.video_seg import ClipRunner, resized_hw
    from vfloodnet_amd.engine import Engine
    from tools import synth
    import torch.distributed as dist

    rank, local_rank, world = vdist.init()
    dist_on = vdist.active(world)       # the collectives run: N > 1 ranks, or VFN_FORCE_DIST=1 (one rank through the RCCL branches)
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs a GPU: the hot path is HIP-only')
    # VFN_SINGLE_DEVICE=1 (+ VFN_DIST_BACKEND=gloo): smoke-run the N > 1 code path on a 1-GPU box
    dev = torch.device('cuda', 0 if os.environ.get('VFN_SINGLE_DEVICE') == '1' else local_rank)
    torch.cuda.set_device(dev)

    K, Wm = args.steps, args.warmup
    args.sample_every = max(1, min(args.sample_every, K))       # at least one frame is sampled for the roofline
    apply_every = args.apply_sample_every if args.apply_sample_every is not None else (5 if K < 48 else 0)
    if world > 1:       # N ranks share the host: keep the CPU-side weight synthesis of each from waking every core
        torch.set_num_threads(len(pinned) if pinned else vdist.host_threads_per_rank(world))
    if args.checkpoint:
        sd = torch.load(args.checkpoint, map_location='cpu')['model']
    else:
        sd = synth.make_state_dict(20200212)
    model = AFB_URR(dev, update_bank=True, precision=args.precision).to(dev).eval()
    model.load_state_dict(sd, strict=True)

    # ---- inputs resident in HBM
    seed = rank + 1
    if stream_mode:
        n_frames = K + 1                           # a stream never repeats: all frames resident (2001 x 1080p = 50 GB of HBM)
        frames, m0 = (synth.clip_on_device(seed, n_frames, H0, W0, dev) if args.clip == 'easy' else
                      synth.clip_hard(seed, n_frames, H0, W0, device=dev, in_place=True))
    else:
        n_frames = CLIP_FRAMES                     # the BASELINE clip, whatever --steps is (longer runs cycle through it)
        frames, m0 = synth.clip(seed, n_frames, H0, W0) if args.clip == 'easy' else synth.clip_hard(seed, n_frames, H0, W0)
        frames = frames.to(dev)
    n_iter = n_frames - 1
    onehot = synth.onehot(m0).unsqueeze(0).to(dev)

    golden_sizes = None
    gpath = os.path.join(ROOT, 'tests', 'golden', 'c2_480x854_100.npz')
    golden = None
    if args.workload == 'C2' and os.path.isfile(gpath):
        import numpy as np
        golden = np.load(gpath)
        golden_sizes = golden['bank_sizes'].tolist()     # (the bank of every seed's clip grows alike: one window for all ranks)
    s_first = 1 if stream_mode else pick_window(K, n_iter, golden_sizes)
    last_iter = max(n_iter, s_first + K - 1)        # K > 99: the timed region cycles through the clip's frames

    timer = ConvTimer()
    timed_launch = timer.install()
    # the memory read (bank scan + apply + finish) of the sampled frames, timed the same way
    mem_records = []                               # (bank entries summed over objects, HW, ev0, ev1)
    orig_memread = Engine._memory_read

    apply_records = []                             # the apply kernel of the same frames on its own: (entries, HW, ev0, ev1)
    from vfloodnet_amd import _lib as vlib
    L_ = vlib.lib()
    orig_apply = L_.vfn_memread_apply
    cur_mem = {}

    def timed_apply(desc_ref, stream_):
        if not (timer.active or timer.light):
            return orig_apply(desc_ref, stream_)
        if timer.light:                              # (not a sampled frame: the query side of later frames may still be running)
            busy = getattr(eng, '_side_busy', None)
            if busy is not None:
                torch.cuda.current_stream().wait_event(busy)
            cur_mem.update(entries=sum(runner.fb._len_host), HW=plan.HW)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = orig_apply(desc_ref, stream_)
        e1.record()
        stored = bool(getattr(getattr(desc_ref, '_obj', None), 'scores', None))      # scores read back from the scan
        apply_records.append((cur_mem['entries'], cur_mem['HW'], e0, e1, stored))
        return rc
    L_.vfn_memread_apply = timed_apply

    def timed_memread(self_, p_, fb_, update_bank_, kv_q_=None, out=None):
        if not timer.active:
            return orig_memread(self_, p_, fb_, update_bank_, kv_q_, out)
        cur_mem.update(entries=sum(fb_._len_host), HW=p_.HW)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_memread(self_, p_, fb_, update_bank_, kv_q_, out)
        e1.record()
        mem_records.append((sum(fb_._len_host), p_.HW, e0, e1))
    Engine._memory_read = timed_memread
    eng = model.engine()
    Hn, Wn = resized_hw(H0, W0, net_size)            # reference semantics: the network always sees the 480p frame
    eng.autotune(Hn, Wn, 2, only_missing=not args.autotune)      # the shipped tables cover C2 / C3 / C5 at reference semantics
    plan = eng.plan(Hn, Wn, 2)
    group_n = mem_every if (args.group and mem_every > 1) else 0
    if group_n:                                      # the batched lists exist before the tuner and the timers look at the plan
        plan.batch_set(group_n).dec_batch()
        eng.autotune(Hn, Wn, 2, only_missing=True)
    for lst in plan.all_lists():
        for l in lst:
            if l.fn is timer.orig:
                l.fn = timed_launch

    def frame_of(t):                                 # loop iteration t >= 1 -> frame index (cycling past the clip's end)
        return ((t - 1) % n_iter) + 1

    # ---- warm-up on a throw-away bank: W steps, repeated until the device has been busy for --min-warm-s
    warm = ClipRunner(model, 2, args.budget, size=net_size, mem_every=mem_every, postprocess=True)
    warm.start(frames[0:1], onehot)
    w0 = time.perf_counter()
    warm_steps = 0
    while warm_steps < Wm or (Wm > 0 and time.perf_counter() - w0 < args.min_warm_s and warm_steps < 400):
        warm_steps += 1
        warm.step(frames[frame_of(warm_steps):frame_of(warm_steps) + 1], want_label=False)
    del warm

    # ---- the clip: pre-roll (untimed) | exactly K timed steps | rest of the clip (untimed)
    runner = ClipRunner(model, 2, args.budget, size=net_size, mem_every=mem_every, postprocess=True, capture_graphs=True)   # largest-blob filter (:116) on the device too
    runner.group_capture = group_n
    runner.start(frames[0:1], onehot)
    n_lab = last_iter + 1
    labels = torch.empty(n_lab, H0, W0, dtype=torch.uint8, device=dev)        # what the loop emits (after :116)
    labels_raw = torch.empty(n_lab, H0, W0, dtype=torch.uint8, device=dev)    # before post-processing (parity vs golden)
    labels[0] = m0.to(dev)
    labels_raw[0] = labels[0]
    if dist_on:
        # the collective of the timed region once, untimed: RCCL sets up its channels / buffers at the first call of a
        # given collective and size, which would otherwise be charged to the clip
        labels[1:].zero_()
        vdist.gather_masks(labels[s_first:s_first + K].unsqueeze(0), world, rank, world)
        torch.cuda.synchronize()
    bank_sizes = []
    frame_ms = []

    last_collect = [0.0]

    def collect_one():
        runner.collect()
        bank_sizes.append(runner.bank_sizes())
        now = time.perf_counter()
        frame_ms.append((len(bank_sizes), 1e3 * (now - last_collect[0])))
        last_collect[0] = now

    def run_iters(t_from, t_to, sampling):
        """Frames t_from..t_to, pipelined across frames as video_seg.main does: frame t+1 is enqueued before the host
        waits for frame t (ClipRunner.launch / collect), so the device never idles on the host between frames.  The
        caller's bracket() drains the pipeline, so every timed region still contains exactly its own frames."""
        last_collect[0] = time.perf_counter()
        for t in range(t_from, t_to + 1):
            idx = frame_of(t)
            # no prefetch into / out of a sampled frame: its kernels are timed alone on the device.  Otherwise the query
            # side looks up to three frames ahead (two frames per batched pass, ClipRunner._look_ahead)
            def is_sampled(u):
                return s_first <= u <= s_first + K - 1 and ((u - s_first + 1) % args.sample_every == 0)
            timer.active = sampling and is_sampled(t)
            eng.eager = timer.active                 # (a sampled frame's launches are bracketed one by one: no graph replay)
            timer.light = (sampling and not timer.active and apply_every > 0 and s_first <= t <= s_first + K - 1
                           and (t - s_first + 1) % apply_every == 0)
            nxt = []
            if not args.no_overlap and not timer.active:
                for u in range(t + 1, min(max(last_iter, t_to), t + 3) + 1):
                    if is_sampled(u):
                        break
                    nxt.append(frames[frame_of(u):frame_of(u) + 1])
            # want_label: the frame's label map goes to pinned host memory (the reference's .cpu(), test_video_seg.py:115)
            runner.launch(frames[idx:idx + 1], next_frames=nxt, want_label=True)
            timer.active = timer.light = False
            eng.eager = False
            if t < n_lab:                            # device-side copies for the parity checks after the run
                labels[t].copy_(runner.label_device(), non_blocking=True)
                labels_raw[t].copy_(runner._label_dev, non_blocking=True)
            if len(runner._pending) == 2:
                collect_one()
        while runner._pending:
            collect_one()

    def collect_group_one():
        g_before = len(runner.size_log)
        runner.collect_group()
        new_sizes = runner.size_log[g_before:]
        now = time.perf_counter()
        for sz in new_sizes:
            bank_sizes.append(list(sz))
            frame_ms.append((len(bank_sizes), 1e3 * (now - last_collect[0]) / len(new_sizes)))
        last_collect[0] = now

    def run_groups(t_from, t_to, sampling):
        """--group: frames t_from..t_to as groups that end on a memorised frame (ClipRunner.launch_group), one group in flight while
        the host collects the one before; the next group's frame-only side is prefetched on the side stream.  A group that holds
        a sampled frame runs eagerly with its launches timed, nothing prefetched into or out of it."""
        last_collect[0] = time.perf_counter()

        def group_at(t):
            return min(group_n - (t - 1) % group_n, t_to - t + 1)

        def is_sampled_group(t, g):
            # (a sampled group runs G frames eagerly: every (sample_every x G)-th frame picks one, so about as many FRAMES are instrumented
            # as in the frame-by-frame loop; at least one group of a short window)
            every = min(args.sample_every * group_n, max(group_n, (K // group_n) * group_n))
            return sampling and any(s_first <= u <= s_first + K - 1 and ((u - s_first + 1) % every == 0) for u in range(t, t + g))
        t = t_from
        while t <= t_to:
            g = group_at(t)
            timer.active = is_sampled_group(t, g) and g == group_n
            eng.eager = timer.active
            nxt = None
            t2 = t + g
            if not args.no_overlap and not timer.active and t2 <= t_to:
                g2 = group_at(t2)
                if not (is_sampled_group(t2, g2) and g2 == group_n):
                    nxt = [frames[frame_of(u):frame_of(u) + 1] for u in range(t2, t2 + g2)]
            runner.launch_group([frames[frame_of(u):frame_of(u) + 1] for u in range(t, t + g)], want_label=True, next_frames=nxt)
            timer.active = False
            eng.eager = False
            for i, fo in enumerate(runner._glast['frames'][:g]):
                if t + i < n_lab:
                    labels[t + i].copy_(fo['post'], non_blocking=True)
                    labels_raw[t + i].copy_(fo['label'], non_blocking=True)
            t = t2
            if len(runner._gpending) == 2:
                collect_group_one()
        while runner._gpending:
            collect_group_one()

    if group_n:
        run_iters = run_groups

    def bracket():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        return time.perf_counter()

    import gc
    gc.collect()
    gc.disable()            # a cyclic collection in the middle of the timed frames is a 10-70 ms hiccup of the launch thread
    clip0 = bracket()
    run_iters(1, s_first - 1, False)                                       # untimed pre-roll
    captures0 = sum(pl.graphs.captures for pl in eng.plans.values())
    t0 = bracket()
    run_iters(s_first, s_first + K - 1, True)                              # ---- exactly K timed steps
    captures_in_window = sum(pl.graphs.captures for pl in eng.plans.values()) - captures0
    gather_s = 0.0
    torch.cuda.synchronize()
    g0 = time.perf_counter()                                               # (this rank's K frames are done: compute time = g0 - t0)
    if dist_on:
        vdist.gather_masks(labels[s_first:s_first + K].unsqueeze(0), world, rank, world)   # one RCCL all-gather
        torch.cuda.synchronize()
        gather_s = time.perf_counter() - g0
    t1 = bracket()
    run_iters(s_first + K, last_iter, False)                               # rest of the clip, untimed
    if dist_on:
        vdist.gather_masks(labels[1:].unsqueeze(0), world, rank, world)
    clip1 = bracket()
    # the int32 [T, obj_n] bank-size vector of every rank's clip travels beside the masks: one more, tiny
    # all-gather, outside every timed bracket
    sizes_all = vdist.gather_bank_sizes([bank_sizes[:last_iter]], world, rank, world, dev)

    # ---- sustained rate beyond the clip: the loop keeps cycling through the clip's frames with the bank at its budget
    # (every update evicts -- the regime a stream longer than 100 frames lives in); not part of `value`
    sustained = None
    if args.min_timed_s > 0 and not stream_mode and (t1 - t0) < args.min_timed_s:
        n_extra = int(min(2000, max(K, (args.min_timed_s - (t1 - t0)) / max(1e-4, (t1 - t0) / K))))
        n_before = len(bank_sizes)
        run_iters(last_iter + 1, last_iter + n_extra, False)
        s1 = bracket()
        ext = bank_sizes[n_before:]
        sustained = {'frames': n_extra, 'seconds': round(s1 - clip1, 3), 'fps_this_rank': round(n_extra / (s1 - clip1), 3),
                     'mean_bank_entries_per_object': round(sum(sum(x) for x in ext) / (2.0 * len(ext)), 1),
                     'replaced_entries_total': [int(v) for v in runner.fb.replace_n],
                     'regime': f'frames {last_iter + 1}-{last_iter + n_extra} of the cycled clip, bank at class_budget '
                               f'({int(runner.fb.class_budget)} entries/object): every update evicts'}

    gc.enable()

    def max_over_ranks(x):
        v = torch.tensor([x], dtype=torch.float64, device=dev if (not dist_on or dist.get_backend() == 'nccl') else 'cpu')
        if dist_on:
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
        return float(v.item())
    elapsed = max_over_ranks(t1 - t0)
    clip_elapsed = max_over_ranks(clip1 - clip0)

    def all_ranks(vals):
        """[world][len(vals)] on every rank: one small all-gather (outside every timed region)."""
        on_dev = (not dist_on or dist.get_backend() == 'nccl')
        v = torch.tensor(vals, dtype=torch.float64, device=dev if on_dev else 'cpu')
        if not dist_on:
            return [v.tolist()]
        parts = [torch.empty_like(v) for _ in range(world)]
        dist.all_gather(parts, v)
        return [p_.tolist() for p_ in parts]
    per_rank = all_ranks([g0 - t0, gather_s, float(torch.cuda.max_memory_allocated(dev))])
    dist_info = {'backend': (dist.get_backend() if dist_on else None),
                 'world_size': (dist.get_world_size() if dist_on else 1),
                 'forced_single_rank_group': bool(dist_on and world == 1),
                 'device_of_rank0': torch.cuda.get_device_name(dev)}

    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return 0

    # ---- roofline of the dominant kernel: the instrumented kernel with the most device time on the sampled frames
    # (every implicit-GEMM instantiation and the memory-read apply kernel; the others are listed in roofline.kernels)
    per = timer.summary()
    roof = None
    if per:
        cands = []                          # (name, flops, ms, launches, frames the launches were sampled on)
        n_full = max(1, len(mem_records))   # fully sampled frames (every instrumented launch bracketed)
        for c, (fl, ms, n) in per.items():
            cands.append((ops.conv_cfg_name(c, ops.MODES[args.precision]), fl, ms, n, n_full))
        if apply_records:
            a_ms = sum(r_[2].elapsed_time(r_[3]) for r_ in apply_records)
            # (2*128 + 2*512) FLOP per (entry, query): scores + P^T V; with the scores read back from the statistics scan
            # (f32 default) the kernel's own work is P^T V alone: 2*512
            stored = all(r_[4] for r_ in apply_records)
            a_fl = sum((1024.0 if r_[4] else 1280.0) * r_[0] * r_[1] for r_ in apply_records)
            # bf16x3 reads the bank's kept split-bf16 image (FeatureBank.lp_image) unless VFN_LP_IMAGE=0
            img = os.environ.get('VFN_LP_IMAGE', '1') != '0'
            x3 = 'memread_apply_shw_kernel<true>' if img else 'memread_apply_lpw_kernel<true>'
            # (plain bf16 takes the image kernel too unless VFN_APPLY_IMG_BF16=0; lines written before the end of round 5 name
            # memread_apply_lpw_kernel<false> here while rocprof shows memread_apply_shw_kernel<false>: profiles/r05_kernel_stats_bf16.csv)
            # (round 6: plain bf16 on the kept image runs the software-pipelined kernel unless VFN_APPLY_PIPE=0)
            b1 = (('memread_apply_pipe_kernel' if os.environ.get('VFN_APPLY_PIPE', '1') != '0' else 'memread_apply_shw_kernel<false>')
                  if (img and os.environ.get('VFN_APPLY_IMG_BF16', '1') != '0') else 'memread_apply_lpw_kernel<false>')
            kn = {'fp32': 'memread_apply_ss_kernel' if stored else 'memread_apply_wide_kernel', 'bf16': b1, 'bf16x3': x3}[args.precision]
            cands.append((kn, a_fl, a_ms, len(apply_records), len(apply_records)))      # (one launch per frame)
        tot_fl = sum(v[0] for v in per.values())
        tot_ms = sum(v[1] for v in per.values())
        # dominant = most device time PER FRAME (the apply kernel is bracketed on more frames than the convolutions)
        all_ms = sum(c_[2] / c_[4] for c_ in cands)
        kname, fl, ms, n, nfr = max(cands, key=lambda c_: c_[2] / c_[4])
        ach = fl / (ms * 1e-3) / 1e12
        tname = f'{PROFILE_ROUND}_pmc_traffic.json' if args.precision == 'fp32' else f'{PROFILE_ROUND}_pmc_traffic_{args.precision}.json'
        tpath = os.path.join(ROOT, 'profiles', tname)
        if not os.path.isfile(tpath):       # (this round's counter passes are not committed yet: the previous round's, named as such)
            tname = tname.replace(PROFILE_ROUND, 'r05')
            tpath = os.path.join(ROOT, 'profiles', tname)
        pmc = json.load(open(tpath)).get('kernels', {}) if os.path.isfile(tpath) else {}

        def traffic_of(name_):              # HBM bytes per launch of this kernel from the committed PMC passes
            for k_, v_ in pmc.items():
                if name_ in k_ and args.workload == 'C2' and 'hbm_bytes_per_launch' in v_:
                    return round(v_['hbm_bytes_per_launch'])
            return None
        roof = {'bound': 'mfma', 'kernel': kname,
                'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                'frac': round(ach / peak, 4), 'traffic': traffic_of(kname),
                'traffic_source': f'profiles/{tname} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this command, '
                                  f'FETCH doubled per MI355X_MICROARCH.md; not re-measured in this run)',
                'launches_timed': n, 'avg_launch_us': round(ms * 1e3 / n, 2),
                'share_of_instrumented_time': round(ms / nfr / all_ms, 4),
                'algorithmic_flop': '2*M*Cout*K per conv launch (a Winograd-domain GEMM launch: the FLOP it executes, 2*36*tiles*Cin*Cout); '
                                    'memory-read apply launch: 1024 * bank entries * HW (P^T V; the scores come from the '
                                    'statistics scan, memread_apply_ss_kernel) or 1280 * entries * HW where it recomputes them',
                'timing': 'HIP events around every launch of frames that take no part in the side-stream overlap (kernel alone '
                          f'on the device: {n_full} frame(s) of the timed region); the memory read\'s apply launch additionally on every '
                          f'{apply_every or "-"}th timed frame behind an idle side stream ({len(apply_records)} launches in all); rocprofv3 counterpart: profiles/{PROFILE_ROUND}_kernel_stats_no_overlap.csv (--no-overlap run); '
                          f'profiles/{PROFILE_ROUND}_kernel_stats.csv is the default command, where overlapped launches run longer',
                'kernels': [{'kernel': c_[0], 'achieved': round(c_[1] / (c_[2] * 1e-3) / 1e12, 2),
                             'frac': round(c_[1] / (c_[2] * 1e-3) / 1e12 / peak, 4), 'avg_launch_us': round(c_[2] * 1e3 / c_[3], 2),
                             'launches_timed': c_[3], 'share_of_instrumented_time': round(c_[2] / c_[4] / all_ms, 4),
                             'traffic': traffic_of(c_[0])}
                            for c_ in sorted(cands, key=lambda c_: -c_[2] / c_[4])],
                'all_conv_achieved': round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                'all_conv_frac': round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4)}

    memread = None
    if mem_records:
        ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in mem_records)
        alg = sum(1280.0 * b * hw for b, hw, _, _ in mem_records)          # 2*(128+512) FLOP per (entry, query): scores once
        # as executed: the scores are formed once (statistics scan) when the apply kernel reads them back, in both passes otherwise
        twice = not (apply_records and all(r_[4] for r_ in apply_records))
        done = sum((1536.0 if twice else 1280.0) * b * hw for b, hw, _, _ in mem_records)
        memread = {'kernels': 'bank_scan_kernel<0> + memread_apply kernel + finish', 'frames_timed': len(mem_records),
                   'ms_per_frame': round(ms / len(mem_records), 3),
                   'achieved_algorithmic': round(alg / (ms * 1e-3) / 1e12, 2), 'achieved_executed': round(done / (ms * 1e-3) / 1e12, 2),
                   'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac_algorithmic': round(alg / (ms * 1e-3) / 1e12 / peak, 4),
                   'mean_bank_entries_per_object': round(sum(b for b, _, _, _ in mem_records) / (2.0 * len(mem_records)), 1)}

    # ---- whole-frame roofline: F_min(B) = 538.48 GFLOP + 3072*B*HW
    timed_sizes = bank_sizes[s_first - 1:s_first - 1 + K]
    b_mean = sum(sum(x) for x in timed_sizes) / (2.0 * K)
    b_mean_clip = sum(sum(x) for x in bank_sizes[:n_iter]) / (2.0 * n_iter)
    fps = world * K / elapsed
    full_clip_fps = world * last_iter / clip_elapsed
    f_min = 538.48e9 + 3072.0 * b_mean * 1620
    frame_frac = (fps / world) * f_min / (peak * 1e12)
    f_ref = 666.56e9 + 3072.0 * b_mean * 1620          # op-for-op reference FLOPs (per-object duplicate convs counted)
    frame_frac_ref = (fps / world) * f_ref / (peak * 1e12)
    f_min_clip = 538.48e9 + 3072.0 * b_mean_clip * 1620
    # what the matrix pipe EXECUTES per frame: the plan's launch lists carry every launch's own FLOP (a Winograd-domain GEMM
    # 2 * 36 * tiles * Cin * Cout = 1/4 of the direct convolution it replaces); one memorize + one decoder pass + half a two-frame
    # query pass per frame, + the memory read / update contractions
    qs0 = plan.qsets[0]
    lsum = lambda lst: sum(l.flops for l in lst)
    conv_exec = lsum(plan.mem) + lsum(qs0.post[0]) + lsum(qs0.pre[2]) / 2.0
    wino_names = [l.name.split('.wino_gemm')[0] for lst in (plan.mem, qs0.post[0], qs0.pre[2]) for l in lst if '.wino_gemm' in l.name]
    f_exec = conv_exec + 3072.0 * b_mean * 1620
    frame_frac_exec = (fps / world) * f_exec / (peak * 1e12)
    tms = sorted(ms_ for t_, ms_ in frame_ms if s_first <= t_ < s_first + K)
    frame_stats = {'p50': round(tms[len(tms) // 2], 3), 'p90': round(tms[min(len(tms) - 1, int(0.9 * len(tms)))], 3),
                   'max': round(tms[-1], 3), 'min': round(tms[0], 3),
                   'note': 'host wall between the completions of consecutive steps (the loop keeps one step in flight); sampled frames (events around every launch, no overlap) are the slow tail'}

    # ---- CPU baseline + parity on the first frames of the same clip
    cpu = None
    parity = None
    if not args.no_cpu_baseline and world == 1:
        from oracle import afb_urr_ref as O
        nthr = min(16, os.cpu_count() or 1)     # fastest setting measured on the GPU box's 256-core host (8/16/32/64/128 tried)
        torch.set_num_threads(nthr)
        n_cpu = min(args.cpu_frames, last_iter)
        fr_cpu = frames[:n_cpu + 1].cpu()
        kw = dict(budget=args.budget, size=net_size, mem_every=mem_every)
        O.run_clip(sd, fr_cpu[:2], m0, **kw)                               # warm the CPU kernels
        c0 = time.perf_counter()
        ref = O.run_clip(sd, fr_cpu, m0, **kw)
        c1 = time.perf_counter()
        cpu = {'value': round(n_cpu / (c1 - c0), 4), 'unit': 'frames/s', 'cores': nthr, 'kind': 'port',
               'sample': f'first {n_cpu} frames of the same {H0}x{W0} clip (incl. first-frame memorize), torch CPU oracle '
                         f'(oracle/afb_urr_ref.py, pinned against the reference: tests/test_oracle_golden.py)'}
        lab = labels_raw[:n_cpu + 1].cpu()
        ious = [miou(lab[t], ref['labels'][t]) for t in range(1, n_cpu + 1)]
        parity = {'frames': n_cpu, 'miou_vs_oracle': round(min(ious), 5), 'miou_vs_oracle_mean': round(sum(ious) / len(ious), 5),
                  'bank_sizes_equal': bank_sizes[:n_cpu] == ref['bank_sizes'],
                  'oracle_dtype': 'f32 (the reduced-precision modes are compared with the f32 oracle labels)'}

    if world == 1 and golden is not None and seed == int(golden['seed']) and last_iter >= n_iter and not args.checkpoint:
        import numpy as np
        refl = torch.from_numpy(np.unpackbits(golden['labels'], axis=-1)[..., :W0])
        lab = labels_raw[:n_frames].cpu()
        ious = [miou(lab[t], refl[t]) for t in range(1, n_frames)]
        parity = dict(parity or {})
        gs = golden['bank_sizes']
        parity.update({'full_clip_frames': n_iter, 'full_clip_miou_min': round(min(ious), 5),
                       'full_clip_miou_mean': round(sum(ious) / len(ious), 5),
                       'full_clip_bank_max_abs_diff': int(max(abs(int(a) - int(b)) for x, y in zip(bank_sizes[:n_iter], gs) for a, b in zip(x, y))),
                       'full_clip_reference': 'tests/golden/c2_480x854_100.npz (reference model + FeatureBank on CPU, '
                                              'oracle/gen_c2_golden.py)'})

    timed_desc = (f'all {n_iter} loop iterations timed' if (s_first == 1 and K == n_iter) else
                  f'{K} iterations timed' + (f' (frames {s_first}-{s_first + K - 1}; frames 1-{s_first - 1} are an untimed pre-roll that '
                                             f'builds the bank, the clip is finished untimed)' if not stream_mode and K < n_iter else ''))
    out = {'metric': 'segmented frames/sec at 480p' if args.workload == 'C2' else f'segmented frames/sec at {H0}p', 'value': round(fps, 3), 'unit': 'frames/s', 'n_gpus': world,
           'steps': K, 'warmup': Wm, 'ms_per_step': round(1e3 * elapsed / K, 3), 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPES[args.precision],
           'data': ('synthetic' if args.clip == 'easy' else 'synthetic (hard clip: tools/synth.clip_hard, water and land differ by texture only)') if not args.checkpoint else
                   f'synthetic frames ({args.clip} clip); weights from {os.path.basename(args.checkpoint)} (trained on synthetic clips)',
           'config': {'workload': f'{args.workload}: {n_frames}-frame {H0}x{W0} synthetic clip per GPU through the test_video_seg.py loop '
                                  f'(' + ('bicubic resize to 480p+' if (Hn, Wn) != (H0, W0) else '') +
                                  f'segment+softmax+memorize' + (f' every {mem_every}th frame' if mem_every > 1 else '') +
                                  f'+bank update+argmax+CCL), {args.precision}, budget {args.budget}; {timed_desc}' +
                                  (f'; the frames between two memorize calls as one batched pass (--group: groups of {group_n})' if group_n else ''),
                      'grouped_frames': group_n or None,
                      'timed_frames': [s_first, s_first + K - 1], 'preroll_frames': s_first - 1, 'warm_steps_run': warm_steps,
                      'mean_bank_entries_per_object': round(b_mean, 1),
                      'full_clip_mean_bank_entries_per_object': round(b_mean_clip, 1),
                      'network_resolution': f'{Hn}x{Wn} ' + ('(native)' if args.native and (Hn, Wn) == (H0, W0) else '(reference semantics: 480-pixel short edge)'),
                      'frame_mfma_frac_Fmin': round(frame_frac, 4) if mem_every == 1 else None,
                      'frame_mfma_frac_executed': round(frame_frac_exec, 4) if mem_every == 1 else None,
                      'conv_gflop_per_frame': {'algorithmic_Fmin': 538.48, 'executed': round(conv_exec / 1e9, 2),
                                               'note': 'Winograd F(4x4,3x3) layers execute 1/4 of their direct-convolution FLOP (csrc/conv_winograd.hip; '
                                                       'frame_mfma_frac_Fmin prices the frame at the ALGORITHMIC FLOP of SURVEY.md 8(d), '
                                                       'frame_mfma_frac_executed at what the matrix pipe runs)'},
                      'winograd_layers': sorted(set(wino_names)),
                      # the look-ahead stream is picked so that it shares no hardware queue with the frame loop's stream (streams on one
                      # queue run in order): [found, candidates tried] per probe of this process (vfloodnet_amd._lib.independent_stream)
                      'side_stream_probes': [[bool(a_), int(b_)] for a_, b_ in vlib.PROBES],
                      'frame_mfma_frac_Fref_reference_equivalent': round(frame_frac_ref, 4) if mem_every == 1 else None},
           'full_clip_fps': round(full_clip_fps, 3),
           # `value` is K frames (0.12 s at the driver's --steps 20): a 1-2 % effect cannot be read off it.  The same run's other clocks
           # on the same kernels: every iteration of the clip (pre-roll + window + rest; includes the instrumented frames, so it is a
           # lower bound) and the median host wall between consecutive step completions inside the window
           'value_ci': {'window_fps': round(fps, 3), 'full_clip_fps': round(full_clip_fps, 3), 'frames_full_clip': int(world * last_iter),
                        'p50_frame_fps': round(world * 1e3 / tms[len(tms) // 2], 3),
                        'low': round(min(fps, full_clip_fps), 3), 'high': round(max(fps, full_clip_fps, world * 1e3 / tms[len(tms) // 2]), 3),
                        'graph_captures_in_window': int(captures_in_window),
                        'note': 'quote A/Bs below 2 % from full_clip_fps (99 steps) of alternating runs on one box, not from `value`'},
           'full_clip_frame_mfma_frac_Fmin': round((full_clip_fps / world) * f_min_clip / (peak * 1e12), 4) if mem_every == 1 else None,
           'frame_ms': frame_stats, 'sustained': sustained,
           'roofline': roof, 'memory_read': memread, 'cpu_baseline': cpu, 'parity': parity}
    # ---- what a multi-GPU record needs to show that N ranks met over RCCL, and where the time went
    rates = [K / r_[0] for r_ in per_rank]
    out['distributed'] = dict(dist_info, **{
        'collective': 'one all_gather_into_tensor of the uint8 label blocks [K, H0, W0] per rank inside the timed bracket' if dist_on else None,
        'all_gather_ms_max': round(1e3 * max(r_[1] for r_ in per_rank), 3) if dist_on else None,
        'all_gather_ms_per_rank': [round(1e3 * r_[1], 3) for r_ in per_rank] if dist_on else None,
        'all_gather_bytes_per_rank': int(K * H0 * W0) if dist_on else None,
        'bank_sizes_gathered': {'clips': len(sizes_all), 'shape_per_clip': [int(v) for v in sizes_all[0].shape], 'dtype': 'int32',
                                'own_block_intact': bool(torch.equal(sizes_all[0], torch.tensor(bank_sizes[:last_iter], dtype=torch.int32))),
                                'note': 'int32 [T, obj_n] live bank entries per frame of every rank\'s clip, gathered beside the masks outside the timed bracket (dist.gather_bank_sizes)'},
        'frames_per_s_per_rank': [round(x, 3) for x in rates],
        'frames_per_s_per_rank_min': round(min(rates), 3), 'frames_per_s_per_rank_max': round(max(rates), 3),
        'per_rank_note': 'K timed frames / (time until this rank\'s last frame has left the GPU); `value` = world * K / (max over ranks of the '
                         'whole bracket incl. the all-gather and both barriers)'})
    # ---- memory: peak HBM of the run ("peak HBM bytes")
    fb_ = runner.fb
    bank_bytes = 0
    for nm_ in ('_kbuf', '_vbuf', '_ibuf'):
        t_ = getattr(fb_, nm_, None)
        if t_ is not None:
            bank_bytes += t_.numel() * t_.element_size()
    out['hbm'] = {'peak_bytes_allocated': int(max(r_[2] for r_ in per_rank)), 'peak_bytes_reserved': int(torch.cuda.max_memory_reserved(dev)),
                  'resident_frames_bytes': int(frames.numel() * frames.element_size()),
                  'bank_slab_bytes_f32': int(bank_bytes), 'bank_capacity_entries_per_object': int(getattr(fb_, '_cap', 0)),
                  'final_bank_entries_per_object': [int(x) for x in bank_sizes[-1]] if bank_sizes else None,
                  'capacity': '288 GB HBM3E per MI355X'}
    # ---- frames/s against the bank size: blocks of the timed frames
    blk = 100 if K >= 400 else (10 if K >= 40 else 0)
    if blk:
        curve = []
        by_t = {t_: ms_ for t_, ms_ in frame_ms}
        for a in range(s_first, s_first + K - blk + 1, blk):
            ms_blk = [by_t[t_] for t_ in range(a, a + blk) if t_ in by_t]
            sz = bank_sizes[a - 1:a - 1 + blk]
            if len(ms_blk) == blk and sz:
                curve.append({'frames': [a, a + blk - 1], 'mean_bank_entries_per_object': round(sum(sum(x) for x in sz) / (2.0 * len(sz)), 1),
                              'frames_per_s': round(1e3 * blk / sum(ms_blk), 2)})
        out['bank_curve'] = {'block_frames': blk, 'points': curve,
                             'note': 'host wall between step completions, this rank; sampled frames (events around every launch) included'}
    if args.main_loop and world == 1:
        # the loop a user of test_video_seg.py runs, files to files, beside the HBM-resident number: its own
        # process (DataLoader workers, writer threads, its own model), started as a child -- this one keeps its GPU state
        import re
        import subprocess
        try:
            cp = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'main_throughput.py'), str(int(args.main_loop))],
                                capture_output=True, text=True, timeout=600)
            m_ = re.search(r'main\(\) frame loop: (\d+) frames, viz=(\w+): ([0-9.]+) frames/s', cp.stdout)
            out['extra'] = {'main_files_to_files_fps': float(m_.group(3)) if m_ else None, 'frames': int(m_.group(1)) if m_ else None,
                            'overlays': (m_.group(2) == 'True') if m_ else None,
                            'what': 'vfloodnet_amd.video_seg.main: 480p JPEG files on disk -> mask + overlay PNG files on disk, frame loop after 4 '
                                    'start-up iterations, second run of the process (scripts/main_throughput.py)',
                            'error': None if m_ else (cp.stderr or cp.stdout)[-300:]}
        except Exception as exc_:                     # (never fail the bench line over the side measurement)
            out['extra'] = {'main_files_to_files_fps': None, 'error': repr(exc_)[:300]}
    print(json.dumps(out), flush=True)
    if os.environ.get('VFN_BENCH_DUMP'):            # per-step host times of the whole run (diagnostics)
        with open(os.environ['VFN_BENCH_DUMP'], 'w') as f:
            json.dump({'frame_ms': frame_ms, 'bank_sizes': bank_sizes, 'timed': [s_first, s_first + K - 1]}, f)
    if dist_on:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main() or 0)
