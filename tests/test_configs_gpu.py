"""BASELINE.json configs C3 and C5 *in the dtype they name* (bf16), at their shapes, against the f32 CPU oracle.

The reference has no reduced-precision path; this package offers two: ``bf16x3`` (every matrix operand
split into two bf16, 16 significant bits, three MFMAs per product) and plain ``bf16`` (8 bits).  Both are run here on
the configs' workloads and the mIoU they actually reach against the oracle's fp32 labels is asserted: bf16x3 is held to
the fidelity bar (>= 0.99 per frame, identical bank sizes); plain bf16 is asserted at the level it honestly reaches with
the margin-free synthetic weights (it is NOT a parity-green configuration, and the numbers say so).

Round 4, the per-layer sensitivity sweep (scripts/precision_sweep.py, profiles/r04_precision_sweep_C3.json: every group of layers
alone in plain bf16 on the 100-frame C3 clip, the rest bf16x3, labels against the f32 run): only the local refinement head
keeps min mIoU >= 0.99 (0.995); decoder.ResMM reaches 0.988, every other group -- either encoder, KeyValue, the memory read,
the bank update's cosine match, convFM, RF3, RF2, pred2 -- falls to 0.84-0.97 on its own.  With these weights there is no
cheaper mixed assignment: bf16x3 is the reduced-precision answer for C3 / C5, and ``AFB_URR.precision_map`` (below) is the
mechanism a checkpoint with real margins can use to move groups to plain bf16."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 20200212


def miou(a, b):
    v = []
    for c in (0, 1):
        inter = ((a == c) & (b == c)).sum().item()
        union = ((a == c) | (b == c)).sum().item()
        v.append(1.0 if union == 0 else inter / union)
    return sum(v) / 2


@pytest.fixture(scope='module')
def sd():
    from tools import synth
    return synth.make_state_dict(SEED)


@pytest.fixture(scope='module')
def c3_oracle(sd):
    """C3 workload: 1280x720 clip, resized to 480p by the loop, bank updated every 5th frame; oracle in f32."""
    from tools import synth
    from oracle import afb_urr_ref as O
    frames, m0 = synth.clip(5, 12, 720, 1280)
    torch.set_num_threads(16)
    ref = O.run_clip(sd, frames, m0, size=480, mem_every=5)
    return frames, m0, ref


@pytest.mark.parametrize('group', [False, True])
@pytest.mark.parametrize('precision', ['bf16x3', 'bf16'])
def test_c3_720p_every_5th_reduced_precision(gpu, sd, c3_oracle, precision, group):
    """``group``: the frames between two key frames as one batched pass (ClipRunner.launch_group, round 6) -- held to the oracle
    exactly as the frame-by-frame loop is."""
    from vfloodnet_amd import AFB_URR
    from vfloodnet_amd.video_seg import run_clip
    frames, m0, ref = c3_oracle
    model = AFB_URR(gpu, update_bank=True, precision=precision).to(gpu).eval()
    model.load_state_dict(sd, strict=True)
    out = run_clip(model, frames.to(gpu), m0, size=480, mem_every=5, group=group)
    T = frames.shape[0]
    assert out['labels'].shape == (T, 720, 1280)
    ious = [miou(out['labels'][t], ref['labels'][t]) for t in range(1, T)]
    print(f'C3 {precision}: mIoU min {min(ious):.5f} mean {sum(ious) / len(ious):.5f}; bank {out["bank_sizes"][-1]} vs {ref["bank_sizes"][-1]}')
    # the bank only changes on frames 5 and 10
    assert out['bank_sizes'][3] == out['bank_sizes'][0] and out['bank_sizes'][4] != out['bank_sizes'][3]
    if precision == 'bf16x3':
        assert out['bank_sizes'] == ref['bank_sizes']
        assert min(ious) >= 0.99, ious
    else:
        # plain bf16 operands: 2^-9 relative noise per product against logits without margin (synthetic weights):
        # several per cent of the pixels flip.  Asserted at what it reaches, not at the parity bar.
        # Measured: 0.94 on the first frame, 0.63-0.75 once its own masks have been memorised (frames 5, 10).
        # The first frame (the bank still holds the given mask only) is asserted tightly so that a regression of the bf16
        # kernels is visible; the later floor is what the closed loop on margin-free weights reaches, not a parity claim.
        print('C3 bf16 per-frame mIoU: ' + ' '.join(f'{x:.4f}' for x in ious))
        assert ious[0] >= 0.92 and min(ious) >= 0.55, ious
        drift = max(abs(a - b) for x, y in zip(out['bank_sizes'], ref['bank_sizes']) for a, b in zip(x, y))
        assert drift <= 0.05 * max(ref['bank_sizes'][-1])


def test_precision_map_moves_one_group_to_bf16(gpu, sd, c3_oracle):
    """``model.precision_map = {'decoder.local': 'bf16'}`` on a bf16x3 model: the local head's 64-channel convolutions launch the
    plain-bf16 kernel, everything else stays bf16x3, and the C3 clip still meets the fidelity bar (the one group the sweep found
    harmless)."""
    from vfloodnet_amd import AFB_URR, ops
    from vfloodnet_amd.video_seg import run_clip
    frames, m0, ref = c3_oracle
    model = AFB_URR(gpu, update_bank=True, precision='bf16x3').to(gpu).eval()
    model.precision_map = {'decoder.local': 'bf16'}
    model.load_state_dict(sd, strict=True)
    out = run_clip(model, frames.to(gpu), m0, size=480, mem_every=5)
    ious = [miou(out['labels'][t], ref['labels'][t]) for t in range(1, frames.shape[0])]
    assert min(ious) >= 0.99 and out['bank_sizes'] == ref['bank_sizes'], ious
    eng = model.engine()
    plan = eng.plan(480, 853, 2)
    modes = {}
    for lst in plan.all_lists():
        for l in lst:
            if l.fn is ops.conv2d_launch:
                modes.setdefault(l.name.split('[')[0], set()).add(int(l.args[2]))
    assert modes['decoder.local_convFM.local'] == {1} and modes['decoder.local_convFM.r1'] == {1}
    assert modes['decoder.local_ResMM.conv1'] == {2}                      # 32 channels: no 64-channel K tile, stays bf16x3
    assert all(m_ == {2} for n_, m_ in modes.items() if not n_.startswith('decoder.local'))
    with pytest.raises(ValueError):
        bad = AFB_URR(gpu, update_bank=True, precision='bf16x3').to(gpu).eval()
        bad.precision_map = {'encoder_q': 'fp64'}
        bad.load_state_dict(sd, strict=True)
        bad.engine()


def test_c5_shape_1080p_long_stream_bf16x3(gpu, sd):
    """C5's shape: a 1920x1080 stream at reference semantics (resize to 480p), every frame memorised, bf16x3.
    180 frames with a budget whose per-object share (200,000 entries) is reached around frame 157 (a good part of the new
    features merge into existing entries instead of being appended), so the run covers
    the growing bank, B >= 200 k and LFU eviction.  The first frames are compared with the f32 oracle; beyond them the
    domain's invariants are checked (FeatureBank.py:102-103,117-143; myutils/data.py:17-37)."""
    from tools import synth
    from vfloodnet_amd import AFB_URR, ops
    from vfloodnet_amd.video_seg import ClipRunner
    from oracle import afb_urr_ref as O
    T, H, W, n_ref = 180, 1080, 1920, 6
    budget = 500000                                   # class_budget = 0.8 * 250000 = 200000.0
    frames, m0 = synth.clip_on_device(7, T, H, W, gpu)
    torch.set_num_threads(16)
    ref = O.run_clip(sd, frames[:n_ref + 1].cpu(), m0, size=480, budget=budget)
    model = AFB_URR(gpu, update_bank=True, precision='bf16x3').to(gpu).eval()
    model.load_state_dict(sd, strict=True)
    runner = ClipRunner(model, 2, budget, size=480, postprocess=True)
    onehot = synth.onehot(m0).unsqueeze(0).to(gpu)
    runner.start(frames[0:1], onehot)
    sizes, ious = [], []
    for t in range(1, T):
        lab = runner.step(frames[t:t + 1], next_frame=frames[t + 1:t + 2] if t + 1 < T else None)
        sizes.append(runner.bank_sizes())
        if t <= n_ref:
            ious.append(miou(runner._label_dev.cpu(), ref['labels'][t]))           # before post-processing, as the oracle's
        if t in (n_ref, T // 2, T - 1):
            post = torch.from_numpy(lab.numpy().copy())
            assert set(post.unique().tolist()) <= {0, 1}
            again = ops.postprocess_pred_device(post.to(gpu)).cpu()
            assert torch.equal(again, post)                                      # largest-blob filter is idempotent
    print(f'C5 bf16x3: mIoU(first {n_ref}) min {min(ious):.5f}; bank {sizes[0]} -> {sizes[-1]}, replace_n {runner.fb.replace_n}')
    # merge / append decisions sit on a float threshold (cosine > 0.95): 2^-16 operand noise may move single entries
    drift = max(abs(a - b) for x, y in zip(sizes[:n_ref], ref['bank_sizes']) for a, b in zip(x, y))
    assert drift <= 3, (sizes[:n_ref], ref['bank_sizes'])
    assert min(ious) >= 0.99, ious
    cb = runner.fb.class_budget
    assert cb == 200000.0
    peak = max(max(s) for s in sizes)
    assert peak >= 195000 and all(max(s) <= cb for s in sizes)                   # B + n_append <= class_budget after remove()
    first_evict = next(i for i in range(1, len(sizes)) if sizes[i][0] < sizes[i - 1][0] or sizes[i][1] < sizes[i - 1][1])
    assert all(sizes[i][c] >= sizes[i - 1][c] for i in range(1, first_evict) for c in (0, 1))   # monotone until the budget
    assert first_evict > 120
    assert runner.fb.replace_n.sum() > 0 and np.all(runner.fb.peak_n <= cb)


def test_c5_shape_plain_bf16_is_measured_not_parity(gpu, sd):
    """C5's shape in the dtype BASELINE.json names literally (plain bf16 operands): 40 frames of the 1080p stream.  Like C3 it is
    asserted at what it reaches with the margin-free synthetic weights -- tight on the first frame (a regression of the bf16 kernels
    shows there), a floor afterwards -- next to the domain's invariants; the parity configuration for C5 is bf16x3 (above, and
    tests/test_round3_gpu.py at the full 2000 frames).  The sweep behind that statement: profiles/r04_precision_sweep_C3.json."""
    from tools import synth
    from vfloodnet_amd import AFB_URR
    from vfloodnet_amd.video_seg import ClipRunner
    from oracle import afb_urr_ref as O
    T, H, W, n_ref = 40, 1080, 1920, 5
    budget = 2 * int(1.25 * 2 * (T + 2) * 1620) + 4
    frames, m0 = synth.clip_on_device(7, T, H, W, gpu)
    torch.set_num_threads(16)
    ref = O.run_clip(sd, frames[:n_ref + 1].cpu(), m0, size=480, budget=budget)
    model = AFB_URR(gpu, update_bank=True, precision='bf16').to(gpu).eval()
    model.load_state_dict(sd, strict=True)
    runner = ClipRunner(model, 2, budget, size=480, postprocess=True)
    runner.start(frames[0:1], synth.onehot(m0).unsqueeze(0).to(gpu))
    sizes, ious = [], []
    for t in range(1, T):
        lab = runner.step(frames[t:t + 1], next_frame=frames[t + 1:t + 2] if t + 1 < T else None)
        sizes.append(runner.bank_sizes())
        if t <= n_ref:
            ious.append(miou(runner._label_dev.cpu(), ref['labels'][t]))
        assert set(torch.from_numpy(lab.numpy().copy()).unique().tolist()) <= {0, 1}
    print('C5 plain bf16 per-frame mIoU vs the f32 oracle: ' + ' '.join(f'{x:.4f}' for x in ious) + f'; bank {sizes[0]} -> {sizes[-1]}')
    # (round 5: the bf16 mode's 3x3 layers run as Winograd with bf16 V / U -- an order of magnitude more operand noise through the inverse
    # transform than the direct bf16 kernel, invisible on trained weights (test_plain_bf16_meets_the_bar_on_trained_weights) and worth
    # 2-3 points of first-frame agreement on these margin-free ones: 0.945 -> 0.918)
    assert ious[0] >= 0.90 and min(ious) >= 0.5, ious
    assert all(sizes[i][c] >= sizes[i - 1][c] for i in range(1, len(sizes)) for c in (0, 1))     # nothing is evicted at this budget
    assert float(runner.fb.replace_n.sum()) == 0.0
    drift = max(abs(a - b) for x, y in zip(sizes[:n_ref], ref['bank_sizes']) for a, b in zip(x, y))
    assert drift <= 0.05 * max(ref['bank_sizes'][-1])


# ------------------------------------------------------------------------------------------------ round 5: weights with margins
@pytest.fixture(scope='module')
def trained_sd(gpu):
    """The synthetic checkpoint after 600 steps of the HIP training step on synthetic clips (tools/train_synth.py, ~20 s): the loss
    falls from 1.0 to ~0.12 and the logits get margins (median |logit_1 - logit_0| ~ 30 instead of 0.1-0.3)."""
    from tools.train_synth import train_checkpoint
    sd_t, info = train_checkpoint(gpu, steps=600, lr=2e-5)
    print('trained checkpoint:', info)
    assert info['restores_after_collapse'] == 0 and info['loss_last_50'] < 0.25 < info['loss_first_50'], info
    return sd_t


@pytest.mark.parametrize('workload', ['C3', 'C5'])
def test_plain_bf16_meets_the_bar_on_trained_weights(gpu, trained_sd, workload):
    """BASELINE.json names *bf16* for C3 (720p, every 5th frame memorised) and C5 (1080p stream).  With the margin-free random
    weights plain bf16 settles at mIoU 0.55-0.8 (asserted above at what it reaches); with weights that have been TRAINED -- the
    situation the reference's own checkpoint is in -- the same kernels meet the fidelity bar: label mIoU >= 0.99 on every frame
    against the f32 CPU oracle run on the same trained weights (which also re-pins the f32 HIP path on them), and against the f32
    HIP run.  The full-length measurement (3000 steps; C3 100 frames, C5 120 frames, C2 100 frames; logit-margin percentiles):
    scripts/bf16_trained_margins.py -> profiles/r05_bf16_trained_margins.json (bf16 min mIoU 0.9997-0.9998, bf16x3 1.0)."""
    from tools import synth
    from vfloodnet_amd import AFB_URR
    from vfloodnet_amd.video_seg import run_clip
    from oracle import afb_urr_ref as O
    if workload == 'C3':
        T, H, W, mem_every, seed = 12, 720, 1280, 5, 5
    else:
        T, H, W, mem_every, seed = 8, 1080, 1920, 1, 7
    frames, m0 = synth.clip(seed, T, H, W)
    torch.set_num_threads(16)
    ref = O.run_clip(trained_sd, frames, m0, size=480, mem_every=mem_every)
    torch.set_num_threads(1)
    labs = {}
    for precision in ('fp32', 'bf16x3', 'bf16') + (('bf16 grouped',) if mem_every > 1 else ()):
        model = AFB_URR(gpu, update_bank=True, precision=precision.split()[0]).to(gpu).eval()
        model.load_state_dict(trained_sd, strict=True)
        # ('bf16 grouped': C3's frames between two key frames as one batched pass, ClipRunner.launch_group -- the same bar)
        out = run_clip(model, frames.to(gpu), m0, size=480, mem_every=mem_every, group=precision.endswith('grouped'))
        labs[precision] = out['labels']
        ious = [miou(out['labels'][t], ref['labels'][t]) for t in range(1, T)]
        drift = max(abs(a - b) for x, y in zip(out['bank_sizes'], ref['bank_sizes']) for a, b in zip(x, y))
        print(f'{workload} {precision} on trained weights: mIoU vs the f32 oracle min {min(ious):.5f} mean {sum(ious) / len(ious):.5f}; '
              f'bank {out["bank_sizes"][-1]} vs {ref["bank_sizes"][-1]} (drift {drift})')
        assert min(ious) >= 0.99, (precision, ious)
        assert drift <= (2 if not precision.startswith('bf16 ') and precision != 'bf16' else 0.01 * max(ref['bank_sizes'][-1]) + 2), (precision, drift)
    for precision in [k for k in labs if k != 'fp32']:
        ious = [miou(labs[precision][t], labs['fp32'][t]) for t in range(1, T)]
        assert min(ious) >= 0.995, (precision, ious)
    # the ground truth of the synthetic clip (the mask rolled with the frame): the trained network actually segments it
    gt = [torch.roll(m0, (2 * t, 5 * t), (0, 1)) for t in range(T)]
    acc = [miou(labs['fp32'][t], gt[t]) for t in range(1, T)]
    assert min(acc) >= 0.9, acc


# ------------------------------------------------------------------------------------------------ round 6: a task whose margins do not saturate
@pytest.fixture(scope='module')
def hard_sd(gpu):
    """The synthetic checkpoint after 1200 steps on the HARD task (tools/train_synth.py task='hard', ~35 s): water and land share
    their colour statistics and differ by texture only, every training label is displaced and flipped along the shoreline
    (~10 % disagree with the image).  The loss plateaus near 0.5 instead of 0.1 and the logits keep moderate margins (median
    |logit_1 - logit_0| 4-7, nothing at the clamp -- the tinted task of ``trained_sd`` has its median AT the clamp)."""
    from tools.train_synth import train_checkpoint
    sd_h, info = train_checkpoint(gpu, steps=1200, lr=2e-5, task='hard')
    print('hard-task checkpoint:', info)
    assert info['restores_after_collapse'] == 0 and 0.3 < info['loss_last_50'] < 0.7 < info['loss_first_50'], info
    return sd_h


def test_plain_bf16_on_a_task_with_unsaturated_margins(gpu, hard_sd):
    """the bf16 claim on weights whose margins do not sit at the clamp.  C3 shape (720p, every 5th frame
    memorised) on a hard clip: the f32 HIP run against the f32 CPU oracle (parity anchor on these weights), bf16x3 and plain bf16
    against the f32 HIP run -- at what they honestly reach -- and WHERE plain bf16's flips are: only where the f32 margin is below
    0.5.  The full-length measurement (3000 steps; C3 / C2 / C5 shapes, agreement binned by the f32 margin):
    scripts/bf16_margins_hard.py -> profiles/r06_bf16_margins_hard.json (bf16 min mIoU 0.9985-0.9989 on C3 / C2, every flip
    in the |margin| < 0.25 bin; bf16x3 1.0 with equal banks)."""
    from tools import synth
    from vfloodnet_amd import AFB_URR
    from vfloodnet_amd.video_seg import ClipRunner
    from oracle import afb_urr_ref as O
    T, H, W, mem_every, seed = 14, 720, 1280, 5, 3
    frames, m0 = synth.clip_hard(seed, T, H, W)
    torch.set_num_threads(16)
    ref = O.run_clip(hard_sd, frames[:6], m0, size=480, mem_every=mem_every)
    torch.set_num_threads(1)
    frames = frames.to(gpu)
    m = (m0 > 0).to(torch.uint8)
    onehot = torch.stack([1 - m, m], 0).unsqueeze(0).to(gpu)
    runs = {}
    for precision in ('fp32', 'bf16x3', 'bf16'):
        model = AFB_URR(gpu, update_bank=True, precision=precision).to(gpu).eval()
        model.load_state_dict(hard_sd, strict=True)
        runner = ClipRunner(model, 2, 250000, size=480, mem_every=mem_every)
        runner.start(frames[0:1], onehot)
        h, w = runner._net_frame(frames[0:1]).shape[-2:]
        plan = model.engine().plan(h, w, 2)
        labels, net, marg = [m.clone()], [], []
        for t in range(1, T):
            lab = runner.step(frames[t:t + 1])
            labels.append(torch.from_numpy(lab.numpy().copy()))
            sc = plan.score[0]
            net.append((sc[1] > sc[0]).cpu())
            marg.append((sc[1] - sc[0]).abs().cpu())
        runs[precision] = (labels, torch.stack(net), torch.stack(marg), runner.bank_sizes())
    ious = [miou(runs['fp32'][0][t], ref['labels'][t]) for t in range(1, 6)]
    assert min(ious) >= 0.999, ious                                       # the f32 HIP path on these weights == the CPU oracle
    mg = runs['fp32'][2].flatten()
    med, p5, clamp = float(mg.median()), float(mg.kthvalue(int(0.05 * mg.numel())).values), float((mg >= 31.0).float().mean())
    print(f'hard task: f32 |margin| median {med:.2f}, p5 {p5:.2f}, at the clamp {clamp:.4f}')
    assert med < 12.0 and p5 < 2.5 and clamp < 0.01, (med, p5, clamp)      # the margins really are unsaturated
    dy, dx = synth.hard_step(H, W)
    gt = [torch.roll(m0, (dy * t, dx * t), (0, 1)) for t in range(T)]
    acc = [miou(runs['fp32'][0][t], gt[t]) for t in range(1, T)]
    assert min(acc) >= 0.75, acc                                           # ... and the network segments the task (not a coin flip)
    for precision, bar in (('bf16x3', 0.999), ('bf16', 0.99)):
        ious = [miou(runs[precision][0][t], runs['fp32'][0][t]) for t in range(1, T)]
        agree = runs[precision][1] == runs['fp32'][1]
        far = runs['fp32'][2] >= 0.5
        print(f'hard task {precision}: mIoU vs the f32 HIP run min {min(ious):.5f} mean {sum(ious) / len(ious):.5f}; pixel agreement '
              f'{float(agree.float().mean()):.6f}, where the f32 margin >= 0.5: {float(agree[far].float().mean()):.6f}; bank {runs[precision][3]} vs {runs["fp32"][3]}')
        assert min(ious) >= bar, (precision, ious)
        assert float(agree[far].float().mean()) >= 0.9999, precision       # flips live where the f32 run itself is undecided
