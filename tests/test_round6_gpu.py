"""Round 6: the software-pipelined plain-bf16 memory-read kernels against the kernels they replace -- same products in the same
order, so the results must be IDENTICAL bit for bit (read-out, hit counts / info bump, per-slice scan partials) -- over bank sizes
that exercise a single chunk, partial last chunks, two objects of different length and empty bank slices; the NaN-preserving
ReLU floor of the branch-free epilogues; the bank-size vector of a real loop."""
import math
import os
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _bank(gpu, B, hw):
    from vfloodnet_amd.feature_bank import FeatureBank
    fb = FeatureBank(2, 250000, gpu, precision='bf16')
    fb._hw = hw
    fb._alloc(hw, B)
    g = torch.Generator(device=gpu).manual_seed(B)
    fb._kbuf[:, :B].copy_(torch.randn(2, B, 128, device=gpu, generator=g))
    fb._vbuf[:, :B].copy_(torch.randn(2, B, 512, device=gpu, generator=g))
    fb._set_lengths([B, max(1, B - 37)])             # object 1 ends inside another chunk than object 0
    kvq = torch.randn(2, hw, 640, device=gpu, generator=g)
    kvq[..., :128] *= 2.0
    return fb, kvq


@pytest.mark.parametrize('B,hw', [(60, 60), (64, 150), (65, 150), (127, 60), (128, 1620), (1000, 150), (5000, 1620), (25037, 1620)])
def test_pipelined_bf16_apply_is_bit_identical(gpu, B, hw):
    """memread_apply_pipe_kernel (softmax of chunk c+1 in the shadow of chunk c's P^T V, value rows a chunk ahead, keys through
    registers, one barrier per chunk) == memread_apply_shw_kernel<false>, incl. the hit counts that feed fb.info."""
    from vfloodnet_amd.engine import Engine
    fb, kvq = _bank(gpu, B, hw)
    plan = types.SimpleNamespace(HW=hw, kv_q=kvq[0:1], ml=torch.empty(2, hw, 2, device=gpu), ml_part=torch.empty(2, 256, hw, 2, device=gpu),
                                 work=torch.zeros(4, dtype=torch.int32, device=gpu), o_part=torch.empty(2, 20, hw, 512, device=gpu),
                                 dec_in=torch.empty(2, hw, 512, device=gpu))
    out = {}
    info0 = fb._ibuf.clone()
    old = os.environ.get('VFN_APPLY_PIPE')
    try:
        for pipe in ('1', '0'):
            os.environ['VFN_APPLY_PIPE'] = pipe
            fb._ibuf.copy_(info0)
            Engine._memory_read(types.SimpleNamespace(mode=1), plan, fb, True)
            torch.cuda.synchronize()
            out[pipe] = (plan.dec_in.clone(), fb._ibuf.clone())
    finally:
        if old is None:
            os.environ.pop('VFN_APPLY_PIPE', None)
        else:
            os.environ['VFN_APPLY_PIPE'] = old
    assert torch.isfinite(out['1'][0]).all()
    assert torch.equal(out['1'][0], out['0'][0])
    assert torch.equal(out['1'][1], out['0'][1])
    assert int(fb._cnt.abs().sum()) == 0                      # the counters are at rest again
    assert not torch.equal(out['1'][1], info0) or B < 2       # (the read really bumped fb.info)


@pytest.mark.parametrize('mode', [0, 1])
@pytest.mark.parametrize('B,hw', [(60, 60), (65, 150), (127, 60), (1000, 150), (5000, 1620), (25037, 1620)])
def test_register_staged_bf16_scans_are_bit_identical(gpu, B, hw, mode):
    """bank_scan_pipe_kernel<mode> (keys through registers two chunks ahead, the update scan's row scales with them through LDS, three
    workgroups per CU) == the LDS-DMA image path of bank_scan_kernel<mode, 1>: every slice partial, bit for bit."""
    from vfloodnet_amd import _lib
    from vfloodnet_amd._lib import ptr, stream, check, BankScanDesc
    from vfloodnet_amd.feature_bank import pick_scan_slices, MAX_SPLIT_SCAN, DK
    fb, kvq = _bank(gpu, B, hw)
    rs = torch.rand(2, fb._cap, device=gpu) + 0.5
    klp, _ = fb.lp_image()
    work = torch.zeros(4, dtype=torch.int32, device=gpu)
    nsplit = pick_scan_slices(hw, 2, fb.len_upper())
    parts = {}
    old = os.environ.get('VFN_SCAN_PIPE')
    try:
        for pipe in ('1', '0'):
            os.environ['VFN_SCAN_PIPE'] = pipe
            part = torch.full((2, MAX_SPLIT_SCAN, hw, 2), float('nan'), device=gpu)
            d = BankScanDesc()
            d.q, d.bank_k, d.bank_len, d.part = ptr(kvq), ptr(fb._kbuf), ptr(fb._len_dev), ptr(part)
            d.rowscale = ptr(rs) if mode == 1 else None
            d.stride_q, d.stride_k, d.stride_rs = (hw * 640 if mode == 1 else 0), fb._cap * DK, (fb._cap if mode == 1 else 0)
            d.scale = 1.0 / math.sqrt(DK)
            d.ldq, d.q_per_obj, d.HW, d.obj_n, d.nsplit, d.mode = 640, (1 if mode == 1 else 0), hw, 2, nsplit, mode
            d.precision = 1
            d.work_counter = ptr(work)
            d.bank_k_lp = ptr(klp)
            check(_lib.lib().vfn_bank_scan(_lib.C.byref(d), stream()), 'vfn_bank_scan')
            torch.cuda.synchronize()
            parts[pipe] = part.flatten()[:2 * nsplit * hw * 2].clone()         # (the kernel's layout: [obj][nsplit][HW][2], densely packed)
    finally:
        if old is None:
            os.environ.pop('VFN_SCAN_PIPE', None)
        else:
            os.environ['VFN_SCAN_PIPE'] = old
    assert not torch.isnan(parts['1']).any()
    assert torch.equal(parts['1'].view(torch.int32), parts['0'].view(torch.int32))


def test_relu_floor_keeps_nan(gpu):
    """the branch-free epilogues applied fmaxf(v, relu ? 0 : -inf), which turns a NaN accumulator of a layer WITHOUT ReLU
    into -inf (and the next layer's ReLU into 0): a diverging run was silently kept alive.  The floor is now a NaN-preserving select:
    a NaN in the input of a convolution (with and without ReLU on its output, direct and Winograd form) leaves a non-finite value in
    the pixels whose window holds it -- and only there."""
    from vfloodnet_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 12, 16, 64, generator=g)
    x[0, 5, 7, 3] = float('nan')
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    for relu_out in (False, True):
        y = ops.conv2d_nhwc(x.to(gpu), ops.pad_rows(__import__('vfloodnet_amd').weights.pack_conv_weight(w)).to(gpu), 64, 3, 3, 1, 1, relu_out=relu_out)
        torch.cuda.synchronize()
        assert (~torch.isfinite(y[0, 5, 7])).any(), relu_out        # the pixel whose window holds the NaN
        assert torch.isfinite(y[0, 0, 0]).all()                     # ... and nothing spreads beyond the window
        yw = ops.conv2d_winograd(x.to(gpu), w.to(gpu), relu_out=relu_out)
        torch.cuda.synchronize()
        assert (~torch.isfinite(yw[0, 5, 7])).any(), relu_out
        assert torch.isfinite(yw[0, 0, 12]).all()


def test_loop_records_the_bank_size_vector(gpu):
    """an int32[T] bank-size vector travels beside the masks.  ClipRunner keeps it (one row per collected frame);
    dist.gather_bank_sizes carries it (tests/test_dist_gloo.py)."""
    from tools import synth
    from vfloodnet_amd import AFB_URR, dist as vdist
    from vfloodnet_amd.video_seg import ClipRunner
    model = AFB_URR(gpu, update_bank=True).to(gpu).eval()
    model.load_state_dict(synth.make_state_dict(20200212), strict=True)
    frames, m0 = synth.clip(2, 5, 96, 160)
    frames = frames.to(gpu)
    runner = ClipRunner(model, 2, 250000)
    runner.start(frames[0:1], synth.onehot(m0).unsqueeze(0).to(gpu))
    for t in range(1, 5):
        runner.step(frames[t:t + 1])
    assert len(runner.size_log) == 5 and runner.size_log[-1] == runner.bank_sizes()
    assert all(b >= a for x, y in zip(runner.size_log, runner.size_log[1:]) for a, b in zip(x, y))
    got = vdist.gather_bank_sizes([runner.size_log], 1, 0, 1, gpu)
    assert got[0].dtype == torch.int32 and got[0].tolist() == runner.size_log


def test_independent_stream_runs_beside_the_current_stream(gpu):
    """A HIP stream is not a hardware queue: streams that share one (GPU_MAX_HW_QUEUES, default 4; PyTorch creates 32 per priority)
    run in order.  ``_lib.independent_stream`` returns a stream measured NOT to share the current stream's queue: a tiny kernel
    on it finishes while the current stream still has milliseconds of work queued -- the property the look-ahead stream, the PNG
    sink and the decode stream of the loop need (round 6: the PNG sink on the frame loop's queue cost video_seg.main 0.85 ms of an
    idle device per frame)."""
    from vfloodnet_amd import _lib
    cur = torch.cuda.current_stream(gpu)
    s1 = _lib.independent_stream(gpu)
    s2 = _lib.independent_stream(gpu, beside=[s1])
    big = torch.zeros(32 * 1024 * 1024, device=gpu)
    small = torch.zeros(16, device=gpu)
    for cand, others in ((s1, [cur]), (s2, [cur, s1])):
        for ref in others:
            torch.cuda.synchronize()
            ref_end, cand_end = torch.cuda.Event(), torch.cuda.Event()
            with torch.cuda.stream(ref):
                for _ in range(48):
                    big.add_(1.0)
                ref_end.record()
            with torch.cuda.stream(cand):
                small.add_(1.0)
                cand_end.record()
            cand_end.synchronize()
            assert not ref_end.query(), 'the candidate stream waited for the busy stream: same hardware queue'
            torch.cuda.synchronize()


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_stream_resumes_bit_identically_from_a_snapshot(gpu, precision, tmp_path):
    """SURVEY.md section 5, "optional: bank snapshot for long streams" (the reference keeps no inference state,
    FeatureBank.py:10-51).  A loop stopped after frame 7, written with torch.save, read back into a NEW runner and continued
    gives the labels, the bank-size vector and the final bank (entries, birth frames, hit accumulators, peak / replace
    statistics) of the uninterrupted loop bit for bit -- through the eviction regime (budget reached before the cut)."""
    from tools import synth
    from vfloodnet_amd import AFB_URR
    from vfloodnet_amd.video_seg import ClipRunner
    T, cut = 15, 7
    model = AFB_URR(gpu, update_bank=True, precision=precision).to(gpu).eval()
    model.load_state_dict(synth.make_state_dict(20200212), strict=True)
    frames, m0 = synth.clip(5, T, 96, 160)
    frames = frames.to(gpu)
    onehot = synth.onehot(m0).unsqueeze(0).to(gpu)

    def run(runner, lo, hi):
        return [runner.step(frames[t:t + 1]).clone() for t in range(lo, hi)]

    a = ClipRunner(model, 2, 600, size=96)
    a.start(frames[0:1], onehot)
    la = run(a, 1, T)
    assert a.fb.replace_n.sum() > 0 and max(a.size_log[cut - 1]) >= a.fb.class_budget - 60      # evicting before the cut

    b = ClipRunner(model, 2, 600, size=96)
    b.start(frames[0:1], onehot)
    lb = run(b, 1, cut)
    b.launch(frames[cut:cut + 1])
    with pytest.raises(RuntimeError):
        b.snapshot()                                               # a frame in flight
    b.collect()
    b2 = ClipRunner(model, 2, 600, size=96)
    b2.start(frames[0:1], onehot)
    run(b2, 1, cut)
    path = str(tmp_path / 'stream.pt')
    torch.save(b2.snapshot(), path)
    del b, b2

    c = ClipRunner(model, 2, 600, size=96)
    c.resume(torch.load(path))
    assert c.t == cut - 1 and c.bank_sizes() == a.size_log[cut - 1]
    lc = run(c, cut, T)
    for t, (x, y) in enumerate(zip(la[:cut - 1], lb)):
        assert torch.equal(x, y), f'frame {t + 1} differs between two uninterrupted runs'
    for t, (x, y) in enumerate(zip(la[cut - 1:], lc)):
        assert torch.equal(x, y), f'frame {cut + t}: the resumed stream differs from the uninterrupted one'
    assert c.size_log == a.size_log[cut - 1:]
    for i in range(2):
        assert torch.equal(a.fb.keys[i], c.fb.keys[i]) and torch.equal(a.fb.values[i], c.fb.values[i])
        assert torch.equal(a.fb.info[i], c.fb.info[i])
    assert (a.fb.peak_n == c.fb.peak_n).all() and (a.fb.replace_n == c.fb.replace_n).all()

    bad = torch.load(path)
    bad['bank']['obj_n'] = 3
    with pytest.raises(ValueError):
        ClipRunner(model, 2, 600, size=96).resume(bad)


@pytest.mark.parametrize('prefetch', [False, True])
def test_grouped_frames_match_the_frame_by_frame_loop(gpu, prefetch):
    """ClipRunner.launch_group / AFB_URR.segment_group: with a key-frame interval (test_video_seg.py:110-112 with mem_every = n; BASELINE
    config C3) the frames between two memorize calls see the same bank, so they go through the query side, ONE memory read and the
    decoder as one batch.  Given the bank the frames are independent: same labels as the frame-by-frame loop up to the summation
    order of the larger GEMMs, the same bank-size vector, the same birth frames, hit accumulators that differ only where a
    probability sits on the 1e-3 threshold -- with and without the next group's frame-only side prefetched on the side stream."""
    from tools import synth
    from vfloodnet_amd import AFB_URR
    from vfloodnet_amd.video_seg import ClipRunner
    T, n = 12, 3
    model = AFB_URR(gpu, update_bank=True).to(gpu).eval()
    model.load_state_dict(synth.make_state_dict(20200212), strict=True)
    frames, m0 = synth.clip(4, T, 96, 160)
    frames = frames.to(gpu)
    onehot = synth.onehot(m0).unsqueeze(0).to(gpu)
    a = ClipRunner(model, 2, 250000, size=96, mem_every=n)
    a.start(frames[0:1], onehot)
    la = [a.step(frames[t:t + 1]).clone() for t in range(1, T)]

    b = ClipRunner(model, 2, 250000, size=96, mem_every=n)
    b.start(frames[0:1], onehot)
    with pytest.raises(ValueError):                               # frame 3 is memorised: the frames behind it must see its update
        b.launch_group([frames[t:t + 1] for t in range(1, 5)])
    lb, t = [], 1
    while t < T:
        g = min(n - (t - 1) % n, T - t)
        nxt = [frames[u:u + 1] for u in range(t + g, min(T, t + g + n))] if prefetch else None
        b.launch_group([frames[u:u + 1] for u in range(t, t + g)], next_frames=nxt or None)
        with pytest.raises(RuntimeError):
            b.launch(frames[t:t + 1])                             # a group in flight
        lb += [x.clone() for x in b.collect_group()]
        t += g
    assert len(lb) == len(la) == T - 1
    for t, (x, y) in enumerate(zip(la, lb)):
        inter = ((x == 1) & (y == 1)).sum().item()
        union = ((x == 1) | (y == 1)).sum().item()
        assert union == 0 or inter / union > 0.999, f'frame {t + 1}: grouped labels differ from the frame-by-frame loop ({inter}/{union})'
    assert a.size_log == b.size_log
    for i in range(2):
        ia, ib = a.fb.info[i], b.fb.info[i]
        assert torch.equal(ia[:, 0], ib[:, 0])
        assert float(((ia[:, 1] - ib[:, 1]).abs() > 2).float().mean()) < 0.01
        assert torch.allclose(a.fb.keys[i], b.fb.keys[i], atol=2e-4, rtol=1e-3)


def test_tuned_split_k_keeps_each_stream_on_its_own_workspace(gpu, monkeypatch):
    """Engine.autotune re-applies the measured (tile, split-K) choice to every launch of a plan; a split-K launch gets the workspace
    of the stream its LIST runs on.  The batch sets' frame-only lists run on the side stream in an inference plan
    (Engine.prefetch_group, beside memorize / update on the main stream): handed the main workspace they raced with memorize's
    split-K layers (C3 bf16x3 grouped: mIoU 0.39 on one run in two before the fix).  Every choice forced to a K split here."""
    from tools import synth
    from vfloodnet_amd import AFB_URR, engine, ops
    model = AFB_URR(gpu, update_bank=True).to(gpu).eval()
    model.load_state_dict(synth.make_state_dict(20200212), strict=True)
    eng = model.engine()
    p = eng.plan(96, 160, 2)
    qs = p.batch_set(3)
    b = qs.dec_batch()
    monkeypatch.setattr(engine, 'tune_desc', lambda d, bf, ws, cnt, **kw: (3, 2, 0))       # 64 x 64 tiles, K cut in two, from tile 0
    saved = [dict(t) for t in engine._TABLES]
    try:
        eng.autotune(96, 160, 2, only_missing=False)
    finally:
        for t, s_ in zip(engine._TABLES, saved):
            t.clear()
            t.update(s_)

    def inside(ptr_, t):
        return t.data_ptr() <= ptr_ < t.data_ptr() + t.numel() * t.element_size()
    seen = {'side': 0, 'main': 0}
    side = [qs.pre[n] for n in qs.sizes] + [q_.pre[n] for q_ in p.qsets for n in q_.sizes]
    main = [b.post, p.mem] + [L for q_ in p.qsets for L in q_.post]
    for kind, lists, ws in (('side', side, p.ws_q), ('main', main, p.ws)):
        for lst in lists:
            for l in lst:
                if l.fn is ops.conv2d_launch and l.args[0].ksplit > 1:
                    part = int(l.args[0].partial or 0)
                    assert inside(part, ws), f'{l.name}: split-K partials outside the {kind} stream\'s workspace'
                    seen[kind] += 1
    assert seen['side'] > 10 and seen['main'] > 10


def test_main_with_a_key_frame_interval_groups_the_frames(gpu, tmp_path, monkeypatch):
    """``video_seg.main --mem-every 3`` on 17 JPEG frames with overlays: the frames between two key frames go through the network as one
    batched pass (ClipRunner.launch_group); the mask PNGs agree with the frame-by-frame loop (VFN_MAIN_GROUP=0) up to the summation
    order of the batched convolutions, every frame and every overlay is written, and the bank-size vectors of the two runs are
    equal."""
    import argparse
    import numpy as np
    from PIL import Image
    from tools import synth
    from vfloodnet_amd import video_seg
    from vfloodnet_amd.data import save_seg_mask, color_palette
    monkeypatch.setenv('VFN_AUTOTUNE', '0')
    T, H, W = 17, 120, 200
    frames, m0 = synth.clip(21, T, H, W)
    fdir = tmp_path / 'frames'
    fdir.mkdir()
    for t in range(T):
        Image.fromarray((frames[t].permute(1, 2, 0).numpy() * 255).astype(np.uint8)).save(str(fdir / f'{t:05d}.jpg'), quality=95)
    ckpt = str(tmp_path / 'ckpt.pth')
    torch.save({'epoch': 0, 'model': synth.make_state_dict(20200212), 'loss': 0.0, 'seed': 20200212}, ckpt)
    out, sizes = {}, {}
    for tag, grp in (('grouped', '1'), ('frames', '0')):
        run = tmp_path / tag
        (run / 'output' / 'segs' / 'clip' / 'mask').mkdir(parents=True)
        save_seg_mask(m0.numpy().astype(np.uint8), str(run / 'output' / 'segs' / 'clip' / 'mask' / '00000.png'), color_palette)
        monkeypatch.chdir(run)
        monkeypatch.setenv('VFN_MAIN_GROUP', grp)
        args = argparse.Namespace(gpu=0, budget=250000, viz=True, model_path=ckpt, update_rate=0.1, merge_thres=0.95,
                                  test_path=str(fdir), test_name='clip', size=96, load_workers=2, mem_every=3, keep_labels=True)
        runner = video_seg.main(args, gpu)
        out[tag] = [np.array(Image.open(str(run / 'output' / 'segs' / 'clip' / 'mask' / f'{t:05d}.png'))) for t in range(T)]
        assert all((run / 'output' / 'segs' / 'clip' / 'overlay' / f'{t:05d}.png').is_file() for t in range(T))
        sizes[tag] = runner.kept_sizes.tolist() if runner is not None else None
    for t, (a, b) in enumerate(zip(out['grouped'], out['frames'])):
        inter, union = ((a == 1) & (b == 1)).sum(), ((a == 1) | (b == 1)).sum()
        assert union == 0 or inter / union > 0.995, f'frame {t}: {inter} / {union}'
    if sizes['grouped'] is not None:
        assert sizes['grouped'] == sizes['frames'] and len(sizes['grouped']) == T


def test_c3_shape_grouped_against_frame_by_frame(gpu):
    """BASELINE config C3 at its own size (720 x 1280 frames, network at 480 x 853, key frame every 5th), f32: 11 frames as groups of
    5 + 5 + 1 with the next group prefetched, against the frame-by-frame loop with its look-ahead -- labels within the summation-order
    noise of the larger GEMMs, bank-size vectors equal."""
    from tools import synth
    from vfloodnet_amd import AFB_URR
    from vfloodnet_amd.video_seg import ClipRunner
    T, n = 12, 5
    model = AFB_URR(gpu, update_bank=True).to(gpu).eval()
    model.load_state_dict(synth.make_state_dict(20200212), strict=True)
    frames, m0 = synth.clip(6, T, 720, 1280)
    frames = frames.to(gpu)
    onehot = synth.onehot(m0).unsqueeze(0).to(gpu)
    a = ClipRunner(model, 2, 250000, mem_every=n, postprocess=True)
    a.start(frames[0:1], onehot)
    la = [a.step(frames[t:t + 1], next_frames=[frames[u:u + 1] for u in range(t + 1, min(T, t + 4))]).clone() for t in range(1, T)]
    b = ClipRunner(model, 2, 250000, mem_every=n, postprocess=True)
    b.group_capture = n
    b.start(frames[0:1], onehot)
    lb, t = [], 1
    while t < T:
        g = min(n - (t - 1) % n, T - t)
        nxt = [frames[u:u + 1] for u in range(t + g, min(T, t + g + n))]
        b.launch_group([frames[u:u + 1] for u in range(t, t + g)], next_frames=nxt or None)
        lb += [x.clone() for x in b.collect_group()]
        t += g
    for t, (x, y) in enumerate(zip(la, lb)):
        inter = ((x == 1) & (y == 1)).sum().item()
        union = ((x == 1) | (y == 1)).sum().item()
        assert union == 0 or inter / union > 0.9995, f'frame {t + 1}: {inter} / {union}'
    assert a.size_log == b.size_log
