"""Round-3 GPU tests: config C5 at FULL size (2000 x 1080p, bank to ~2.2 M entries/object) with a teacher-forced check of
the memory read at B ~ 1 M and ~ 2 M, bit-identity of the sequential and the pipelined loop, the refused undersized score
buffer, and the clip-sharded benchmark driver on two ranks."""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


def _rb(t):
    return t.bfloat16().float()


def _memread_reference_x3(fb, kv_q, obj, cols, chunk=1 << 18):
    """f64 value (on the device, torch) of what the bf16x3 memory read forms for query columns ``cols`` of object ``obj``:
    scores from split operands (ah+al)(bh+bl) - al*bl, f32 softmax over the bank, P^T V from split operands again
    (tests/test_kernels_gpu.py::test_memory_read_reduced_precision, chunked over the bank)."""
    B = int(fb._len_host[obj])
    K = fb._kbuf[obj, :B]                                   # [B,128] f32
    V = fb._vbuf[obj, :B]                                   # [B,512]
    q = kv_q[0, cols, :128].t().contiguous()                # [128, n]
    qh = _rb(q)
    ql = _rb(q - qh)
    s = torch.empty(B, len(cols), dtype=torch.float64, device=K.device)
    for b0 in range(0, B, chunk):
        k = K[b0:b0 + chunk]
        kh = _rb(k)
        kl = _rb(k - kh)
        s[b0:b0 + chunk] = ((kh.double() + kl.double()) @ (qh.double() + ql.double()) - kl.double() @ ql.double()) / math.sqrt(128)
    p = F.softmax(s.float(), dim=0)                         # the kernel's statistics are f32
    ph = _rb(p)
    pl = _rb(p - ph)
    mem = torch.zeros(512, len(cols), dtype=torch.float64, device=K.device)
    for b0 in range(0, B, chunk):
        v = V[b0:b0 + chunk].t()                            # [512, c]
        vh = _rb(v)
        vl = _rb(v - vh)
        pp_h, pp_l = ph[b0:b0 + chunk].double(), pl[b0:b0 + chunk].double()
        mem += (vh.double() + vl.double()) @ (pp_h + pp_l) - vl.double() @ pp_l
    return mem.float()                                      # [512, n]


def test_c5_full_size_2000_frames_bf16x3(gpu, sd):
    """BASELINE config C5 as written: a 1920x1080 stream of 2000 frames, reference semantics (resize to 480p), every frame
    memorised, bf16x3, the bank sized so that nothing is evicted.  Asserted: the first frames against the f32 CPU oracle
    (mIoU >= 0.99, bank sizes within 3 entries), monotone growth to ~2.2 M entries per object with replace_n == 0
    (FeatureBank.py:102-103), idempotence of the largest-component filter, and -- teacher-forced on the live bank at
    B ~ 1 M and at the end (~2.2 M) -- sampled query columns of the memory read against an f64 evaluation of the same
    split operands."""
    from tools import synth
    from vfloodnet_amd import AFB_URR, ops
    from vfloodnet_amd.video_seg import ClipRunner
    from oracle import afb_urr_ref as O
    T, H, W, n_ref = 2001, 1080, 1920, 5
    budget = 2 * int(1.25 * 2 * (T + 2) * 1620) + 4         # class_budget >= T * HW: the bank only grows (bench.py C5)
    frames, m0 = synth.clip_on_device(9, T, H, W, gpu)
    torch.set_num_threads(16)
    ref = O.run_clip(sd, frames[:n_ref + 1].cpu(), m0, size=480, budget=budget)
    model = AFB_URR(gpu, update_bank=True, precision='bf16x3').to(gpu).eval()
    model.load_state_dict(sd, strict=True)
    eng = model.engine()
    runner = ClipRunner(model, 2, budget, size=480, postprocess=True)
    runner.start(frames[0:1], synth.onehot(m0).unsqueeze(0).to(gpu))
    sizes, ious, checked = [], [], []
    cols = torch.tensor([0, 1, 63, 64, 127, 128, 500, 811, 1023, 1400, 1618, 1619], device=gpu)

    def check_memory_read(tag):
        torch.cuda.synchronize()
        plan = eng.plan(480, 853, 2)
        eng._memory_read(plan, runner.fb, False)            # the live bank against the query that is in the plan
        torch.cuda.synchronize()
        for obj in (0, 1):
            want = _memread_reference_x3(runner.fb, plan.kv_q, obj, cols)
            got = plan.dec_in[obj].reshape(plan.HW, -1)[cols, :512].t()
            err = (got - want).abs().max().item()
            scale = max(1.0, want.abs().max().item())
            checked.append((tag, obj, int(runner.fb._len_host[obj]), err / scale))
            assert err < 2e-4 * scale, (tag, obj, err, scale)

    import time
    t0 = time.perf_counter()
    for t in range(1, T):
        lab = runner.step(frames[t:t + 1], next_frame=frames[t + 1:t + 2] if t + 1 < T else None)
        sizes.append(runner.bank_sizes())
        if t <= n_ref:
            ious.append(miou(runner._label_dev.cpu(), ref['labels'][t]))
        if t in (n_ref, T // 2, T - 1):
            post = torch.from_numpy(lab.numpy().copy())
            assert set(post.unique().tolist()) <= {0, 1}
            assert torch.equal(ops.postprocess_pred_device(post.to(gpu)).cpu(), post)
        if len(checked) == 0 and min(sizes[-1]) >= 1_000_000:
            check_memory_read('B~1M')
    secs = time.perf_counter() - t0
    check_memory_read('end')
    print(f'C5 full size bf16x3: {T - 1} frames in {secs:.1f} s ({(T - 1) / secs:.1f} frames/s incl. checks); mIoU(first {n_ref}) '
          f'{[round(x, 5) for x in ious]}; bank {sizes[0]} -> {sizes[-1]}; memory read rel. err {checked}')
    drift = max(abs(a - b) for x, y in zip(sizes[:n_ref], ref['bank_sizes']) for a, b in zip(x, y))
    assert drift <= 3 and min(ious) >= 0.99, (ious, sizes[:n_ref], ref['bank_sizes'])
    assert all(sizes[i][c] >= sizes[i - 1][c] for i in range(1, len(sizes)) for c in (0, 1))     # the bank only grows
    assert float(runner.fb.replace_n.sum()) == 0.0
    assert min(sizes[-1]) >= 1_800_000 and max(sizes[-1]) <= (T - 1) * 1620 + 1620
    assert len(checked) == 4 and checked[0][2] >= 1_000_000 and checked[-1][2] >= 1_800_000


def test_sequential_and_pipelined_loop_are_bit_identical(gpu, sd):
    """``ClipRunner.step`` and ``launch`` / ``collect`` (frame t+1 enqueued before the host has seen frame t) slice the bank
    alike (FeatureBank.len_upper is a function of the update history, not of the host's bookkeeping), so labels, bank
    sizes, keys and hit statistics must be EQUAL, not merely close."""
    from tools import synth
    from vfloodnet_amd import AFB_URR
    from vfloodnet_amd.video_seg import ClipRunner
    model = AFB_URR(gpu, update_bank=True).to(gpu).eval()
    model.load_state_dict(sd, strict=True)
    T, H, W = 14, 240, 432
    frames, m0 = synth.clip(3, T, H, W)
    frames = frames.to(gpu)
    onehot = synth.onehot(m0).unsqueeze(0).to(gpu)
    res = []
    for mode in ('step', 'pipelined'):
        r = ClipRunner(model, 2, 6000, size=240)            # small budget: eviction from frame ~6 on
        r.start(frames[0:1], onehot)
        labs, sizes = [], []
        if mode == 'step':
            for t in range(1, T):
                labs.append(torch.from_numpy(r.step(frames[t:t + 1], next_frames=[frames[u:u + 1] for u in range(t + 1, min(T, t + 4))]).numpy().copy()))
                sizes.append(r.bank_sizes())
        else:
            for t in range(1, T):
                r.launch(frames[t:t + 1], next_frames=[frames[u:u + 1] for u in range(t + 1, min(T, t + 4))])
                if len(r._pending) == 2:
                    labs.append(torch.from_numpy(r.collect().numpy().copy()))
                    sizes.append(r.bank_sizes())
            while r._pending:
                labs.append(torch.from_numpy(r.collect().numpy().copy()))
                sizes.append(r.bank_sizes())
        torch.cuda.synchronize()
        res.append((labs, sizes, [k.clone() for k in r.fb.keys], [i.clone() for i in r.fb.info], r.fb.replace_n.copy()))
    a, b = res
    assert a[1] == b[1], (a[1], b[1])
    assert all(torch.equal(x, y) for x, y in zip(a[0], b[0]))
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2])) and all(torch.equal(x, y) for x, y in zip(a[3], b[3]))
    assert np.array_equal(a[4], b[4]) and a[4].sum() > 0     # eviction took part


def test_undersized_score_buffer_is_refused(gpu):
    """vfn_bank_scan / vfn_memread_apply check stride_scores against the capacity of the key slab."""
    from vfloodnet_amd import _lib
    from vfloodnet_amd._lib import BankScanDesc, ptr
    L = _lib.lib()
    HW, cap = 200, 1000
    q = torch.randn(HW, 640, device=gpu)
    k = torch.randn(1, cap, 128, device=gpu)
    d = BankScanDesc()
    blen = torch.tensor([cap], dtype=torch.int32, device=gpu)
    part = torch.empty(1, 4, HW, 2, device=gpu)
    work = torch.zeros(4, dtype=torch.int32, device=gpu)
    need = ((cap + 63) // 64) * ((HW + 127) // 128) * 8192
    d.q, d.bank_k, d.bank_len, d.part = ptr(q), ptr(k), ptr(blen), ptr(part)
    d.stride_q, d.stride_k, d.stride_rs, d.scale = 0, cap * 128, 0, 1.0
    d.ldq, d.q_per_obj, d.HW, d.obj_n, d.nsplit, d.mode, d.precision = 640, 0, HW, 1, 4, 0, 0
    d.work_counter = ptr(work)
    small = torch.empty(need - 8192, device=gpu)
    d.scores, d.stride_scores = ptr(small), small.numel()
    assert L.vfn_bank_scan(_lib.C.byref(d), _lib.stream()) != 0
    ok = torch.empty(need, device=gpu)
    d.scores, d.stride_scores = ptr(ok), ok.numel()
    assert L.vfn_bank_scan(_lib.C.byref(d), _lib.stream()) == 0
    torch.cuda.synchronize()


def test_batch_video_seg_two_ranks_on_one_device(gpu, sd, tmp_path):
    """``python -m vfloodnet_amd.batch_video_seg --benchmark_path DIR --gpus 2``: two clip folders of different size, one per
    rank (both ranks on the one device, gloo), through the real ``video_seg.main``; the gathered masks rank 0 saves must
    equal the mask PNGs the ranks wrote, and a single-process run of the same directory must produce the same masks."""
    from PIL import Image
    from tools import synth
    from vfloodnet_amd.data import save_seg_mask, color_palette
    bench = tmp_path / 'bench'
    shapes = {'clip_a': (5, 96, 160), 'clip_b': (4, 120, 200)}
    for i, (name, (T, H, W)) in enumerate(shapes.items()):
        frames, m0 = synth.clip(11 + i, T, H, W)
        d = bench / name
        d.mkdir(parents=True)
        for t in range(T):
            Image.fromarray((frames[t].permute(1, 2, 0).numpy() * 255).astype(np.uint8)).save(str(d / f'{t:05d}.jpg'), quality=95)
        for run in ('two', 'one'):
            md = tmp_path / run / 'output' / 'segs' / name / 'mask'
            md.mkdir(parents=True)
            save_seg_mask(m0.numpy().astype(np.uint8), str(md / '00000.png'), color_palette)
    ckpt = str(tmp_path / 'ckpt.pth')
    torch.save({'epoch': 0, 'model': sd, 'loss': 0.0, 'seed': SEED}, ckpt)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(PYTHONPATH=ROOT, VFN_DIST_BACKEND='gloo', VFN_SINGLE_DEVICE='1')
    got = {}
    for run, gpus in (('two', '2'), ('one', '1')):
        npz = str(tmp_path / run / 'gathered.npz')
        r = subprocess.run([sys.executable, '-m', 'vfloodnet_amd.batch_video_seg', '--benchmark_path', str(bench), '--model_path', ckpt,
                            '--gpus', gpus, '--size', '96', '--save-gathered', npz, '--load-workers', '0'],
                           cwd=str(tmp_path / run), env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        summary = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"clips"')][0]
        assert summary['clips'] == 2 and summary['ranks'] == int(gpus)
        assert [c['rank'] for c in summary['per_clip']] == ([0, 1] if gpus == '2' else [0, 0])
        got[run] = np.load(npz)
        for name, (T, H, W) in shapes.items():
            m = got[run][name]
            assert m.shape == (T, H, W)
            for t in range(T):
                png = np.array(Image.open(str(tmp_path / run / 'output' / 'segs' / name / 'mask' / f'{t:05d}.png')))
                assert np.array_equal(png, m[t]), (run, name, t)
    for name in shapes:
        assert np.array_equal(got['two'][name], got['one'][name])
