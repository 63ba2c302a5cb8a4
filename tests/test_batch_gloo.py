"""The clip-sharded benchmark driver (``vfloodnet_amd.batch_video_seg``; seam: scripts/batch_test_video_seg.py:40-47) on a
CPU-only machine: directory walk, clip c -> rank c mod N, the ragged mask all-gather over gloo with world_size 2, the
self-launcher's fail-fast behaviour and the per-rank CPU pinning.  (The per-clip worker is a stand-in here -- the real one is
HIP-only; tests/test_round3_gpu.py runs the real loop on two ranks.)"""
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_benchmark(root, shapes):
    """shapes: list of (T, H, W); clip folders clip_00, clip_01, ... of PNG frames."""
    rng = np.random.RandomState(7)
    for c, (T, H, W) in enumerate(shapes):
        d = os.path.join(root, f'clip_{c:02d}')
        os.makedirs(d)
        for t in range(T):
            Image.fromarray(rng.randint(0, 256, (H, W, 3)).astype(np.uint8)).save(os.path.join(d, f'{t:05d}.png'))


def _standin_clip(args, device):
    """'Segments' a clip folder by thresholding the red channel: uint8 [T,H,W]."""
    from glob import glob
    files = sorted(glob(os.path.join(args.test_path, '*.png')))
    return torch.stack([torch.from_numpy((np.array(Image.open(f).convert('RGB'))[:, :, 0] > 127).astype(np.uint8)) for f in files], 0)


def _standin_sizes(m):
    """A stand-in for the per-frame bank sizes int32 [T, 2] (they travel beside the masks): derived from the masks."""
    T = m.shape[0]
    water = m.reshape(T, -1).sum(1).to(torch.int32)
    return torch.stack([water, torch.full((T,), m.shape[1] * m.shape[2], dtype=torch.int32) - water], 1)


def _worker(rank, world, port, bench_dir, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import batch_video_seg as B
    args = B.get_args(['--benchmark_path', bench_dir, '--model_path', 'unused.pth', '--gpus', str(world)])
    ran = []

    def clip(a, d):
        ran.append(a.test_name)
        m = _standin_clip(a, d)
        return m, _standin_sizes(m)
    names, masks = B.run(args, run_clip=clip, device=torch.device('cpu'), backend='gloo')
    q.put((rank, ran, names, [m.numpy() for m in masks], sorted(os.sched_getaffinity(0)), [z.numpy() for z in B.run.last_bank_sizes]))


@pytest.mark.parametrize('shapes', [[(3, 10, 12), (2, 8, 16)], [(3, 10, 12), (2, 8, 16), (4, 6, 6)]], ids=['2clips', '3clips_ragged'])
def test_batch_driver_world2_gloo(tmp_path, shapes):
    bench = str(tmp_path / 'bench')
    os.makedirs(bench)
    _make_benchmark(bench, shapes)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bench, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    names = [f'clip_{c:02d}' for c in range(len(shapes))]
    assert res[0][1] == names[0::2] and res[1][1] == names[1::2]             # clip c -> rank c mod 2
    args = type('A', (), {})()
    for rank, _, got_names, masks, _, sizes in res:
        assert got_names == names
        for c, m in enumerate(masks):                                        # every rank holds every clip's masks
            args.test_path = os.path.join(bench, names[c])
            assert m.shape == shapes[c] and np.array_equal(m, _standin_clip(args, None).numpy())
            # ... and every clip's int32 [T, obj_n] bank-size vector (ragged T), in clip order
            assert sizes[c].dtype == np.int32 and np.array_equal(sizes[c], _standin_sizes(torch.from_numpy(m)).numpy())
    if len(os.sched_getaffinity(0)) >= 2:                                     # disjoint CPU slices per rank
        assert not (set(res[0][4]) & set(res[1][4]))


def test_spawn_ranks_kills_the_survivors_when_one_rank_fails(tmp_path):
    """Rank 1 dies at bring-up; rank 0 would sit in its barrier for the collective timeout.  The launcher must return rank
    1's exit code within seconds and leave no child behind."""
    script = tmp_path / 'rank.py'
    pidfile = tmp_path / 'pid0'
    script.write_text(
        'import os, sys, time\n'
        'if os.environ["RANK"] == "1":\n'
        '    sys.exit(3)\n'
        f'open({str(pidfile)!r}, "w").write(str(os.getpid()))\n'
        'time.sleep(120)\n')
    sys.path.insert(0, ROOT)
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import dist as vdist
    t0 = time.time()
    rc = vdist.spawn_ranks([sys.executable, str(script)], 2)
    assert rc == 3 and time.time() - t0 < 30
    pid0 = int(pidfile.read_text()) if pidfile.exists() else None
    if pid0 is not None:
        time.sleep(0.2)
        assert not os.path.exists(f'/proc/{pid0}') or open(f'/proc/{pid0}/stat').read().split()[2] == 'Z'


def test_bench_self_launch_uses_the_fail_fast_launcher():
    """bench.py --gpus 2 --launch-check with a rank that cannot come up (bad MASTER_PORT is not needed: an argument the
    children reject) returns non-zero promptly instead of hanging."""
    env = dict(os.environ, VFN_BENCH_FAIL_RANK='1')
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-check'], env=env,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode != 0 and time.time() - t0 < 120, (r.stdout, r.stderr[-500:])
