"""Clip sharding + the single mask all-gather, world_size 2 over gloo on CPU (RCCL on the GPU box)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_clip(c, T=3, H=5, W=7):
    g = torch.Generator().manual_seed(100 + c)
    return (torch.rand(T, H, W, generator=g) > 0.5).to(torch.uint8)


def _fake_sizes(c):
    """Bank sizes int32 [T_c, 2] of clip c: T_c differs from clip to clip."""
    T = 3 + c % 3
    return (torch.arange(T * 2, dtype=torch.int32).view(T, 2) * (c + 1) + 1620)


def _worker(rank, world, port, n_clips, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import dist as vdist
    r, lr, w = vdist.init(backend='gloo')
    assert (r, w) == (rank, world)
    mine = vdist.clips_of_rank(n_clips, rank, world)
    out = vdist.run_sharded(_fake_clip, n_clips, rank, world, torch.device('cpu'))
    ok = all(torch.equal(out[c], _fake_clip(c)) for c in range(n_clips))
    # the int32 bank-size vectors beside the masks: ragged lengths, clip order, every rank holds all of them
    sizes = vdist.gather_bank_sizes([_fake_sizes(c) for c in mine], n_clips, rank, world, torch.device('cpu'))
    ok = ok and len(sizes) == n_clips and all(z.dtype == torch.int32 and torch.equal(z, _fake_sizes(c)) for c, z in enumerate(sizes))
    q.put((rank, mine, ok, tuple(out.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_clips', [2, 3, 8])
def test_sharded_gather_world2(n_clips):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == [c for c in range(n_clips) if c % 2 == 0]
    assert res[1][1] == [c for c in range(n_clips) if c % 2 == 1]
    assert all(r[2] for r in res)
    assert all(r[3] == (n_clips, 3, 5, 7) for r in res)


def test_single_process_is_identity():
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import dist as vdist
    out = vdist.run_sharded(_fake_clip, 3, 0, 1, torch.device('cpu'))
    assert all(torch.equal(out[c], _fake_clip(c)) for c in range(3))
    sizes = vdist.gather_bank_sizes([_fake_sizes(c) for c in range(3)], 3, 0, 1, torch.device('cpu'))
    assert all(torch.equal(z, _fake_sizes(c)) for c, z in enumerate(sizes))
    assert vdist.gather_bank_sizes([], 0, 0, 1, torch.device('cpu')) == []


def test_no_clips_reports_instead_of_crashing(tmp_path):
    """Empty inputs: zero clips enter no collective and return empty results; an empty benchmark directory is a
    clear error raised before any collective."""
    import argparse
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import dist as vdist, batch_video_seg
    out = vdist.run_sharded(_fake_clip, 0, 0, 1, torch.device('cpu'))
    assert out.numel() == 0 and out.dtype == torch.uint8
    assert vdist.gather_ragged([], 0, 0, 1, torch.device('cpu')) == []
    empty = tmp_path / 'bench'
    empty.mkdir()
    args = argparse.Namespace(benchmark_path=str(empty), gpu=0, save_gathered=None)
    env = {k: os.environ.pop(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE') if k in os.environ}
    try:
        with pytest.raises(ValueError, match='no clip sub-folders'):
            batch_video_seg.run(args, run_clip=lambda a, d: None, device=torch.device('cpu'), backend='gloo')
    finally:
        os.environ.update(env)
