"""Clip-sharded multi-GPU driver: one process per GPU, one RCCL all-gather of the label masks.

The reference has no distributed code; its only multi-clip entry point is the *sequential*
loop of ``scripts/batch_test_video_seg.py:40-47``.  Clips are independent (all state -- the
``FeatureBank`` -- is per clip, weights are replicated), so clip ``c`` goes to rank
``c mod world`` and nothing is exchanged during inference.  At the end each rank contributes
its ``uint8[clips_per_rank, T, H, W]`` label block to a single ``all_gather`` (backend ``nccl``
= RCCL over xGMI on ROCm; ``gloo`` in the CPU tests).  A single long stream does not shard:
``bench.py --gpus N`` then runs N independent replicas.
"""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def forced():
    """VFN_FORCE_DIST=1: bring the process group up and run every collective of the N > 1 code path even with ONE rank -- how the
    RCCL branches (``all_gather_into_tensor`` on device tensors, the device-side all-reduce / all-gather of bench.py) are
    exercised on a 1-GPU box (tests/test_round5_gpu.py) before they meet an 8-GPU node (BASELINE.json config 4)."""
    return os.environ.get('VFN_FORCE_DIST') == '1'


def active(world):
    """Do the collectives run?  More than one rank, or the forced single-rank group above."""
    return world > 1 or (forced() and dist.is_initialized())


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process, unless VFN_FORCE_DIST=1)."""
    rank, local_rank, world = env_rank_world()
    if (world > 1 or forced()) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('VFN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:                 # (a forced single rank started without a launcher)
            os.environ['MASTER_PORT'] = str(free_port())
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # RCCL writes its version banner (and anything NCCL_DEBUG asks for) to stdout, where rank 0's ONE JSON line lives: send it to a file
        os.environ.setdefault('NCCL_DEBUG_FILE', '/tmp/vfn_rccl_%h_%p.log')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def host_threads_per_rank(local_world):
    """Host threads a rank may use when ``local_world`` ranks share the machine (>= 1, <= 16: more never helped the
    CPU-side work of this path, and an idle OpenMP pool spinning on every core starves the launch threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return max(1, min(16, n // max(1, local_world)))


def pin_rank_threads(local_rank, local_world):
    """Give this rank its own contiguous slice of the CPUs the process may run on and cap its thread pools to it.
    Call before anything touches the GPU or spawns threads (DataLoader workers and PNG writer threads inherit the mask).
    Returns the CPU set (empty = affinity not available / left alone).  VFN_NO_PIN=1 switches it off."""
    if local_world <= 1 or os.environ.get('VFN_NO_PIN') == '1':
        return set()
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return set()
    per = len(cpus) // local_world
    if per < 1:
        return set()
    mine = set(cpus[local_rank * per:(local_rank + 1) * per])
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return set()
    n = str(max(1, min(16, per)))
    for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):
        os.environ[var] = n
    torch.set_num_threads(int(n))
    return mine


def spawn_ranks(argv, n, poll_s=0.1, extra_env=None):
    """Start ``n`` ranks of ``argv`` (a full command line) as CHILD processes of a parent that never touches the GPU
    (no exec of an initialised process), one per device; wait for all of them.  The first rank that exits non-zero
    takes the others down with it (exact PIDs we started) -- otherwise the survivors would sit in a barrier or in
    ``init_process_group`` until the collective timeout.  Returns the first non-zero exit code, or 0."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', str(host_threads_per_rank(n)))
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env))
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    if code != 0:
                        rc = code
                        break
            if live and rc == 0:
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def clips_of_rank(n_clips, rank, world):
    """Clip c -> rank c mod world."""
    return [c for c in range(n_clips) if c % world == rank]


def clips_per_rank(n_clips, world):
    return (n_clips + world - 1) // world


def gather_masks(local_labels, n_clips, rank, world):
    """local_labels: uint8 [n_local, T, H, W] for ``clips_of_rank`` (same T,H,W on every rank).
    Returns uint8 [n_clips, T, H, W] in clip order on every rank.  One collective."""
    if not active(world):
        return local_labels
    per = clips_per_rank(n_clips, world)
    T, H, W = local_labels.shape[1:]
    send = torch.zeros(per, T, H, W, dtype=torch.uint8, device=local_labels.device)
    send[:local_labels.shape[0]] = local_labels
    recv = torch.empty(world * per, T, H, W, dtype=torch.uint8, device=local_labels.device)
    if dist.get_backend() == 'nccl':
        dist.all_gather_into_tensor(recv, send)
    else:                                   # gloo (CPU tests, or a 1-GPU smoke run of the N > 1 code path)
        send_h = send.cpu()
        parts = [torch.empty_like(send_h) for _ in range(world)]
        dist.all_gather(parts, send_h)
        recv.copy_(torch.stack(parts, 0).view_as(recv))
    recv = recv.view(world, per, T, H, W)
    out = torch.empty(n_clips, T, H, W, dtype=torch.uint8, device=local_labels.device)
    for c in range(n_clips):
        out[c] = recv[c % world, c // world]
    return out


def gather_bank_sizes(local_sizes, n_clips, rank, world, device):
    """The feature bank's live entries per object after every frame, beside the masks ("plus an int32[T] bank-size
    vector"): ``local_sizes`` = list of int [T_c, obj_n] tensors (or nested lists) for ``clips_of_rank``.  Returns the list of
    int32 [T_c, obj_n] tensors (host) in clip order on every rank.  One tiny all-gather (a few KB), outside any timed bracket:
    the blocks are padded to the longest clip, row 0 of every block carries the clip's (T_c, obj_n)."""
    if n_clips <= 0:
        return []
    per = clips_per_rank(n_clips, world)
    loc = []
    for x in local_sizes:
        x = torch.as_tensor(x, dtype=torch.int32).cpu()
        loc.append(x.view(-1, 1) if x.dim() == 1 else x.reshape(x.shape[0], x.shape[1] if x.dim() > 1 else 0))   # ([T, 0]: a clip without sizes)
    dims = torch.zeros(2, dtype=torch.int64)
    if loc:
        dims = torch.tensor([max(int(x.shape[0]) for x in loc), max(int(x.shape[1]) for x in loc)], dtype=torch.int64)
    on_dev = active(world) and dist.get_backend() == 'nccl'
    if active(world):                     # (the padded block shape: the maximum over all ranks)
        d_ = dims.to(device) if on_dev else dims
        dist.all_reduce(d_, op=dist.ReduceOp.MAX)
        dims = d_.cpu()
    T, K = int(dims[0]), max(2, int(dims[1]))
    send = torch.zeros(per, T + 1, K, dtype=torch.int32)
    for i, x in enumerate(loc):
        send[i, 0, 0], send[i, 0, 1] = int(x.shape[0]), int(x.shape[1])
        send[i, 1:1 + x.shape[0], :x.shape[1]] = x
    if active(world):
        if on_dev:
            send_d = send.to(device)
            recv = torch.empty(world * per, T + 1, K, dtype=torch.int32, device=device)
            dist.all_gather_into_tensor(recv, send_d)
            recv = recv.cpu()
        else:
            parts = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(parts, send)
            recv = torch.stack(parts, 0)
        recv = recv.view(world, per, T + 1, K)
    else:
        recv = send.view(1, per, T + 1, K)
    out = []
    for c in range(n_clips):
        blk = recv[c % world, c // world]
        t, k = int(blk[0, 0]), int(blk[0, 1])
        out.append(blk[1:1 + t, :k].clone())
    return out


def run_sharded(run_one_clip, n_clips, rank, world, device):
    """``run_one_clip(c) -> uint8 [T,H,W]`` (host or device) for every clip of this rank; gather all."""
    if n_clips <= 0:                      # nothing to run: no collective is entered (every rank sees the same n_clips)
        return torch.zeros(0, 0, 0, 0, dtype=torch.uint8, device=device)
    mine = clips_of_rank(n_clips, rank, world)
    labels = [run_one_clip(c).to(device) for c in mine]
    if labels:
        local = torch.stack(labels, 0)
    else:
        local = None
    # ranks without a clip still take part in the collective; learn the shape from rank 0 (clip 0 -> rank 0: with at
    # least one clip rank 0 always has one)
    shape = torch.zeros(3, dtype=torch.int64, device=device)
    if rank == 0:
        shape = torch.tensor(local.shape[1:], dtype=torch.int64, device=device)
    if active(world):
        dist.broadcast(shape, 0)
    if local is None:
        local = torch.zeros(0, *[int(x) for x in shape], dtype=torch.uint8, device=device)
    return gather_masks(local, n_clips, rank, world)


def gather_ragged(local_labels, n_clips, rank, world, device):
    """Clips of different length / frame size (a benchmark directory): ``local_labels`` = list of uint8 [T_c,H_c,W_c] for
    ``clips_of_rank``.  The shapes travel first (3 integers per clip), every block is padded to the largest clip and the
    masks then cross in the ONE all-gather of ``gather_masks``.  Returns the list of uint8 [T_c,H_c,W_c] in clip order."""
    if n_clips <= 0:                      # an empty benchmark directory: nothing to gather, no collective
        return []
    per = clips_per_rank(n_clips, world)
    # (gloo gathers host tensors only; RCCL device tensors only)
    shp = torch.zeros(per, 3, dtype=torch.int64, device=device if (active(world) and dist.get_backend() == 'nccl') else 'cpu')
    for i, l in enumerate(local_labels):
        shp[i] = torch.tensor(l.shape, dtype=torch.int64)
    if active(world):
        parts = [torch.empty_like(shp) for _ in range(world)]
        dist.all_gather(parts, shp)
        shapes = torch.stack(parts, 0).cpu()
    else:
        shapes = shp.unsqueeze(0).cpu()
    T, H, W = (int(shapes[..., k].max()) for k in range(3))
    block = torch.zeros(len(local_labels), T, H, W, dtype=torch.uint8, device=device)
    for i, l in enumerate(local_labels):
        block[i, :l.shape[0], :l.shape[1], :l.shape[2]] = l.to(device)
    allm = gather_masks(block, n_clips, rank, world)
    out = []
    for c in range(n_clips):
        t, h, w = (int(v) for v in shapes[c % world, c // world])
        out.append(allm[c, :t, :h, :w])
    return out
