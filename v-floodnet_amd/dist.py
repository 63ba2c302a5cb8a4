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

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('VFN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def clips_of_rank(n_clips, rank, world):
    """Clip c -> rank c mod world."""
    return [c for c in range(n_clips) if c % world == rank]


def clips_per_rank(n_clips, world):
    return (n_clips + world - 1) // world


def gather_masks(local_labels, n_clips, rank, world):
    """local_labels: uint8 [n_local, T, H, W] for ``clips_of_rank`` (same T,H,W on every rank).
    Returns uint8 [n_clips, T, H, W] in clip order on every rank.  One collective."""
    if world == 1:
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


def run_sharded(run_one_clip, n_clips, rank, world, device):
    """``run_one_clip(c) -> uint8 [T,H,W]`` (host or device) for every clip of this rank; gather all."""
    mine = clips_of_rank(n_clips, rank, world)
    labels = [run_one_clip(c).to(device) for c in mine]
    if labels:
        local = torch.stack(labels, 0)
    else:
        local = None
    # ranks without a clip still take part in the collective; learn the shape from rank 0
    shape = torch.zeros(3, dtype=torch.int64, device=device)
    if rank == 0:
        shape = torch.tensor(local.shape[1:], dtype=torch.int64, device=device)
    if world > 1:
        dist.broadcast(shape, 0)
    if local is None:
        local = torch.zeros(0, *[int(x) for x in shape], dtype=torch.uint8, device=device)
    return gather_masks(local, n_clips, rank, world)
