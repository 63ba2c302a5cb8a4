"""A directory of clips, sharded over the GPUs of one node: drop-in for ``scripts/batch_test_video_seg.py``.

The reference walks ``--benchmark_path`` and calls ``test_video_seg.main`` on one sub-folder after the other
(``scripts/batch_test_video_seg.py:40-47``).  Clips share nothing (the ``FeatureBank`` is per clip, weights are
replicated), so here sub-folder ``c`` (sorted order) goes to rank ``c mod N``; every rank runs ``video_seg.main`` on
its clips -- same files under ``./output/segs/<name>/`` as the sequential loop -- and the label masks of all clips
meet in ONE all-gather at the end (RCCL over xGMI; ``gloo`` in the CPU tests), so that every rank -- rank 0 in
particular -- holds the whole benchmark's masks without going back to the disk.

    python -m vfloodnet_amd.batch_video_seg --benchmark_path DIR --model_path CKPT --gpus N [--save-gathered masks.npz]

``--gpus N`` without a launcher: this process starts the N ranks itself as child processes before anything touches
the GPU (``dist.spawn_ranks``: the first rank that fails takes the others down); under ``torch.distributed.run`` the ranks
come from the environment.  Each rank pins its host threads to its own slice of the CPUs first (``dist.pin_rank_threads``).
"""
import argparse
import json
import os
import sys
from glob import glob

import torch


def get_args(argv=None):
    """scripts/batch_test_video_seg.py:9-27 (+ the harness options of ``video_seg.get_args``)."""
    parser = argparse.ArgumentParser(description='Test Video Segmentation Benchmark (clip-sharded, MI355X-native)')
    parser.add_argument('--gpu', type=int, default=0, help='GPU card id (single-process runs; ranks use LOCAL_RANK).')
    parser.add_argument('--gpus', type=int, default=1, help='Ranks = GPUs of this node to shard the clips over.')
    parser.add_argument('--budget', type=int, default='250000',
                        help='Max number of features that feature bank can store. Default: 300000')
    parser.add_argument('--viz', action='store_true', help='Visualize data.')
    parser.add_argument('--model_path', '--model-path', dest='model_path', type=str, required=True,
                        help='Path to the checkpoint (default: none)')
    parser.add_argument('--update-rate', type=float, default=0.1, help='Update Rate. Impact of merging new features.')
    parser.add_argument('--merge-thres', type=float, default=0.95,
                        help='Merging Rate. If similarity higher than this, then merge, else append.')
    parser.add_argument('--benchmark_path', '--benchmark-path', '--test-path', dest='benchmark_path', type=str, required=True,
                        help='Benchmark Path: every sub-folder is one clip')
    parser.add_argument('--decode', choices=['device', 'pil'], default='device')
    parser.add_argument('--size', type=int, default=480)
    parser.add_argument('--mem-every', type=int, default=1)
    parser.add_argument('--load-workers', type=int, default=4)
    parser.add_argument('--png-workers', type=int, default=4)
    parser.add_argument('--save-gathered', type=str, default=None,
                        help='rank 0 writes the gathered masks of all clips here (.npz: one uint8 [T,H,W] array per clip name)')
    return parser.parse_args(argv)


def list_clips(benchmark_path):
    """scripts/batch_test_video_seg.py:40-42: sorted sub-folders; name = the folder's name."""
    test_list = sorted(glob(os.path.join(benchmark_path, '*/')))
    return [(p.split('/')[-2], p) for p in test_list]


def run_clip_on_gpu(args, device):
    """One clip through ``video_seg.main``; returns its uint8 [T,H,W] label masks (device)."""
    from . import video_seg
    args.keep_labels = True
    runner = video_seg.main(args, device)
    return runner.kept_labels, runner.kept_sizes


def run(args, run_clip=run_clip_on_gpu, device=None, backend=None):
    """The sharded loop.  ``run_clip(args, device) -> uint8 [T,H,W]`` with ``args.test_name`` / ``args.test_path`` set
    (tests pass a stand-in so that walking, sharding and the gather run on a CPU-only machine).
    Returns (clip names, list of uint8 [T,H,W] masks in clip order) on every rank."""
    from . import dist as vdist
    rank, local_rank, world = vdist.env_rank_world()
    vdist.pin_rank_threads(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    rank, local_rank, world = vdist.init(backend=backend)
    if device is None:
        if not torch.cuda.is_available():
            raise ValueError('CUDA is required. --gpu must be >= 0.')      # scripts/batch_test_video_seg.py:34-37
        idx = 0 if os.environ.get('VFN_SINGLE_DEVICE') == '1' else (local_rank if world > 1 else args.gpu)
        device = torch.device('cuda', idx)
        torch.cuda.set_device(device)
    assert os.path.isdir(args.benchmark_path)
    clips = list_clips(args.benchmark_path)
    if not clips:                         # every rank sees the same (empty) directory: leave before any collective
        if vdist.active(world):
            import torch.distributed as dist
            dist.destroy_process_group()
        raise ValueError(f'no clip sub-folders in {args.benchmark_path}')
    mine = vdist.clips_of_rank(len(clips), rank, world)
    local, local_sizes = [], []
    for c in mine:
        args.test_name, args.test_path = clips[c]
        print(f'[rank {rank}] Process video', args.test_name, 'from path', args.test_path, flush=True)
        res = run_clip(args, device)
        # ``run_clip`` returns the masks, or (masks, int [T, obj_n] bank sizes per frame): SURVEY.md 8(e) sends both
        lab, sz = res if isinstance(res, tuple) else (res, torch.zeros(int(res.shape[0]), 0, dtype=torch.int32))
        local.append(lab)
        local_sizes.append(sz)
    masks = vdist.gather_ragged(local, len(clips), rank, world, device)
    sizes = vdist.gather_bank_sizes(local_sizes, len(clips), rank, world, device)      # one more, tiny all-gather beside the masks
    run.last_bank_sizes = sizes
    names = [n for n, _ in clips]
    if rank == 0:
        summary = {'clips': len(clips), 'ranks': world,
                   'per_clip': [{'name': n, 'rank': c % world, 'frames': int(m.shape[0]), 'size': [int(m.shape[1]), int(m.shape[2])],
                                 'water_fraction': round(float((m > 0).float().mean()), 6),
                                 'final_bank_entries': [int(v) for v in z[-1]] if z.numel() else None}
                                for c, (n, m, z) in enumerate(zip(names, masks, sizes))]}
        print(json.dumps(summary), flush=True)
        if args.save_gathered:
            import numpy as np
            np.savez_compressed(args.save_gathered, **{n: m.cpu().numpy() for n, m in zip(names, masks)})
    if vdist.active(world):
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return names, masks


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = get_args(argv)
    if os.environ.get('WORLD_SIZE') is None and args.gpus > 1:
        from . import dist as vdist
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # (the import shim vfloodnet_amd.py lives there)
        pp = root + (os.pathsep + os.environ['PYTHONPATH'] if os.environ.get('PYTHONPATH') else '')
        return vdist.spawn_ranks([sys.executable, '-m', 'vfloodnet_amd.batch_video_seg'] + argv, args.gpus,
                                 extra_env={'PYTHONPATH': pp})
    if os.environ.get('WORLD_SIZE') is not None and int(os.environ['WORLD_SIZE']) != args.gpus and args.gpus > 1:
        raise SystemExit(f'batch_video_seg: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ["WORLD_SIZE"]} ranks')
    run(args)
    return 0


if __name__ == '__main__':
    sys.exit(main() or 0)
