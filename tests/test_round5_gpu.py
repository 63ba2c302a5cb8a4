"""Round-5 GPU tests: the engine after an optimizer step + ``model.eval()``, the RCCL branch of the
collectives with one rank, and the bench line's round-5 fields."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 20200212
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eval_after_a_torch_optimizer_step_runs_on_current_weights(gpu):
    """torch's own AdamW steps the nn.Parameters in place; the LAST ``optimizer.step()`` is followed by no
    training-mode call, only by ``model.eval()`` and inference.  The engine's packed filters, Winograd banks and folded BatchNorm
    constants must follow: the eval-mode outputs equal those of a model freshly built from the stepped state dict, bit for bit
    (before the fix they were those of the weights one step back)."""
    from tools import synth
    from vfloodnet_amd import AFB_URR, FeatureBank
    sd = synth.make_state_dict(SEED)
    model = AFB_URR(gpu, update_bank=False).to(gpu)
    model.load_state_dict(sd, strict=True)
    frames, m0 = synth.clip(3, 2, 96, 160)
    frames = frames.to(gpu)
    oh = synth.onehot(m0).unsqueeze(0).to(gpu)

    def infer(m):
        k, v = m.memorize(frames[0:1], oh)
        fb = FeatureBank(2, 250000, gpu)
        fb.init_bank(k, v)
        score, _ = m.segment(frames[1:2], fb)
        return torch.stack(list(k)).clone(), score.clone()

    model.eval()
    k_before, s_before = infer(model)                       # (the engine exists and holds the initial weights)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    gen = torch.Generator(device=gpu).manual_seed(1)
    for p in model.parameters():
        p.grad = torch.randn(p.shape, device=gpu, generator=gen) * p.detach().abs().mean()
    opt.step()
    # path 1: eval() right after the step
    model.eval()
    k_eval, s_eval = infer(model)
    # path 2: a step taken while ALREADY in eval mode (no train() / eval() call follows): the sentinels catch it
    for p in model.parameters():
        p.grad = torch.randn(p.shape, device=gpu, generator=gen) * p.detach().abs().mean()
    opt.step()
    k_eval2, s_eval2 = infer(model)

    def fresh_from(m):
        f = AFB_URR(gpu, update_bank=False).to(gpu).eval()
        f.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()}, strict=True)
        return f
    k_f2, s_f2 = infer(fresh_from(model))
    assert torch.equal(k_eval2, k_f2) and torch.equal(s_eval2, s_f2)
    assert not torch.equal(s_eval2, s_eval) and not torch.equal(s_eval, s_before)
    # the first step alone: rebuild its weights by undoing nothing -- compare against a model that took the same first step
    model2 = AFB_URR(gpu, update_bank=False).to(gpu)
    model2.load_state_dict(sd, strict=True)
    model2.train()
    opt2 = torch.optim.AdamW(model2.parameters(), lr=1e-3)
    gen2 = torch.Generator(device=gpu).manual_seed(1)
    for p in model2.parameters():
        p.grad = torch.randn(p.shape, device=gpu, generator=gen2) * p.detach().abs().mean()
    opt2.step()
    k_f1, s_f1 = infer(fresh_from(model2))
    assert torch.equal(k_eval, k_f1) and torch.equal(s_eval, s_f1)


_NCCL_CHILD = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
import vfloodnet_amd
from vfloodnet_amd import dist as vdist
rank, local_rank, world = vdist.init()                      # VFN_FORCE_DIST=1: a one-rank RCCL group
assert world == 1 and dist.is_initialized() and dist.get_backend() == 'nccl', (world, dist.is_initialized())
assert vdist.active(world)
dev = torch.device('cuda', 0)
# gather_masks: all_gather_into_tensor on device tensors (dist.py, the nccl branch)
lab = (torch.arange(3 * 4 * 6 * 8, device=dev) % 251).to(torch.uint8).view(3, 4, 6, 8)
out = vdist.gather_masks(lab, 3, rank, world)
assert out.is_cuda and torch.equal(out, lab)
# run_sharded: broadcast of the shape + the gather
got = vdist.run_sharded(lambda c: lab[c], 3, rank, world, dev)
assert torch.equal(got, lab)
# gather_ragged: shapes travel as device tensors under nccl, blocks padded to the largest clip
clips = [lab[0, :2, :5, :7].contiguous(), lab[1], lab[2, :3, :6, :4].contiguous()]
rag = vdist.gather_ragged(clips, 3, rank, world, dev)
assert all(torch.equal(a, b) for a, b in zip(rag, clips))
# the two small collectives of bench.py (max over ranks, per-rank values) on device tensors
v = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(v, op=dist.ReduceOp.MAX)
parts = [torch.empty_like(v)]
dist.all_gather(parts, v)
assert float(parts[0].item()) == 1.25
dist.barrier()
dist.destroy_process_group()
print(json.dumps({'ok': True, 'backend': 'nccl'}))
'''


def _child_env():
    env = dict(os.environ, VFN_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'VFN_DIST_BACKEND', 'VFN_SINGLE_DEVICE'):
        env.pop(k, None)
    return env


def test_rccl_branch_of_the_collectives_with_one_rank(gpu):
    """``backend='nccl'`` (RCCL) with WORLD_SIZE=1 in a child process: ``dist.gather_masks`` / ``run_sharded`` / ``gather_ragged``
    and bench.py's two small collectives run through their device-tensor branches (``all_gather_into_tensor`` had never executed
    in this repo -- the multi-rank tests use gloo)."""
    r = subprocess.run([sys.executable, '-c', _NCCL_CHILD, ROOT], env=_child_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    assert json.loads(line) == {'ok': True, 'backend': 'nccl'}


def test_bench_line_through_the_rccl_branch_and_with_five_apply_samples(gpu):
    """bench.py as one forced RCCL rank: the line's ``distributed.backend`` is 'nccl', the mask all-gather ran inside the bracket
    (its time and byte count are reported), and the dominant kernel's roofline rests on >= 3 bracketed launches with the driver's
    --steps 20 (one fully sampled frame + the apply kernel alone on every 5th frame)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '2', '--min-warm-s', '0',
                        '--min-timed-s', '0', '--no-cpu-baseline'], env=_child_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    d = out['distributed']
    assert d['backend'] == 'nccl' and d['world_size'] == 1 and d['forced_single_rank_group'] is True
    assert d['all_gather_bytes_per_rank'] == 20 * 480 * 854 and d['all_gather_ms_max'] > 0
    assert out['n_gpus'] == 1 and out['steps'] == 20
    roof = out['roofline']
    assert roof['kernel'] == 'memread_apply_ss_kernel' and roof['launches_timed'] >= 3, roof
    assert 0.5 < roof['frac'] < 1.0


def test_graph_replay_of_the_launch_lists_is_bit_identical(gpu):
    """The fixed-shape launch lists (memory encoder, query side in two halves, decoder per slot) are captured into HIP graphs on their third
    run and replayed with one host call each (engine.GraphCache).  Replay must be the same kernels on the same buffers: a clip run with
    the graphs equals the launch-by-launch run (Engine.eager) bit for bit -- labels, logits of the last frame, bank sizes -- and the
    graphs must really have been captured."""
    from tools import synth
    from vfloodnet_amd import AFB_URR, engine as E
    from vfloodnet_amd.video_seg import ClipRunner
    assert E._GRAPHS
    sd = synth.make_state_dict(SEED)
    frames, m0 = synth.clip(4, 14, 96, 160)
    frames = frames.to(gpu)
    onehot = synth.onehot(m0).unsqueeze(0).to(gpu)
    out = {}
    for mode in ('graphs', 'eager'):
        model = AFB_URR(gpu, update_bank=True).to(gpu).eval()
        model.load_state_dict(sd, strict=True)
        model.engine().eager = (mode == 'eager')
        runner = ClipRunner(model, 2, 250000)
        runner.start(frames[0:1], onehot)
        labs = []
        for t in range(1, frames.shape[0]):
            lab = runner.step(frames[t:t + 1], next_frames=[frames[u:u + 1] for u in range(t + 1, min(frames.shape[0], t + 4))])
            labs.append(torch.from_numpy(lab.numpy().copy()))
        plan = model.engine().last_query[0]                 # (the plan of the network-resolution frames: the loop resizes to a 480-pixel short edge)
        out[mode] = (torch.stack(labs), plan.score.clone(), runner.bank_sizes(), sum(len(pl.graphs.graphs) for pl in model.engine().plans.values()))
    assert out['eager'][3] == 0 and out['graphs'][3] >= 4, (out['eager'][3], out['graphs'][3])
    assert torch.equal(out['graphs'][0], out['eager'][0])
    assert torch.equal(out['graphs'][1], out['eager'][1])
    assert out['graphs'][2] == out['eager'][2]
