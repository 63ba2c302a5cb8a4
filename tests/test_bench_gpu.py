"""bench.py end to end on the GPU box: the driver's command line, the self-launched N > 1 path (two ranks sharing the
one device over gloo), and the C3 / C5 lines with their parity object."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return lines[0]


def test_driver_command_is_the_c2_clip(gpu):
    d = _run(['--gpus', '1', '--steps', '20', '--warmup', '5', '--cpu-frames', '3'])
    assert d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5 and d['unit'] == 'frames/s' and d['dtype'] == 'f32'
    assert d['config']['workload'].startswith('C2: 100-frame 480x854')
    # the 20 timed frames carry the full clip's mean bank (within 3 %) and the whole clip was run
    assert abs(d['config']['mean_bank_entries_per_object'] / d['config']['full_clip_mean_bank_entries_per_object'] - 1) < 0.03
    assert 50000 < d['config']['mean_bank_entries_per_object'] < 62000
    assert d['parity']['full_clip_frames'] == 99 and d['parity']['full_clip_miou_min'] >= 0.99
    assert d['parity']['miou_vs_oracle'] >= 0.99 and d['parity']['bank_sizes_equal']
    assert abs(d['value'] / d['full_clip_fps'] - 1) < 0.1
    r = d['roofline']
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 0.2 < r['frac'] < 1.0
    assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['value'] > 0


def test_self_launch_two_ranks_on_one_device(gpu):
    d = _run(['--gpus', '2', '--steps', '6', '--warmup', '1', '--min-warm-s', '0', '--no-cpu-baseline'],
             {'VFN_DIST_BACKEND': 'gloo', 'VFN_SINGLE_DEVICE': '1'})
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['value'] > 0
    # what a SCALE record needs to show that N ranks met over the collective backend, and where the time went
    x = d['distributed']
    assert x['backend'] == 'gloo' and x['world_size'] == 2                   # ('nccl' = RCCL on a multi-GPU node)
    assert len(x['frames_per_s_per_rank']) == 2 and x['frames_per_s_per_rank_min'] <= x['frames_per_s_per_rank_max']
    assert len(x['all_gather_ms_per_rank']) == 2 and x['all_gather_ms_max'] > 0 and x['all_gather_bytes_per_rank'] == 6 * 480 * 854
    # the whole-job value is bounded by the per-rank rates (the bracket also holds the gather and two barriers)
    assert d['value'] <= sum(x['frames_per_s_per_rank']) * 1.001
    assert d['hbm']['peak_bytes_allocated'] > d['hbm']['bank_slab_bytes_f32'] > 0


@pytest.mark.parametrize('workload,steps', [('C3', 12), ('C5', 12)])
def test_reduced_precision_lines_carry_parity(gpu, workload, steps):
    d = _run(['--workload', workload, '--precision', 'bf16x3', '--steps', str(steps), '--warmup', '1', '--min-warm-s', '0',
              '--cpu-frames', '5'])
    assert d['dtype'].startswith('bf16x3') and d['parity'] is not None
    assert d['parity']['miou_vs_oracle'] >= 0.99 and d['parity']['bank_sizes_equal']
    assert d['config']['network_resolution'].startswith('480x853')
    assert d['distributed']['world_size'] == 1 and d['distributed']['backend'] is None
    assert d['hbm']['peak_bytes_allocated'] >= d['hbm']['resident_frames_bytes'] > 0


def test_c5_line_carries_bank_curve_and_peak_hbm(gpu):
    """frames/s against the bank size and the peak HBM bytes are part of the line."""
    d = _run(['--workload', 'C5', '--precision', 'bf16x3', '--steps', '60', '--warmup', '1', '--min-warm-s', '0', '--no-cpu-baseline'])
    pts = d['bank_curve']['points']
    assert d['bank_curve']['block_frames'] == 10 and len(pts) == 6
    sizes = [p_['mean_bank_entries_per_object'] for p_ in pts]
    assert sizes == sorted(sizes) and sizes[-1] > sizes[0] > 1620          # the bank only grows in C5
    assert all(p_['frames_per_s'] > 0 for p_ in pts)
    assert d['hbm']['final_bank_entries_per_object'][0] <= d['hbm']['bank_capacity_entries_per_object']
