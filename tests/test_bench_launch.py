"""``python bench.py --gpus N`` must bring N ranks up by itself when no launcher did (WORLD_SIZE unset) and must not
silently ignore ``--gpus`` under a launcher.  ``--launch-check`` runs the bring-up + the run's one collective on host
tensors (gloo), so the launch logic is exercised on a CPU-only machine; the GPU counterpart is
tests/test_bench_gpu.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'LOCAL_WORLD_SIZE')}
    env['VFN_DIST_BACKEND'] = 'gloo'
    env['VFN_SINGLE_DEVICE'] = '1'
    return env


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith('{')]


def test_self_launch_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-check'],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                       # rank 0 only
    assert lines[0]['n_gpus'] == 2 and lines[0]['gather_ok'] is True


def test_single_rank_default():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--launch-check'],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout)[0]['n_gpus'] == 1


def test_gpus_mismatch_under_a_launcher_is_an_error():
    env = _clean_env()
    env.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-check'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_window_selection():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.pick_window(99, 99) == 1
    assert bench.pick_window(20, 99) == 40                 # centred without the golden sizes
    import numpy as np
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'c2_480x854_100.npz'))
    sizes = g['bank_sizes'].tolist()
    s = bench.pick_window(20, 99, sizes)
    full = sum(sum(x) for x in sizes) / 99
    win = sum(sum(x) for x in sizes[s - 1:s + 19]) / 20
    assert abs(win - full) / full < 0.01, (s, win, full)
