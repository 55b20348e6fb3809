"""bench.py launches its own ranks when called as `python bench.py --gpus N` (the driver's command line): CPU-only tests of the
launcher -- the re-exec under torch.distributed.run (gloo, no GPU touched: TFNAS_BENCH_DRY) and the one-JSON-line error when
fewer than N devices are visible.  Reference counterpart: the nn.DataParallel wrap of train_search.py:95,158."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ)
    e.update(env)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=e, capture_output=True, text=True,
                          timeout=300)


def test_too_few_devices_is_one_json_line_and_nonzero_exit():
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'], TFNAS_FAKE_DEVICES='1')
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout + r.stderr
    j = json.loads(lines[0])
    assert 'error' in j and j['n_gpus'] == 2 and j['devices_visible'] == 1


import pytest


@pytest.mark.parametrize('n', [2, 8])
def test_self_launch_brings_up_n_ranks(n):
    """N = 8: what the driver's scaling run launches on an 8-GPU node (dry ranks: rendezvous + one all-reduce + one all-gather)."""
    r = _run(['--gpus', str(n), '--steps', '3', '--warmup', '1'], TFNAS_FAKE_DEVICES=str(n), TFNAS_BENCH_DRY='1', OMP_NUM_THREADS='1')
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j['dry_run'] and j['n_gpus'] == n and j['ranks_seen'] == n and j['rank_ids'] == list(range(n))
    assert j['argv'] == ['--gpus', str(n), '--steps', '3', '--warmup', '1']


def test_dist_object_of_the_bench_line_is_self_explanatory():
    """The N > 1 line carries per-rank step times (list, min / max, spread) and the gradient exchange per iteration pair."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    for key in ('w_step_ms_per_rank', 'a_step_ms_per_rank', 'w_step_ms_min_max', 'a_step_ms_min_max', 'allreduce_bytes_per_pair',
                'allreduce_calls_per_pair', 'replicas_bit_identical', 'rccl_ranks'):
        assert key in src, key
