"""bench.py's launch contract: `--gpus N` must mean N ranks.

CPU: the launcher logic with the GPU step stubbed (`--dry-run`: the same rendezvous / barrier / max-over-ranks protocol over gloo) -
argument -> command line and environment of the children, rank 0's line relayed, exit code propagated, WORLD_SIZE != --gpus refused.
GPU: one rank under torch.distributed.run through RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_launcher_command_is_the_drivers_command():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(8, ['--gpus', '8', '--steps', '5'], 29511)
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=8' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29511'
    assert cmd[-5:] == [BENCH, '--gpus', '8', '--steps', '5']
    a = bench.parse_args(['--config', '4'])
    assert (a.batch, a.frames, a.size, a.scale) == (16, 5, 256, 2)
    a = bench.parse_args(['--config', '5', '--batch', '4'])
    assert (a.batch, a.frames, a.size, a.scale) == (4, 11, 96, 4)
    a = bench.parse_args([])
    assert (a.config, a.batch, a.frames, a.size, a.scale, a.gpus) == (2, 8, 7, 128, 4, 1)


def test_gpus_2_without_world_size_starts_two_ranks():
    r = _run(['--gpus', '2', '--dry-run', '--steps', '2', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _line(r.stdout)
    assert out['n_gpus'] == 2 and out['config']['rccl_world'] == 2 and out['config']['global_batch'] == 16
    assert out['config']['master_addr'] == '127.0.0.1' and out['config']['local_rank_env'] == '0'


def test_a_failing_rank_fails_the_launcher():
    r = _run(['--gpus', '2', '--dry-run', '--dry-run-fail-rank', '1', '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = _run(['--gpus', '4', '--dry-run'], env={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr
    r = _run(['--gpus', '1', '--dry-run', '--steps', '1', '--warmup', '0'])            # one rank, in process
    assert r.returncode == 0 and _line(r.stdout)['n_gpus'] == 1


@pytest.mark.gpu
def test_one_rank_under_torchrun_goes_through_rccl():
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1', '--master-port', '29533',
           BENCH, '--gpus', '1', '--steps', '1', '--warmup', '1', '--no-secondary', '--no-cpu-baseline']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert r.returncode == 0, r.stderr[-3000:]
    out = _line(r.stdout)
    assert out['n_gpus'] == 1 and out['config']['rccl_world'] == 1 and out['value'] > 0 and out['config']['gates'] == 'stored'
