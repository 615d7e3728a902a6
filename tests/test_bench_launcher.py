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
    a = bench.parse_args(['--config', 'yaml'])          # the reference YAML's own training shape (exp1_x4.yaml:20-33)
    assert (a.config, a.batch, a.frames, a.size, a.scale) == ('yaml', 16, 7, 32, 4)


def test_rank_cpu_map_splits_each_numa_node_among_its_ranks():
    """launch_ranks pins every rank to a share of the CPUs of its GPU's NUMA node (rocm-smi --showtoponuma, /sys/devices/system/node)."""
    sys.path.insert(0, ROOT)
    import bench
    smi = '\n'.join(f'GPU[{g}]\t\t: (Topology) Numa Node: {g // 4}\nGPU[{g}]\t\t: (Topology) Numa Affinity: {g // 4}' for g in range(8))
    nodes = bench.gpu_numa_nodes(8, smi)
    assert nodes == {g: g // 4 for g in range(8)}
    assert bench.gpu_numa_nodes(2, smi) == {0: 0, 1: 0}
    cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    m = bench.rank_cpu_map(8, nodes, cpus, allowed=range(256))
    assert sorted(m) == list(range(8)) and all(len(v) == 32 for v in m.values())
    assert set(m[0]) | set(m[1]) | set(m[2]) | set(m[3]) == set(cpus[0]) and not set(m[3]) & set(m[4])
    assert len({c for v in m.values() for c in v}) == 256                   # nobody shares a core
    assert bench.rank_cpu_map(8, {0: 0}, cpus, allowed=range(256)) == {}    # a GPU of unknown node: no pinning at all
    assert bench.rank_cpu_map(2, {0: 0, 1: 0}, {0: [0]}, allowed=[0]) == {}  # fewer CPUs than ranks: no pinning
    assert bench.rank_cpu_map(2, {0: 0, 1: 1}, cpus, allowed=range(0, 100)) == {0: list(range(0, 64)), 1: list(range(64, 100))}
    assert bench.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]


def test_apply_rank_affinity_takes_the_launchers_cpus(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    import torch
    mine, threads = sorted(os.sched_getaffinity(0)), torch.get_num_threads()
    try:
        monkeypatch.setenv('BENCH_RANK_CPUS', json.dumps({'0': mine[:1], '1': mine[-1:]}))
        assert bench.apply_rank_affinity(1) == mine[-1:] and sorted(os.sched_getaffinity(0)) == mine[-1:]
        monkeypatch.delenv('BENCH_RANK_CPUS')
        assert bench.apply_rank_affinity(0) is None
    finally:
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(threads)          # (the CPU oracle is bit-exact against the goldens at the thread count they were made with)


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


def test_eight_ranks_over_gloo_with_a_slow_rank_and_with_a_failing_rank():
    """The N = 8 launch the driver makes on a whole node, rehearsed on CPU: eight ranks rendezvous over gloo; the line reports the MAX over
    ranks (a slow rank sets the step time and shows in the per-rank spread); a rank that dies fails the launcher and no line is printed."""
    r = _run(['--gpus', '8', '--dry-run', '--steps', '3', '--warmup', '1', '--dry-run-slow-rank', '5'], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _line(r.stdout)
    assert out['n_gpus'] == 8 and out['config']['rccl_world'] == 8 and out['config']['global_batch'] == 64
    lo, hi = out['config']['rank_ms_per_step_wall_min_max']
    assert hi >= 50.0 > lo and out['ms_per_step'] >= hi                      # rank 5 sleeps 50 ms per step more than the others
    r = _run(['--gpus', '8', '--dry-run', '--dry-run-fail-rank', '6', '--steps', '1', '--warmup', '0'], timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = _run(['--gpus', '4', '--dry-run'], env={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr
    r = _run(['--gpus', '1', '--dry-run', '--steps', '1', '--warmup', '0'])            # one rank, in process
    assert r.returncode == 0 and _line(r.stdout)['n_gpus'] == 1


@pytest.mark.gpu
def test_one_rank_under_torchrun_goes_through_rccl():
    sys.path.insert(0, ROOT)
    import bench
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1', '--master-port', str(bench.free_port()),
           BENCH, '--gpus', '1', '--steps', '1', '--warmup', '1', '--no-secondary', '--no-cpu-baseline']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert r.returncode == 0, r.stderr[-3000:]
    out = _line(r.stdout)
    assert out['n_gpus'] == 1 and out['config']['rccl_world'] == 1 and out['value'] > 0 and out['config']['gates'] == 'stored'
    assert out['config']['rank_ms_per_step_median_min_max'][0] > 0 and 'grad_allreduce_ms_min_max' in out['config']


def test_committed_pmc_records_belong_to_the_committed_kernel_sources():
    """bench.py quotes `roofline.traffic` out of profiles/*_hbm_bytes.json only while the record's sha256 is that of the kernel source it was measured
    on - a one-line comment edit after the PMC pass silently turns the traffic of the driver's line into null (it happened in round 6).  The records
    bench.py reads at config 2 (the fp32 headline's F(4x4) cell, the bf16 secondary's cell) must match the tree."""
    sys.path.insert(0, ROOT)
    import bench
    for record, source in (('lstm44_kernel_hbm_bytes.json', 'conv_wino44.hip'), ('lstm_bf16_kernel_hbm_bytes.json', 'conv_bf16.hip')):
        rec = json.load(open(os.path.join(ROOT, 'profiles', record)))
        assert rec['kernel_source_sha256'] == bench.kernel_source_sha256(source), f'{record} was measured on another revision of {source}: re-run tools/prof_pmc_*.sh'
        traffic, src = bench.quoted_traffic(record, source, (8, 128, 128))
        assert traffic and traffic > rec['algorithmic_bytes_per_launch'] * 0.9 and src['command']
