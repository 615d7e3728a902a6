"""Multi-GPU row of the hot path, exercised with 2 CPU processes over gloo: samples are sharded over ranks, weights
replicated, and ONE all-reduce over the flat gradient buffer averages the gradients (hipvsr/dp.py).

Property checked: because every loss term is a mean over the batch and samples are independent (SURVEY.md 8e,
quirk Q8), the average of the per-rank gradients on equal shards equals the single-process gradient of the whole
batch.  The engine runs over the torch test double here (no GPU in this container); on the GPU box the same
code path runs with backend "nccl" (= RCCL) - bench.py --gpus N.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Holder(torch.nn.Module):
    """Owns the parameters under the reference's names; gradients are produced by the engine."""

    def __init__(self, sd):
        super().__init__()
        self.names = list(sd.keys())
        self.params = torch.nn.ParameterList([torch.nn.Parameter(v.clone()) for v in sd.values()])
        self._flat_grad = None


def _grads_for(sd, kwargs, inputs, targets, pos, flat=True):
    from hipvsr import lib as L
    from hipvsr.engine import RefineNetEngine
    from hipvsr.spec import NetConfig
    from torch_ops import TorchOps
    cfg = NetConfig(**kwargs)
    ops = TorchOps('cpu')
    eng = RefineNetEngine(cfg, ops)
    O, ctx = eng.forward(sd, inputs, pos, need_grad=True)
    S, T, G = cfg.num_stages, len(targets), 3 * cfg.num_stages
    y = torch.stack(targets, 0).permute(0, 1, 3, 4, 2).contiguous()
    gscale = torch.tensor([np.power(0.5, S - 1 - g // 3) / T for g in range(G) for _ in range(T)], dtype=torch.float32)
    _, dO = ops.loss(O.reshape(G * T, -1), y.reshape(T, -1), G, T, L.LOSS_L1, 0.0, gscale, want_grad=True)
    n = sum(v.numel() for v in sd.values())
    buf = torch.zeros(n) if flat else None
    return eng.backward(sd, ctx, dO.reshape(O.shape), flat=buf), buf


def _worker(rank, world, port, case_path, out_path):
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401  (puts the package on sys.path)
    from hipvsr import dp
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    c = torch.load(case_path, weights_only=False)
    sd = {k: (v.clone() if rank == 0 else torch.zeros_like(v)) for k, v in c['state_dict'].items()}
    net = _Holder(sd)
    dp.broadcast_parameters(net)                               # rank 1 starts from zeros and must receive rank 0's weights
    sd_r = {k: p.detach() for k, p in zip(net.names, net.params)}
    n = c['inputs'][0].shape[0]
    per = n // world
    sl = slice(rank * per, (rank + 1) * per)
    grads, flat = _grads_for(sd_r, c['kwargs'], [x[sl] for x in c['inputs']], [t[sl] for t in c['targets']], c['pos_codes'][sl])
    for p, k in zip(net.params, net.names):
        p.grad = grads[k]
    net._flat_grad = flat
    nbytes = dp.allreduce_gradients(net)
    assert nbytes == flat.numel() * 4                          # one collective over the whole buffer, in place
    for p, k in zip(net.params, net.names):
        if p.grad is not None:
            assert p.grad.data_ptr() == grads[k].data_ptr()
    # epoch log: per-rank sums and counts become the whole split's (ReduceLROnPlateau / Monitor / early stop then agree)
    log, count = dp.allreduce_log({'Loss': 3.0 * (rank + 1), 'PSNR': 10.0 + rank}, 4 + rank, torch.device('cpu'))
    assert log == {'Loss': 9.0, 'PSNR': 21.0} and count == 9.0 and dp.rank() == rank
    # HIP-graph replay of the step beside more than one rank has never run on hardware: refused unless RNH_GRAPH_DP=1 opts in -
    # before anything touches a device
    from src.runner.trainers import AcdcVSRRefineNetTrainer
    tr = object.__new__(AcdcVSRRefineNetTrainer)
    tr.graph, tr._graphed = True, None
    os.environ.pop('RNH_GRAPH_DP', None)
    try:
        tr.train_step(None, None, None)
        raise AssertionError('graph=True with two ranks was not refused')
    except RuntimeError as e:
        assert 'RNH_GRAPH_DP=1' in str(e), e
    if rank == 0:
        torch.save({k: (p.grad.clone() if p.grad is not None else None) for p, k in zip(net.params, net.names)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradient_allreduce_equals_full_batch(golden_dir, tmp_path):
    c = torch.load(os.path.join(golden_dir, 'g1_tiny.pt'), weights_only=False)['x2_pos1_mem1']
    case_path, out_path = str(tmp_path / 'case.pt'), str(tmp_path / 'out.pt')
    torch.save({k: c[k] for k in ('kwargs', 'state_dict', 'inputs', 'targets', 'pos_codes')}, case_path)
    mp.spawn(_worker, args=(2, _free_port(), case_path, out_path), nprocs=2, join=True)
    got = torch.load(out_path, weights_only=False)
    for k, gref in c['grads'].items():                         # reference gradient of the full batch of 2
        if gref is None:
            assert got[k] is None
            continue
        scale = float(gref.abs().max()) + 1e-12
        assert float((got[k] - gref).abs().max()) <= 2e-4 * scale + 1e-7, k


def test_allreduce_rebuilds_buffer_when_views_were_replaced():
    from hipvsr import dp
    assert dp.world() == 1
    net = torch.nn.Linear(3, 2)
    for p in net.parameters():
        p.grad = torch.ones_like(p)
    assert dp.allreduce_gradients(net) == 0                    # single process: nothing to do


def test_loader_order_is_a_function_of_seed_and_epoch():
    """GpuCineLoader (host logic, no GPU): ranks share one permutation per epoch and take disjoint shards of it; set_epoch
    makes order and augmentation draws a function of (seed, epoch) - a resumed run replays nothing and skips nothing;
    the default seed comes from Python's `random`, i.e. from main.random_seed."""
    import random
    import types
    from hipvsr.cine_cache import GpuCineLoader
    cache = types.SimpleNamespace(table=[None] * 3, train_items=lambda: [(c, t) for c in range(3) for t in range(10)])
    mk = lambda **kw: GpuCineLoader(cache, type='train', batch_size=4, shuffle=True, **kw)       # noqa: E731
    random.seed('vsr')
    a = mk(rank=0, world_size=2)
    random.seed('vsr')
    b = mk(rank=1, world_size=2)
    assert a.seed == b.seed
    random.seed('other')
    assert mk(rank=0, world_size=2).seed != a.seed
    for ep in (1, 2):
        a.set_epoch(ep)
        b.set_epoch(ep)
        oa, ob = a._order(), b._order()
        assert len(oa) == len(ob) == 15 and not set(oa) & set(ob) and set(oa) | set(ob) == set(range(30))
    a.set_epoch(1)
    o1, d1 = a._order(), [a.rng.random() for _ in range(4)]
    a.set_epoch(2)
    assert a._order() != o1
    a.set_epoch(1)                                            # "resume": epoch 1 again gives epoch 1's order and draws
    assert a._order() == o1 and [a.rng.random() for _ in range(4)] == d1
    assert [b.rng.random() for _ in range(4)] != d1           # ranks draw differently
