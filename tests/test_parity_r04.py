"""Round-4 parity (VERDICT r03 "next" items 1 and 2), on the GPU box, through the C ABI.

* BASELINE config 4 at its STATED batch: x2, N = 16, T = 5, 256x256 -> 512x512 - a full training step (forward + loss + backward)
  in fp32 and in the bf16-storage path on one MI355X.  N = 1 is held against the CPU oracle (== reference
  src/model/nets/refine_net.py:61-135 + trainer :83-94); the batch is 16 copies of that sample, every one of which must come out
  bit-identical to it (samples are independent, quirk Q8) with the gradients (a batch mean) unchanged.  What makes it fit is the
  engine's activation-memory plan (hipvsr/engine.py FrameStore: only the states around the supervised frames survive the stage's
  forward; gate recomputation where the stored gates would not fit) - the same test pins that plan: stored and recomputed gates
  give bit-identical gradients.
* Training trajectories: K optimizer steps of the reference's training loop (trainer :17-62 with Adam, exp1_x4.yaml:56-60) on the
  HIP path against the CPU oracle + oracle Adam, and the bf16-storage path against the fp32 path at trained scale.
"""
import functools
import os

import pytest
import torch

from oracle import refinenet_oracle as orc
from oracle import step_tail_oracle as sto

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def _net(cfg, sd, dtype):
    from src.model.nets import RefineNet
    net = RefineNet(**dict(cfg))
    net.load_state_dict(sd)
    return net.to(_dev()).set_compute_dtype(dtype)


def _trainer(net, loss_fn=None):
    from src.model.metrics import PSNR
    from src.runner.trainers import AcdcVSRRefineNetTrainer
    from src.utils import denormalize
    tr = object.__new__(AcdcVSRRefineNetTrainer)
    tr.net, tr.loss_fns, tr.metric_fns = net, [loss_fn or torch.nn.L1Loss()], [PSNR().to(_dev())]
    tr._denormalize = functools.partial(denormalize, dataset='acdc')
    return tr


def _grad_close(mine, ref, name, atol=1e-5, rtol=1e-3, l2=1e-3):
    a, b = mine.detach().cpu().double(), ref.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    d = (a - b).abs()
    over = d - (atol + rtol * b.abs())
    i = int(over.argmax())
    assert float(over.flatten()[i]) <= 0, (name, 'element', i, float(a.flatten()[i]), float(b.flatten()[i]), 'max|g|', float(b.abs().max()))
    assert float(d.norm()) <= l2 * float(b.norm()) + 1e-12, (name, 'L2', float(d.norm()), float(b.norm()))


def _step(net, inputs, targets, pos):
    dev = _dev()
    tr = _trainer(net)
    net.train()
    outs = net([x.to(dev) for x in inputs], pos.to(dev))
    loss = tr._compute_losses(outs, [t.to(dev) for t in targets])[0]
    net.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
    return outs, loss


def _grads(net):
    return {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in net.named_parameters()}


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 4 at N = 16
# ---------------------------------------------------------------------------------------------------------------------
def test_config4_full_batch_training_step_fp32_and_bf16():
    cfg = orc.exp1_x4_config(upscale_factor=2)
    t, size, nfull = 5, 256, 16
    sd = orc.init_state_dict(cfg, seed=51)
    inputs, targets, pos = orc.synthetic_batch(cfg, 1, t, size, size, seed=52)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref_out, ref_loss, ref_grads = orc.step(sd, cfg, [x.clone() for x in inputs], targets, pos)
    rep = lambda x, m: torch.cat([x] * m, 0)                               # noqa: E731
    dev = _dev()
    total = torch.cuda.get_device_properties(dev).total_memory
    report = []

    # ---- fp32: N = 1 against the oracle
    net = _net(cfg, sd, 'f32')
    outs1, loss1 = _step(net, inputs, targets, pos)
    for go, gr in zip(outs1, ref_out):
        for a, b in zip(go, gr):
            torch.testing.assert_close(a.detach().cpu(), b, atol=1e-4, rtol=1e-4)
    assert abs(float(loss1.detach()) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss)), (float(loss1), float(ref_loss))
    for k, p in net.named_parameters():
        if ref_grads[k] is None:
            assert p.grad is None
        else:
            _grad_close(p.grad, ref_grads[k], k)
    o1 = [[o.detach().clone() for o in grp] for grp in outs1]
    del outs1, loss1

    # ---- fp32: stored gates == recomputed gates, bit for bit (N = 4: both plans fit)
    res = {}
    for mode in ('store', 'recompute'):
        net.set_gate_memory(mode)
        assert net._engine().recompute_gates(4, size, size, t + 12) is (mode == 'recompute')
        outs, loss = _step(net, [rep(x, 4) for x in inputs], [rep(y, 4) for y in targets], rep(pos, 4))
        res[mode] = (float(loss.detach()), _grads(net))
        for ga, gb in zip(outs, o1):
            for a, b in zip(ga, gb):
                for q in range(4):
                    assert torch.equal(a[q:q + 1], b), ('fp32 N=4', mode, 'sample', q)
        del outs, loss
    assert res['store'][0] == res['recompute'][0]
    for k, g in res['store'][1].items():
        if g is not None:
            assert torch.equal(g, res['recompute'][1][k]), ('stored vs recomputed gates', k)
            _grad_close(g, ref_grads[k], k)
    del res
    net.zero_grad(set_to_none=True)
    torch.cuda.empty_cache()

    # ---- fp32: the full batch of 16 ('auto' picks the plan: on a 288 GB card the stored-gates step is estimated at ~235 GB -> recompute)
    net.set_gate_memory('auto')
    eng = net._engine()
    plan_store = eng.memory_plan(nfull, size, size, t + 12, recompute=False)['peak']
    recomputes = eng.recompute_gates(nfull, size, size, t + 12)
    assert recomputes is (plan_store > eng.AUTO_FRACTION * total)
    torch.cuda.reset_peak_memory_stats(dev)
    outs, loss = _step(net, [rep(x, nfull) for x in inputs], [rep(y, nfull) for y in targets], rep(pos, nfull))
    peak = torch.cuda.max_memory_allocated(dev)
    for ga, gb in zip(outs, o1):
        for a, b in zip(ga, gb):
            for q in range(nfull):
                assert torch.equal(a[q:q + 1], b), ('fp32 N=16 sample', q)
    assert abs(float(loss.detach()) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))
    for k, p in net.named_parameters():
        if ref_grads[k] is not None:
            _grad_close(p.grad, ref_grads[k], k)
    est = eng.memory_plan(nfull, size, size, t + 12, recompute=recomputes)['peak']
    report.append(f'fp32 N={nfull}: gates {"recomputed" if recomputes else "stored"}, peak HBM {peak / 2**30:.1f} GiB (estimated {est / 2**30:.1f}; '
                  f'stored-gates estimate {plan_store / 2**30:.1f} of {total / 2**30:.0f} GiB)')
    assert peak <= 1.25 * est, (peak, est)                        # the estimate the 'auto' plan relies on is not wildly low
    del outs, loss, net, eng
    torch.cuda.empty_cache()

    # ---- bf16-storage path: N = 1 (against the oracle: loose, the PSNR tests are the criterion), then the full batch, fwd + bwd
    nb = _net(cfg, sd, 'bf16')
    b1, lossb1 = _step(nb, inputs, targets, pos)
    for a, b in zip(b1[-1], ref_out[-1]):
        assert float((a.detach().cpu() - b).norm()) <= 2e-2 * float(b.norm())
    assert abs(float(lossb1.detach()) - float(ref_loss)) <= 1e-2 * abs(float(ref_loss))
    ob1 = [[o.detach().clone() for o in grp] for grp in b1]
    gb1 = _grads(nb)
    del b1
    torch.cuda.reset_peak_memory_stats(dev)
    assert not nb._engine().recompute_gates(nfull, size, size, t + 12)      # bf16 gates of the full batch fit
    outs, loss = _step(nb, [rep(x, nfull) for x in inputs], [rep(y, nfull) for y in targets], rep(pos, nfull))
    peakb = torch.cuda.max_memory_allocated(dev)
    for ga, gb in zip(outs, ob1):
        for a, b in zip(ga, gb):
            for q in range(nfull):
                assert torch.equal(a[q:q + 1], b), ('bf16 N=16 sample', q)
    assert abs(float(loss.detach()) - float(lossb1.detach())) <= 1e-6 * abs(float(lossb1))
    for k, p in nb.named_parameters():
        if gb1[k] is not None:
            # the same per-sample arithmetic, summed over 16 copies in another order: fp32 re-association only
            _grad_close(p.grad, gb1[k], k)
            assert float((p.grad.cpu() - ref_grads[k]).norm()) <= 6e-2 * float(ref_grads[k].norm()) + 1e-7, k
    report.append(f'bf16 N={nfull}: peak HBM {peakb / 2**30:.1f} GiB')
    print('config 4 (x2, T=5, 256x256) training step at the stated batch: ' + '; '.join(report))
