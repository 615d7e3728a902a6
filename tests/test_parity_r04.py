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
def test_config4_full_batch_training_step_fp32_and_bf16(monkeypatch):
    # (the fp32 cell in F(4x4, 3x3) form whatever RNH_WINO44_MIN says: the per-sample bit-identity below compares N = 1, 4 and 16)
    monkeypatch.setenv('RNH_WINO44', 'force')
    # (at N = 16 the transformed h' live in a ring and refine conv1 runs in F(2x2) form - what bench.py --config 4 runs; the same for N = 1 and 4 here)
    monkeypatch.setenv('RNH_WINO44_REFINE', '0')
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

    # ---- fp32: the full batch of 16 ('auto' picks the plan: on a 288 GiB card the stored-gates step is estimated at ~235 GiB, over the 80 %
    # the plan allows itself -> the first stage recomputes its gates, the other two store them)
    net.set_gate_memory('auto')
    eng = net._engine()
    plan_store = eng.memory_plan(nfull, size, size, t + 12, recompute=False)['peak']
    n_rc = eng.recompute_stages(nfull, size, size, t + 12)
    assert (n_rc > 0) is (plan_store > eng.AUTO_FRACTION * total) and eng.recompute_gates(nfull, size, size, t + 12) is (n_rc > 0)
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
    est = eng.memory_plan(nfull, size, size, t + 12, recompute=n_rc)['peak']
    report.append(f'fp32 N={nfull}: gates recomputed in {n_rc} of {cfg["num_stages"]} stages, peak HBM {peak / 2**30:.1f} GiB (estimated {est / 2**30:.1f}; '
                  f'stored-gates estimate {plan_store / 2**30:.1f} of {total / 2**30:.0f} GiB)')
    assert peak <= 1.08 * est, (peak, est)                        # the estimate the 'auto' plan relies on is not low
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


# ---------------------------------------------------------------------------------------------------------------------
# training trajectories (reference trainer :17-62: forward, _compute_losses, zero_grad, backward, optimizer.step; Adam, exp1_x4.yaml:56-60)
# ---------------------------------------------------------------------------------------------------------------------
def _train_trainer(net, lr):
    from hipvsr.step_tail import FlatAdam
    tr = _trainer(net)
    tr.optimizer = FlatAdam(net.parameters(), lr=lr, weight_decay=0)
    tr.loss_weights = torch.tensor([1.0], device=_dev())
    tr.graph, tr._graphed = False, None
    return tr


def test_ten_training_steps_follow_the_oracle_trajectory():
    """The HIP fp32 path and the CPU oracle (+ oracle Adam == torch.optim.Adam, tests/test_step_tail.py) take the SAME 10 optimizer steps
    from the same initialisation on the same batches; the losses of every step and the parameters after step 10 must agree.  The
    criterion on the parameters is per tensor ||p_hip - p_oracle||_2 <= 1e-4 ||p_oracle||_2 - and, sharper, on what the 10 steps
    changed: ||(p_hip - p_0) - (p_oracle - p_0)||_2 <= 2e-2 ||p_oracle - p_0||_2 (Adam's normalised update amplifies the relative
    gradient error wherever a gradient is small)."""
    cfg = orc.Config(in_channels=1, out_channels=1, num_features=[8, 8], num_stages=3, refine_window_size=5, upscale_factor=4,
                     update_memory=True, num_updated_frames=3, positional_encoding=True)
    steps, lr = 10, 1e-3
    sd0 = orc.init_state_dict(cfg, seed=31)
    batches = [orc.structured_cine(cfg, 2, 4, 6, 5, seed=400 + (i % 3)) for i in range(steps)]
    names = list(sd0.keys())
    # oracle trajectory
    p = [sd0[k].clone() for k in names]
    m, v = [torch.zeros_like(t) for t in p], [torch.zeros_like(t) for t in p]
    ref_losses = []
    for i, (inputs, targets, pos) in enumerate(batches):
        _, loss, grads = orc.step(dict(zip(names, p)), cfg, [x.clone() for x in inputs], targets, pos)
        ref_losses.append(float(loss))
        sto.adam_step(p, [grads[k] for k in names], m, v, i + 1, lr=lr)
    # HIP trajectory
    dev = _dev()
    net = _net(cfg, sd0, 'f32').train()
    tr = _train_trainer(net, lr)
    losses = []
    for inputs, targets, pos in batches:
        _, loss, _ = tr.train_step([x.to(dev) for x in inputs], [t.to(dev) for t in targets], pos.to(dev))
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(losses, ref_losses)):
        assert abs(a - b) <= 1e-4 * abs(b), ('loss of step', i, a, b)
    worst_p = worst_d = 0.0
    for k, q, (kn, pn) in zip(names, p, net.named_parameters()):
        assert k == kn
        mine, init = pn.detach().cpu().double(), sd0[k].double()
        q = q.double()
        ep = float((mine - q).norm()) / (float(q.norm()) + 1e-30)
        moved = float((q - init).norm())
        ed = float(((mine - init) - (q - init)).norm()) / (moved + 1e-30) if moved > 0 else 0.0
        worst_p, worst_d = max(worst_p, ep), max(worst_d, ed)
        assert ep <= 1e-4, (k, 'parameters after 10 steps', ep)
        assert ed <= 2e-2, (k, 'update over 10 steps', ed, moved)
    print(f'10 Adam steps (lr {lr}): loss {losses[0]:.6f} -> {losses[-1]:.6f} (oracle {ref_losses[0]:.6f} -> {ref_losses[-1]:.6f}); worst tensor: '
          f'parameters {worst_p:.2e} relative L2, accumulated update {worst_d:.2e}')


_TRAIN = dict(steps=600, batch=16, crop=32, t=7, pool=64, lr=1e-4, check=50)


@pytest.fixture(scope='module')
def trained():
    """600 training steps at the reference YAML's training shape (exp1_x4.yaml:21-33: batch 16, 32 x 32 crops, T = 7, Adam lr 1e-4, L1) on the
    structured cine, from the same initialisation: in fp32, in fp32 from an initialisation perturbed by 1e-6 relative (the noise floor: how
    far two equally valid fp32 trajectories drift apart), and in the bf16-storage path.  Loss curves, validation PSNR against the TRUE
    high-resolution frames every 50 steps, final weights."""
    cfg = orc.exp1_x4_config()
    c = _TRAIN
    sd0 = orc.init_state_dict(cfg, seed=61)
    dev = _dev()
    pool = orc.structured_cine(cfg, c['pool'], c['t'], c['crop'], c['crop'], seed=62)
    pin, ptg, ppos = [x.to(dev) for x in pool[0]], [y.to(dev) for y in pool[1]], pool[2].to(dev)
    val = orc.structured_cine(cfg, 4, c['t'], 64, 64, seed=63)
    vin, vtg, vpos = [x.to(dev) for x in val[0]], [y.to(dev) for y in val[1]], val[2].to(dev)
    out = dict(cfg=cfg)
    nb = c['pool'] // c['batch']
    for name, dt, eps in (('f32', 'f32', 0.0), ('f32 perturbed', 'f32', 1e-6), ('bf16', 'bf16', 0.0)):
        g = torch.Generator().manual_seed(5)
        sd = {k: v * (1 + eps * torch.randn(v.shape, generator=g)) for k, v in sd0.items()}
        net = _net(cfg, sd, dt).train()
        tr = _train_trainer(net, c['lr'])
        losses, psnrs = [], []
        for i in range(c['steps']):
            sl = slice((i % nb) * c['batch'], (i % nb + 1) * c['batch'])
            _, loss, _ = tr.train_step([x[sl] for x in pin], [y[sl] for y in ptg], ppos[sl])
            losses.append(loss.detach())
            if (i + 1) % c['check'] == 0:
                net.eval()
                with torch.no_grad():
                    psnrs.append(float(tr._compute_metrics(net(vin, vpos), vtg)[0]))
                net.train()
        torch.cuda.synchronize()
        losses = [float(x) for x in losses]
        out[name] = dict(losses=losses, windows=[sum(losses[i:i + 25]) / 25 for i in range(0, c['steps'], 25)], psnr=psnrs,
                         sd={k: p.detach().cpu().clone() for k, p in net.state_dict().items()})
        del net, tr
        torch.cuda.empty_cache()
    return out


def test_bf16_training_run_tracks_the_fp32_run(trained):
    """where the bf16-storage run ends up against where the fp32 run ends up.  Training is chaotic: an fp32 run whose
    initialisation differs by 1e-6 relative leaves the fp32 run after ~150 steps and from then on sits 1-2 % away in the 25-step loss
    windows and 0.05-0.2 dB away in validation PSNR (profiles/ARCHIVE/r04_e_training_trajectories.txt; the size of that drift itself varies by
    2-3 x from one realisation to the next: a change of the summation order in one reduction kernel moved it) - so "within 1 % / 0.05 dB
    of the fp32 run" is not a property even fp32 has.  Asserted instead: (1) before the trajectories decorrelate (the first 125 steps,
    loss 6.4 -> 0.5, PSNR 25 -> 30 dB) the bf16 run follows the fp32 run to 1e-3 in every loss window and 0.02 dB in PSNR; (2) over the
    whole run it stays in the neighbourhood two fp32 runs stay in: never more than 0.5 dB / 10 % of a loss window away, on average over
    the 12 check points no further than 4 x the perturbed fp32 run (+ 0.05 dB); (3) no systematic lag: the mean signed PSNR difference
    over the last 6 check points is above -0.15 dB; (4) it trains to the same loss level (5 %)."""
    f, p, b = trained['f32'], trained['f32 perturbed'], trained['bf16']
    rel = lambda x, y: [abs(u - v) / u for u, v in zip(x, y)]                 # noqa: E731
    wp, wb = rel(f['windows'], p['windows']), rel(f['windows'], b['windows'])
    dp_, db = [v - u for u, v in zip(f['psnr'], p['psnr'])], [v - u for u, v in zip(f['psnr'], b['psnr'])]
    mean_abs = lambda v: sum(abs(x) for x in v) / len(v)                       # noqa: E731
    print(f'{_TRAIN["steps"]} steps at batch {_TRAIN["batch"]}, {_TRAIN["crop"]}x{_TRAIN["crop"]} crops: loss {f["losses"][0]:.4f} -> {f["windows"][-1]:.4f} (fp32) / '
          f'{b["windows"][-1]:.4f} (bf16), PSNR vs true HR {f["psnr"][0]:.2f} -> {f["psnr"][-1]:.2f} / {b["psnr"][-1]:.2f} dB; first 5 windows: bf16 within '
          f'{max(wb[:5]):.1e} (perturbed fp32 {max(wp[:5]):.1e}); whole run: loss windows within {max(wb):.1e} (perturbed fp32 {max(wp):.1e}), |dPSNR| max '
          f'{max(abs(x) for x in db):.3f} / mean {mean_abs(db):.3f} dB (perturbed fp32 {max(abs(x) for x in dp_):.3f} / {mean_abs(dp_):.3f}), mean signed dPSNR of '
          f'the last 6 check points {sum(db[-6:]) / 6:+.3f} ({sum(dp_[-6:]) / 6:+.3f})')
    assert f['windows'][-1] < 0.05 * f['losses'][0] and f['psnr'][-1] > 38.0           # the run did train
    assert max(wb[:5]) <= 1e-3, wb[:5]
    assert abs(db[0]) <= 0.02 and abs(db[1]) <= 0.02, db[:2]
    assert max(wb) <= 0.10 and max(abs(x) for x in db) <= 0.5, (max(wb), db)
    assert mean_abs(db) <= 4 * mean_abs(dp_) + 0.05, (db, dp_)
    assert sum(db[-6:]) / 6 >= -0.15, db
    assert abs(b['windows'][-1] - f['windows'][-1]) <= 0.05 * f['windows'][-1]


@pytest.mark.parametrize('name,n,t,size', [('config 1', 1, 3, 64), ('config 2 geometry', 2, 7, 128)])
def test_psnr_parity_with_trained_weights(trained, name, n, t, size):
    """the contract criterion at trained scale.  The weights the fp32 run ended with (saturating gates, a wider dynamic
    range than the default initialisation), BASELINE config 1 and config 2's geometry on the structured cine: PSNR of the fused group
    against the TRUE HR frames through the fp32 oracle (== reference) and through the HIP path: fp32 |delta| < 0.01 dB on every frame (the
    contract's tolerance) in fp32 AND in the bf16-storage path.  (Round 5 had to relax the bf16 bound to 0.03 dB: a one-sided shift of 0.016-0.021 dB
    at the weights the fp32 run ends with since the cells run in F(4x4, 3x3) form.  Round 6 found its source - the 8-bit WEIGHTS of the upsampler's
    forward, a fixed perturbation one or two linear maps in front of the output, profiles/r06_a_*, r06_b_* - and took it away: those layers
    contract in IEEE half, 0.002-0.005 dB.  The same criterion at three FROZEN weight sets: test_psnr_parity_at_frozen_trained_weights.)"""
    cfg, sd = trained['cfg'], trained['f32']['sd']
    inputs, targets, pos = orc.structured_cine(cfg, n, t, size, size, seed=71)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = orc.forward(orc.as_leaf_params(sd), cfg, [x.clone() for x in inputs], pos)
    ref_last = [o.detach() for o in ref[-1]]
    want = [float(sto.trainer_metrics([o], [y])[0]) for o, y in zip(ref_last, targets)]
    from src.model.metrics import PSNR, fused_metrics
    dev = _dev()
    msg = []
    for dt in ('f32', 'bf16'):
        net = _net(cfg, sd, dt).eval()
        with torch.no_grad():
            outs = net([x.to(dev) for x in inputs], pos.to(dev))
            got = [float(x) for x in fused_metrics(outs[-1], [y.to(dev) for y in targets], [PSNR()], per_frame=True)[:, 0]]
        worst = max(abs(a - b) for a, b in zip(got, want))
        rel = max(float((o.cpu() - r).norm() / r.norm()) for o, r in zip(outs[-1], ref_last))
        msg.append(f'{dt}: worst frame |dPSNR| {worst:.1e} dB, worst output error {rel:.1e} rel. L2')
        assert worst < 0.01, (name, dt, got, want)
        del net
    print(f'{name}, weights after {_TRAIN["steps"]} fp32 steps: PSNR vs true HR {sum(want) / len(want):.3f} dB (oracle); ' + '; '.join(msg))


@pytest.mark.parametrize('name,n,t,size', [('config 1', 1, 3, 64), ('config 2 geometry', 2, 7, 128)])
@pytest.mark.parametrize('wset', ['A', 'B', 'C'])
def test_psnr_parity_at_frozen_trained_weights(golden_dir, wset, name, n, t, size):
    """ADVICE r05 / VERDICT r05 item 1: the contract criterion |dPSNR| < 0.01 dB per frame, fp32 and bf16 storage, at three weight sets FROZEN as
    fixtures (tests/golden/trained/weights_{A,B,C}.pt: 600 fp32 training steps each at the reference YAML's training shape - A: seed 61 in round
    5's forms, the set at which the bf16 path showed +0.019 ... +0.024 dB; B: seed 61 with every Winograd launch in F(2x2) form, round 4's
    trajectory; C: seed 161; made by tests/ablation/bf16_psnr_ablation.py train) - a kernel change is then judged at the same weights before and
    after, not at wherever a chaotic training run happens to end.  Measured in round 6 (profiles/r06_c_bf16_f16_upsampler.txt): bf16 worst frame
    0.0033 / 0.0049 (A), 0.0053 / 0.0044 (B), 0.0028 / 0.0024 dB (C)."""
    w = torch.load(os.path.join(golden_dir, 'trained', f'weights_{wset}.pt'), weights_only=False)
    cfg, sd = orc.exp1_x4_config(), w['sd']
    inputs, targets, pos = orc.structured_cine(cfg, n, t, size, size, seed=71)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = orc.forward(orc.as_leaf_params(sd), cfg, [x.clone() for x in inputs], pos)
    want = [float(sto.trainer_metrics([o.detach()], [y])[0]) for o, y in zip(ref[-1], targets)]
    from src.model.metrics import PSNR, fused_metrics
    dev = _dev()
    msg = []
    for dt in ('f32', 'bf16'):
        net = _net(cfg, sd, dt).eval()
        with torch.no_grad():
            outs = net([x.to(dev) for x in inputs], pos.to(dev))
            got = [float(x) for x in fused_metrics(outs[-1], [y.to(dev) for y in targets], [PSNR()], per_frame=True)[:, 0]]
        worst = max(abs(a - b) for a, b in zip(got, want))
        msg.append(f'{dt} worst frame |dPSNR| {worst:.1e} dB')
        assert worst < 0.01, (wset, name, dt, got, want)
        del net
    print(f'frozen weights {wset} (seed {w["seed"]}, {w["env"] or "default forms"}), {name}: oracle {sum(want) / len(want):.3f} dB; ' + '; '.join(msg))


# ---------------------------------------------------------------------------------------------------------------------
# the helper stream (hipvsr.hip_ops.HipOps.aside): weight gradients and finished stages' upsamplers beside the critical chain
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype,nf', [('f32', [16, 16]), ('f32', [64, 64]), ('bf16', [64, 64])])
def test_helper_stream_changes_no_bit_even_when_it_runs_late(dtype, nf, monkeypatch):
    """Race detector for engine.forward / backward's ops.aside() blocks: three training steps (gradients after each) with the blocks on
    the current stream (RNH_ASIDE=0), on the helper stream, and on a helper stream that starts every block ~2 ms late (RNH_ASIDE_DELAY:
    a spin kernel in front of each block) - a launch on another stream that failed to wait for the helper, or rewrote a buffer it still
    reads, would then see or produce other values.  All three must agree bit for bit.  Width 16 takes the pixel-contraction weight
    gradients and the generic refine path, width 64 the Winograd / side-path forms (fp32) and the LDS-DMA weight gradients (bf16)."""
    cfg = orc.Config(in_channels=1, out_channels=1, num_features=nf, num_stages=3, refine_window_size=5, upscale_factor=4,
                     update_memory=True, num_updated_frames=2, positional_encoding=True)
    sd = orc.init_state_dict(cfg, seed=8)
    dev = _dev()
    batches = [tuple(orc.synthetic_batch(cfg, 2, 3, 32, 32, seed=90 + i)) for i in range(3)]
    runs = {}
    for mode in ('main stream', 'helper', 'late helper'):
        monkeypatch.setenv('RNH_ASIDE', '0' if mode == 'main stream' else '1')
        if mode == 'late helper':
            monkeypatch.setenv('RNH_ASIDE_DELAY', '4000000')
        else:
            monkeypatch.delenv('RNH_ASIDE_DELAY', raising=False)
        net = _net(cfg, sd, dtype).train()
        tr = _train_trainer(net, 1e-3)
        hist = []
        for inputs, targets, pos in batches:
            _, loss, _ = tr.train_step([x.to(dev) for x in inputs], [t.to(dev) for t in targets], pos.to(dev))
            torch.cuda.synchronize()
            hist.append((float(loss.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}))
        runs[mode] = hist
        del net, tr
    for mode in ('helper', 'late helper'):
        for i, ((la, ga), (lb, gb)) in enumerate(zip(runs['main stream'], runs[mode])):
            assert la == lb, (mode, i, la, lb)
            for k in ga:
                assert torch.equal(ga[k], gb[k]), (mode, 'step', i, k, float((ga[k] - gb[k]).abs().max()))


# ---------------------------------------------------------------------------------------------------------------------
# the small direct convolution (rnh_outconv_fwd: weights through the scalar cache, double-buffered halo) and its strided form
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B,H,W,Cin,Cout', [(1, 1, 1, 4, 1), (2, 3, 17, 4, 5), (1, 16, 16, 64, 5), (2, 37, 9, 20, 3), (1, 40, 33, 64, 8),
                                            (3, 18, 50, 36, 2), (1, 33, 31, 132, 5)])
def test_small_direct_convolution_vs_float64(B, H, W, Cin, Cout):
    """rnh_outconv_fwd (reference refine_net.py:201 / the per-frame form of conv1's channel 128, :149) against float64 F.conv2d: channel counts
    that are not multiples of the 16-channel chunk, one and several chunks, tile tails, single pixels."""
    import torch.nn.functional as F
    from hipvsr.hip_ops import HipOps
    dev = _dev()
    ops = HipOps(dev)
    g = torch.Generator('cpu').manual_seed(B * 1000 + H * 10 + Cin)
    x, w, b = torch.randn(B, H, W, Cin, generator=g), torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1, torch.randn(Cout, generator=g)
    y = ops.outconv_fwd(x.to(dev), w.to(dev), b.to(dev))
    torch.cuda.synchronize()
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    err = float((y.cpu().double() - ref).abs().max())
    assert err <= 2e-6 * float(ref.abs().max()) + 1e-6, err


@pytest.mark.parametrize('B,H,W,Cin', [(1, 1, 1, 64), (2, 5, 19, 64), (1, 37, 50, 64), (2, 16, 16, 16), (7, 128, 128, 64)])
def test_data_gradient_column_vs_float64(B, H, W, Cin):
    """rnh_outconv_fwd_ld as the engine uses it: conv2's data gradient w.r.t. conv1's channel 128 (autograd of refine_net.py:152) written
    into channel 128 of the 132-channel gradient tensor, pad channels zeroed, the 128 columns in front untouched."""
    import torch.nn.functional as F
    from hipvsr.hip_ops import HipOps
    dev = _dev()
    ops = HipOps(dev)
    g = torch.Generator('cpu').manual_seed(B * 1000 + H * 10 + W)
    col, C1, C1p = 2 * Cin, 2 * Cin + 1, 2 * Cin + 4
    dR, w2 = torch.randn(B, H, W, Cin, generator=g), torch.randn(Cin, C1, 3, 3, generator=g) * 0.1
    out = torch.full((B + 2, H, W, C1p), float('nan'))
    out[..., :col] = 3.0
    od = out.to(dev)
    ops.conv_to_column(dR.to(dev), w2.to(dev), col, od[1:B + 1], col, yzero=C1p - C1)
    torch.cuda.synchronize()
    x = dR.permute(0, 3, 1, 2).double()
    ref = F.conv_transpose2d(x, w2[:, col:col + 1].double(), padding=1)[:, 0]
    got = od.cpu()
    err = float((got[1:B + 1, ..., col].double() - ref).abs().max())
    assert err <= 2e-6 * float(ref.abs().max()) + 1e-6, err
    assert float(got[1:B + 1, ..., col + 1:].abs().max()) == 0.0
    assert bool((got[1:B + 1, ..., :col] == 3.0).all()) and bool(torch.isnan(got[0]).sum() == got[0][..., col:].numel())
    assert bool(torch.isnan(got[B + 1][..., col:]).all())


# ---------------------------------------------------------------------------------------------------------------------
# cold training steps beside a streaming kernel (the race of DESIGN.md section 4d (e))
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype,width,size,reps', [('f32', 16, 32, 160), ('f32', 16, 128, 120), ('bf16', 16, 32, 80)])
def test_cold_training_steps_repeat_bit_for_bit_beside_the_helper_stream(dtype, width, size, reps, monkeypatch):
    """One training step of a FRESH net (new engine, streams, scratch buffers) per repetition, every loss and gradient bitwise equal to the first
    repetition's.  RNH_POISON fills every new buffer on its stream (timing noise, and NaN wherever something unwritten is read); the weight gradients
    stream on the helper stream meanwhile.  With the direct implicit-GEMM kernel of rounds 1-3 (two loop tails that hipcc folded behind copies of
    operand registers whose loads were still in flight) 1.2-2 % of these steps came out wrong at either size
    (tools/probes/flake_width16.py, profiles/ARCHIVE/r04_ab_*): 160 / 120 repetitions miss that with probability < 10 %."""
    monkeypatch.setenv('RNH_POISON', '1')
    monkeypatch.setenv('RNH_ASIDE_OFF', 'up_fwd')
    cfg = orc.Config(in_channels=1, out_channels=1, num_features=[width, width], num_stages=3, refine_window_size=5, upscale_factor=4,
                     update_memory=True, num_updated_frames=2, positional_encoding=True)
    sd = orc.init_state_dict(cfg, seed=8)
    dev = _dev()
    inputs, targets, pos = orc.synthetic_batch(cfg, 2, 3, size, size, seed=90)
    first, wrong = None, []
    for r in range(reps):
        net = _net(cfg, sd, dtype).train()
        tr = _train_trainer(net, 1e-3)
        _, loss, _ = tr.train_step([x.to(dev) for x in inputs], [t.to(dev) for t in targets], pos.to(dev))
        torch.cuda.synchronize()
        cur = {'loss': loss.detach().clone()}
        cur.update({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
        assert all(bool(torch.isfinite(v).all()) for v in cur.values()), (r, 'non-finite')
        if first is None:
            first = cur
        elif any(not torch.equal(cur[k], first[k]) for k in cur):
            wrong.append((r, [k for k in cur if not torch.equal(cur[k], first[k])][:4]))
        del net, tr
    assert not wrong, (len(wrong), 'of', reps, wrong[:3])


def test_engines_of_a_process_share_their_side_streams():
    """Two HipOps instances of one device use the SAME side streams, role by role (DESIGN.md 4d c: a second engine with streams of its own ran
    its step 5 % slower - hardware-queue pairing), and touch_side_streams() names exactly those."""
    from hipvsr import hip_ops as ho
    dev = _dev()
    a, b = ho.HipOps(dev), ho.HipOps(dev)
    for ops in (a, b):
        ops.fork(6, bank=0)
        ops.join(6)
        ops.fork(6, bank=1)
        ops.join(6)
        with ops.aside('up_w'):
            pass
        ops.rejoin()
    assert [s.cuda_stream for s in a._banks[0]] == [s.cuda_stream for s in b._banks[0]] and len(a._banks[0]) == 3
    assert [s.cuda_stream for s in a._banks[1]] == [s.cuda_stream for s in b._banks[1]]
    assert a._helper.cuda_stream == b._helper.cuda_stream
    assert len({s.cuda_stream for s in a._banks[0] + a._banks[1] + [a._helper]}) == 7          # seven distinct streams
    ho.touch_side_streams(dev)
    assert ho._shared_stream(dev, ('helper', 0)).cuda_stream == a._helper.cuda_stream
    assert ho._shared_stream(dev, ('lstm', 1, 2)).cuda_stream == a._banks[1][2].cuda_stream
    torch.cuda.synchronize()
