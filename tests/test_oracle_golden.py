"""The CPU oracle against the golden vectors captured from the reference itself (tests/golden/make_golden.py).

Bar: bit-exact (max |delta| == 0) for outputs, losses and every parameter gradient, because the oracle
issues the same ATen ops in the same order as the reference (reference src/model/nets/refine_net.py:61-135).
"""
import os

import pytest
import torch

from oracle import refinenet_oracle as orc


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


@pytest.fixture(scope='module')
def g1(golden_dir):
    return _load(golden_dir, 'g1_tiny.pt')


G1_CASES = [f'x{s}_pos{p}_mem{m}' for s in (2, 3, 4) for p in (1, 0) for m in (1, 0)] + ['x8_pos1_mem1']


@pytest.mark.parametrize('case', G1_CASES)
def test_g1_forward_backward_bitexact(g1, case):
    c = g1[case]
    cfg = orc.Config(**c['kwargs'])
    assert list(orc.state_dict_spec(cfg).keys()) == list(c['state_dict'].keys())
    for k, shp in orc.state_dict_spec(cfg).items():
        assert tuple(c['state_dict'][k].shape) == tuple(shp), k
    outs, loss, grads = orc.step(c['state_dict'], cfg, [x.clone() for x in c['inputs']], c['targets'],
                                 c['pos_codes'])
    assert len(outs) == 3 * cfg.num_stages
    for go, gr in zip(outs, c['outputs']):
        assert len(go) == len(gr)
        for a, b in zip(go, gr):
            assert torch.equal(a, b)
    assert torch.equal(loss, c['train_loss'])
    for k, gref in c['grads'].items():
        if gref is None:
            assert grads[k] is None, k            # quirk Q1: refine_block.prelu.weight never gets a grad
        else:
            assert torch.equal(grads[k], gref), k
    assert c['grads']['refine_block.prelu.weight'] is None


@pytest.mark.parametrize('case', ['x4_pos1_mem1', 'x2_pos0_mem0', 'x3_pos1_mem0'])
def test_g1_eval_and_charbonnier(g1, case):
    c = g1[case]
    cfg = orc.Config(**c['kwargs'])
    with torch.no_grad():
        outs = orc.forward(c['state_dict'], cfg, [x.clone() for x in c['inputs']], c['pos_codes'])
        assert torch.equal(orc.eval_loss(outs, c['targets']), c['eval_loss'])
        for a, b in zip(outs[-1], c['eval_last']):
            assert torch.equal(a, b)
    _, loss, grads = orc.step(c['state_dict'], cfg, [x.clone() for x in c['inputs']], c['targets'], c['pos_codes'],
                              loss_fn=orc.charbonnier_loss)
    assert torch.equal(loss, c['charbonnier_train_loss'])
    for k, gref in c['charbonnier_grads'].items():
        if gref is not None:
            assert torch.equal(grads[k], gref), k


def test_g3_g4_losses_and_metrics(golden_dir):
    r = _load(golden_dir, 'g3_g4_losses.pt')
    outs, tg = r['outputs'], r['targets']
    fns = {'L1Loss': orc.l1_loss, 'CharbonnierLoss': orc.charbonnier_loss,
           'HuberLoss': lambda o, t: orc.huber_loss(o, t, 0.01)}
    for name, fn in fns.items():
        assert torch.equal(orc.training_loss(outs, tg, fn), r[f'{name}_train'])
        assert torch.equal(orc.eval_loss(outs, tg, fn), r[f'{name}_eval'])
        o = outs[0][0].clone().requires_grad_(True)
        v = fn(o, tg[0])
        v.backward()
        assert torch.equal(v.detach(), r[f'{name}_value'])
        assert torch.equal(o.grad, r[f'{name}_grad'])
    assert torch.equal(torch.stack(orc.frame_psnr(outs[-1], tg)).mean(), r['PSNR_metric'])
    assert torch.equal(orc.denormalize(r['denorm_in'], 'acdc'), r['denorm_acdc'])
    assert torch.equal(orc.denormalize(r['denorm_in'], 'dsb15'), r['denorm_dsb15'])
    assert torch.equal(orc.psnr(orc.denormalize(outs[0][0]), orc.denormalize(tg[0])), r['psnr_pair'])


def test_g5_edges(golden_dir):
    r = _load(golden_dir, 'g5_edges.pt')
    base = dict(in_channels=1, out_channels=1, num_features=[8, 8])
    with pytest.raises(ValueError) as e:
        orc.Config(upscale_factor=5, **base)
    assert (type(e.value).__name__, str(e.value)) == r['bad_upscale']
    with pytest.raises(ValueError) as e:
        orc.Config(num_updated_frames=2, update_memory=False, **base)
    assert (type(e.value).__name__, str(e.value)) == r['update_memory_off']
    xs = [torch.zeros(1, 1, 4, 4) for _ in range(6)]
    for U, key in ((0, 'U0_forward'), (1, 'U1_forward')):          # quirk Q2
        cfg = orc.Config(num_stages=2, update_memory=True, num_updated_frames=U, positional_encoding=True, **base)
        sd = orc.init_state_dict(cfg, 0)
        with pytest.raises(IndexError):
            orc.forward(sd, cfg, xs, torch.zeros(1, 6, 1))
        assert r[key][0] == 'IndexError'
    c = r['cycle']
    cfg = orc.Config(**c['kwargs'])
    with torch.no_grad():
        last = orc.forward(c['state_dict'], cfg, [x.clone() for x in c['inputs']], c['pos_codes'])[-1]
    assert len(last) == 30
    for a, b in zip(last, c['last']):
        assert torch.equal(a, b)
    cfg = orc.exp1_x4_config()
    n = sum(int(torch.tensor(s).prod()) for s in orc.state_dict_spec(cfg).values())
    assert n == r['param_count_exp1_x4'] == 2890993


def test_g2_cfg1_full_width(golden_dir):
    """BASELINE config 1 (x4, N=1, T=3, 64x64->256x256, [64,64,64]) regenerated from seeds: digests must match."""
    r = _load(golden_dir, 'g2_cfg1.pt')
    cfg = orc.exp1_x4_config()
    sd = orc.init_state_dict(cfg, seed=r['seed_weights'])
    inputs, targets, pos = orc.synthetic_batch(cfg, r['n'], r['t'], r['h'], r['w'], seed=r['seed_inputs'])
    outs, loss, grads = orc.step(sd, cfg, inputs, targets, pos)
    assert float(loss.double()) == r['train_loss']
    for g, grp in enumerate(outs):
        for i, o in enumerate(grp):
            assert float(o.double().sum()) == r['out_sum'][g][i]
            assert torch.equal(o[0, 0, 100:116, 100:116], r['out_crop'][g][i])
    for k, v in r['grad_l2'].items():
        if v is None:
            assert grads[k] is None
        else:
            assert float(grads[k].double().norm()) == v, k
            assert torch.equal(grads[k].flatten()[:16], r['grad_head'][k])
