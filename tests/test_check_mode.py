"""RNH_CHECK=1 (hipvsr/check_ops.py): every big launch held against float64 right behind the launch - a debugging mode (VERDICT r05
item 8).  Here: a clean training step passes with several hundred launches checked, in both cell forms and in the bf16-storage path; a WRONG PACKED
WEIGHT - one element of one plan's packed slab, the kind of fault an operand ring with a miscounted s_waitcnt leaves behind - is caught at the launch
that used it, by name; and the checker's own float64 convolution is what torch computes."""
import os

import pytest
import torch

from oracle import refinenet_oracle as orc


def test_checker_float64_convolution_is_torch_conv2d():
    """(CPU) The checker's im2col convolution and its vectorised effective-weight gather against F.conv2d / the plans' index conventions."""
    import torch.nn.functional as F
    from hipvsr.check_ops import _conv64, effective_weight
    from hipvsr.plans import NetPlans
    from hipvsr.spec import NetConfig
    g = torch.Generator().manual_seed(3)
    x, w, b = torch.randn(3, 5, 7, 6, generator=g).double(), torch.randn(4, 5, 3, 3, generator=g).double(), torch.randn(4, generator=g).double()
    torch.testing.assert_close(_conv64(x, w, b, 1), F.conv2d(x, w, b, padding=1), atol=1e-12, rtol=1e-12)
    torch.testing.assert_close(_conv64(x, w[..., :1, :1].contiguous(), None, 0), F.conv2d(x, w[..., :1, :1]), atol=1e-12, rtol=1e-12)
    P = NetPlans(NetConfig(**orc.exp1_x4_config(num_features=[16, 16], num_updated_frames=2)))
    pl = P.lstm[('forward', 1)]
    wt = torch.randn(64, 32, 3, 3, generator=g)
    we = effective_weight(pl['full'], wt)
    n = next(i for i, c in enumerate(pl['full'].colmap) if c == 37)
    assert torch.equal(we[n, 5], wt[37, 5])
    wd = effective_weight(pl['dgrad'], wt)                   # the data gradient: roles swapped, taps flipped
    assert torch.equal(wd[3, 40], torch.flip(wt[40, 3], dims=(0, 1)))


def _step(cfg, sd, inputs, targets, pos, dtype):
    from src.model.nets import RefineNet
    from src.runner.trainers import AcdcVSRRefineNetTrainer
    dev = torch.device('cuda:0')
    net = RefineNet(**dict(cfg))
    net.load_state_dict(sd)
    net = net.to(dev).set_compute_dtype(dtype).train()
    tr = object.__new__(AcdcVSRRefineNetTrainer)
    tr.net, tr.loss_fns, tr.metric_fns = net, [torch.nn.L1Loss()], []
    outs = net([x.to(dev) for x in inputs], pos.to(dev))
    loss = tr._compute_losses(outs, [t.to(dev) for t in targets])[0]
    net.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    return net, loss


def _case():
    cfg = orc.exp1_x4_config(num_features=[32, 32], num_stages=2, num_updated_frames=2)
    sd = orc.init_state_dict(cfg, seed=77)
    inputs, targets, pos = orc.synthetic_batch(cfg, 2, 2, 16, 32, seed=78)
    return cfg, sd, inputs, targets, pos


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,w44', [('f32', 'force'), ('f32', '0'), ('bf16', '1')])
def test_clean_step_passes_the_runtime_check(monkeypatch, dtype, w44):
    from hipvsr.check_ops import CheckedOps
    monkeypatch.setenv('RNH_CHECK', '1')
    monkeypatch.setenv('RNH_WINO44', w44)
    cfg, sd, inputs, targets, pos = _case()
    net, loss = _step(cfg, sd, inputs, targets, pos, dtype)
    ops = net._engine().ops
    assert isinstance(ops, CheckedOps) and ops.checked > 150, ops.checked
    fm = net._engine().resolve_forms(2, 16, 32, len(inputs))
    assert fm.cells44 == (dtype == 'f32' and w44 == 'force')
    # the checked step is the product's step: same loss as without the checker
    monkeypatch.setenv('RNH_CHECK', '0')
    _, loss0 = _step(cfg, sd, inputs, targets, pos, dtype)
    assert float(loss0) == float(loss)
    print(f'{dtype}, RNH_WINO44={w44}: {ops.checked} comparisons against float64, loss {float(loss):.6f}')


@pytest.mark.gpu
@pytest.mark.parametrize('form', ['f4x4', 'f2x2', 'bf16'])
def test_a_wrong_packed_weight_is_caught_at_the_launch_that_used_it(monkeypatch, form):
    """One element of the packed weights of ONE plan (the cell of layer 1 of the backward-direction ConvLSTM) is spoiled right behind its pack; the step
    under RNH_CHECK=1 stops at that plan's first launch and says which plan, which shape and which form."""
    from hipvsr.check_ops import CheckError
    from hipvsr.hip_ops import HipOps
    monkeypatch.setenv('RNH_CHECK', '1')
    monkeypatch.setenv('RNH_WINO44', 'force' if form == 'f4x4' else '0')
    which = '_pack44' if form == 'f4x4' else '_pack'
    store = '_packed44' if form == 'f4x4' else '_packed'
    orig = getattr(HipOps, which)

    def spoiled(self, plan, w, b=None):
        orig(self, plan, w, b)
        if plan.name == 'backward1.fwd':
            buf = getattr(self, store)[id(plan)][0]
            buf[buf.numel() // 2 + 3] += 0.75              # (one element; bf16 and fp32 slabs alike)
    monkeypatch.setattr(HipOps, which, spoiled)
    cfg, sd, inputs, targets, pos = _case()
    with pytest.raises(CheckError) as e:
        _step(cfg, sd, inputs, targets, pos, 'bf16' if form == 'bf16' else 'f32')
    msg = str(e.value)
    assert 'backward1.fwd' in msg and '16x32' in msg, msg
    assert {'f4x4': 'F(4x4,3x3)', 'f2x2': 'F(2x2,3x3)', 'bf16': 'bf16 MFMA'}[form] in msg, msg
    torch.cuda.synchronize()
    print(msg)
