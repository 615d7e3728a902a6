"""The structured cine of SURVEY.md section 8(d) (oracle.refinenet_oracle.structured_cine): the generator behind the PSNR-parity
tests of tests/test_parity_r03.py.  CPU only."""
import torch
import torch.nn.functional as F

from oracle import refinenet_oracle as orc


def test_structured_cine_contract():
    cfg = orc.exp1_x4_config()
    inputs, targets, pos = orc.structured_cine(cfg, 2, 3, 16, 12, seed=5)
    assert len(inputs) == 15 and len(targets) == 3 and tuple(pos.shape) == (2, 15, 1)
    assert tuple(inputs[0].shape) == (2, 1, 16, 12) and tuple(targets[0].shape) == (2, 1, 64, 48)
    assert inputs[0].dtype == torch.float32 and float(pos.abs().max()) <= 1.0
    lo, hi = (0 - 54.089) / 48.084, (255 - 54.089) / 48.084
    for t in targets + inputs:
        assert float(t.min()) >= lo - 1e-5 and float(t.max()) <= hi + 1e-5          # grey levels in [0, 255]
    # LR = avg_pool(scale) of HR, frame U + i of the inputs belongs to target i (normalisation is affine)
    for i, t in enumerate(targets):
        torch.testing.assert_close(F.avg_pool2d(t, 4), inputs[6 + i], atol=1e-5, rtol=1e-5)
    # structured: neighbouring HR pixels are strongly correlated (sigma = 3 px blur), consecutive frames differ but little
    a = targets[0][:, :, :, 1:].flatten()
    b = targets[0][:, :, :, :-1].flatten()
    assert float(torch.corrcoef(torch.stack([a, b]))[0, 1]) > 0.9
    d = float((targets[1] - targets[0]).abs().mean())
    assert 1e-4 < d < 0.5
    again = orc.structured_cine(cfg, 2, 3, 16, 12, seed=5)
    assert all(torch.equal(x, y) for x, y in zip(inputs, again[0])) and torch.equal(pos, again[2])
