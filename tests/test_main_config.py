"""Config / CLI surface of src.main (reference src/main.py:19-190) on CPU: the YAML sections resolve by name to this
package's classes, the dataset honours the sample contract of SURVEY.md 8a row A0, losses resolve torch.nn first."""
import os

import torch
import yaml

from conftest import PKG


def _cfg():
    from src.main import Cfg
    with open(os.path.join(PKG, 'configs', 'refine_net_x4_synthetic.yaml')) as f:
        return Cfg(yaml.safe_load(f))


def test_yaml_sections_resolve_by_name():
    import src
    from src.main import _get_instance
    cfg = _cfg()
    assert cfg.main.random_seed == 'vsr' and cfg.trainer.kwargs.device == 'cuda:0'
    net = _get_instance(src.model.nets, cfg.net)
    assert type(net).__name__ == 'RefineNet' and sum(p.numel() for p in net.parameters()) == 2890993
    assert 'Trainable parameters: 2.890993 M' in repr(net)
    assert hasattr(src.runner.trainers, cfg.trainer.name) and hasattr(src.callbacks.monitor, cfg.monitor.name)
    assert hasattr(src.callbacks.loggers, cfg.logger.name) and hasattr(src.data.dataloader, cfg.dataloader.name)
    loss = _get_instance(torch.nn, cfg.losses[0])
    assert isinstance(loss, torch.nn.L1Loss)
    assert isinstance(_get_instance(src.model.metrics, cfg.metrics[0]), torch.nn.Module)


def test_dataset_sample_contract():
    import src
    from src.main import _get_instance
    cfg = _cfg()
    cfg.dataset.kwargs.update(type='train')
    ds = _get_instance(src.data.datasets, cfg.dataset)
    s = ds[3]
    F, T = 7 + 12, 7
    assert len(s['lr_imgs']) == F and len(s['hr_imgs']) == T and tuple(s['pos_code'].shape) == (F, 1)
    assert tuple(s['lr_imgs'][0].shape) == (1, 32, 32) and tuple(s['hr_imgs'][0].shape) == (1, 128, 128)
    assert float(s['pos_code'].abs().max()) <= 1.0
    loader = src.data.dataloader.Dataloader(ds, batch_size=2)
    b = next(iter(loader))
    assert len(b['lr_imgs']) == F and tuple(b['lr_imgs'][0].shape) == (2, 1, 32, 32) and tuple(b['pos_code'].shape) == (2, F, 1)
    cfg.dataset.kwargs.update(type='valid')
    dv = _get_instance(src.data.datasets, cfg.dataset)
    v = dv[0]
    assert len(v['lr_imgs']) == 30 + 12 and len(v['hr_imgs']) == 30


def test_state_dict_roundtrip_with_reference_layout(golden_dir):
    from src.model.nets import RefineNet
    c = torch.load(os.path.join(golden_dir, 'g1_tiny.pt'), weights_only=False)['x3_pos0_mem0']
    net = RefineNet(**c['kwargs'])
    net.load_state_dict(c['state_dict'])                      # a reference checkpoint loads, strict
    for k, v in net.state_dict().items():
        assert torch.equal(v, c['state_dict'][k])
