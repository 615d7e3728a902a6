"""Parameter layout of RefineNet: names, shapes and order of the reference ``state_dict()``
(reference src/model/nets/refine_net.py:36-59 builds the sub-modules in this order)."""
import math
from collections import OrderedDict


class NetConfig:
    """Constructor kwargs of the reference RefineNet (refine_net.py:18-19) with the same validation (:30-34)."""

    def __init__(self, in_channels, out_channels, num_features, num_stages=1, refine_window_size=5, upscale_factor=4,
                 update_memory=False, num_updated_frames=0, memory=True, positional_encoding=False):
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_features = list(num_features)
        self.num_stages, self.refine_window_size, self.upscale_factor = num_stages, refine_window_size, upscale_factor
        self.update_memory, self.num_updated_frames = update_memory, num_updated_frames
        self.memory, self.positional_encoding = memory, positional_encoding
        if upscale_factor not in [2, 3, 4, 8]:
            raise ValueError(f'The upscale factor should be 2, 3, 4 or 8. Got {upscale_factor}.')
        if update_memory == False and num_updated_frames != 0:       # noqa: E712 - mirrors the reference's test
            raise ValueError('The "update_memory" is not activated!')


def upsampler_layers(cfg):
    """[(name, cout, cin, r)]: r > 1 means the conv is followed by nn.PixelShuffle(r) (refine_net.py:194-205)."""
    c, s = cfg.num_features[0], cfg.upscale_factor
    if s == 3:
        return [('conv1', 9 * c, c, 3), ('conv2', cfg.out_channels, c, 1)]
    n = int(round(math.log2(s)))
    return [(f'conv{i + 1}', 4 * c, c, 2) for i in range(n)] + [(f'conv{n + 1}', cfg.out_channels, c, 1)]


def state_dict_spec(cfg):
    nf = cfg.num_features
    c0, cl, w = nf[0], nf[-1], cfg.refine_window_size
    spec = OrderedDict()
    spec['in_block.conv.weight'] = (c0, cfg.in_channels, 3, 3)
    spec['in_block.conv.bias'] = (c0,)
    spec['in_block.prelu.weight'] = (1,)
    for d in ('forward', 'backward'):
        for i, hd in enumerate(nf):
            cx = c0 if i == 0 else nf[i - 1]
            cin = cx + hd if cfg.memory else 2 * cx
            spec[f'{d}_lstm_block.cell_list.{i}.conv.weight'] = (4 * hd, cin, 3, 3)
            spec[f'{d}_lstm_block.cell_list.{i}.conv.bias'] = (4 * hd,)
    if cfg.positional_encoding:
        c1 = 2 * cl + 1
        spec['refine_block.body.conv1.weight'] = (c1, w * c1, 3, 3)
        spec['refine_block.body.conv1.bias'] = (c1,)
        spec['refine_block.body.conv2.weight'] = (cl, c1, 3, 3)
        spec['refine_block.body.conv2.bias'] = (cl,)
    else:
        spec['refine_block.body.conv1.weight'] = (cl, w * 2 * cl, 1, 1)
        spec['refine_block.body.conv1.bias'] = (cl,)
    spec['refine_block.prelu.weight'] = (1,)
    for name, cout, cin, _ in upsampler_layers(cfg):
        spec[f'out_block.{name}.weight'] = (cout, cin, 3, 3)
        spec[f'out_block.{name}.bias'] = (cout,)
    return spec
