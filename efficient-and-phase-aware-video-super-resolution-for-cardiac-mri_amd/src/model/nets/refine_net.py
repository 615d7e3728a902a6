"""``RefineNet`` with the reference's module surface, computed by the MI355X HIP kernels.

Same constructor, same ``forward(inputs, pos_codes)`` contract and the same ``state_dict()`` (names, shapes,
order, OIHW fp32) as reference src/model/nets/refine_net.py:10-135, so reference checkpoints load and
``python -m src.main <yaml>`` constructs it by name.  The sub-modules below exist to own the parameters
under the reference's names and default initialisation; they are never called - ``forward`` hands everything to
``hipvsr.engine.RefineNetEngine`` through one autograd Function.  There is no PyTorch fallback: on a non-HIP
device ``forward`` raises.
"""
import os

import torch
import torch.nn as nn

from hipvsr.autograd import HipOutputs, RefineNetFn
from hipvsr.spec import NetConfig, state_dict_spec, upsampler_layers
from src.model.nets.base_net import BaseNet


class _Conv(nn.Conv2d):
    def forward(self, *a, **k):       # parameter holder only
        raise RuntimeError('RefineNet sub-modules are parameter holders; call RefineNet.forward')


def _lstm_block(cfg):
    blk = nn.Module()
    cells = []
    nf = cfg.num_features
    for i, hd in enumerate(nf):
        cx = nf[0] if i == 0 else nf[i - 1]
        cell = nn.Module()
        cell.conv = _Conv(cx + hd if cfg.memory else 2 * cx, 4 * hd, kernel_size=3, padding=1, bias=True)
        cells.append(cell)
    blk.cell_list = nn.ModuleList(cells)
    return blk


class RefineNet(BaseNet):
    def __init__(self, in_channels, out_channels, num_features, num_stages=1, refine_window_size=5, upscale_factor=4,
                 update_memory=False, num_updated_frames=0, memory=True, positional_encoding=False):
        super().__init__()
        cfg = NetConfig(in_channels, out_channels, num_features, num_stages, refine_window_size, upscale_factor,
                        update_memory, num_updated_frames, memory, positional_encoding)
        if not isinstance(num_features, (list, tuple)) or len(num_features) < 1:
            raise ValueError('Inconsistent list length.')
        self.cfg = cfg
        self.in_channels, self.out_channels, self.num_features = in_channels, out_channels, num_features
        self.num_stages, self.refine_window_size, self.upscale_factor = num_stages, refine_window_size, upscale_factor
        self.update_memory, self.num_updated_frames = update_memory, num_updated_frames

        nf, cl, w = list(num_features), num_features[-1], refine_window_size
        self.in_block = nn.Module()
        self.in_block.conv = _Conv(in_channels, nf[0], kernel_size=3, padding=1)
        self.in_block.prelu = nn.PReLU(num_parameters=1, init=0.2)
        self.forward_lstm_block = _lstm_block(cfg)
        self.backward_lstm_block = _lstm_block(cfg)
        self.refine_block = nn.Module()
        self.refine_block.body = nn.Module()
        if positional_encoding:
            c1 = 2 * cl + 1
            self.refine_block.body.conv1 = _Conv(w * c1, c1, kernel_size=3, padding=1)
            self.refine_block.body.conv2 = _Conv(c1, cl, kernel_size=3, padding=1)
        else:
            self.refine_block.body.conv1 = _Conv(w * 2 * cl, cl, kernel_size=1)
        self.refine_block.prelu = nn.PReLU(num_parameters=1, init=0.2)     # registered, never applied (quirk Q1)
        self.out_block = nn.Module()
        for name, cout, cin, _ in upsampler_layers(cfg):
            setattr(self.out_block, name, _Conv(cin, cout, kernel_size=3, padding=1))

        self._param_names = list(state_dict_spec(cfg).keys())
        got = [k for k, _ in self.named_parameters()]
        assert got == self._param_names, (got, self._param_names)
        self._eng = None
        self._flat_grad = None
        # compute precision of the HIP engine: 'f32' (the reference's) or 'bf16' (bf16 storage / bf16 MFMA with fp32
        # accumulation, BASELINE.json configs[2]).  Not a constructor argument - the reference's constructor is the
        # boundary - and never visible in state_dict(): parameters and checkpoints are fp32 either way.
        self.compute_dtype = os.environ.get('RNH_DTYPE', 'f32')
        # activation-memory plan of the training step (hipvsr.engine.RefineNetEngine.recompute_gates): 'store' keeps the ConvLSTM gates of
        # the supervised frames for the backward, 'recompute' re-runs the cell launch there instead (bit-identical gradients, one more
        # launch per cell and frame), 'auto' recomputes only where the stored-gates step is estimated not to fit the device
        self.gate_memory = 'auto'
        # bf16-storage path only: {class: 'f32'} overrides of the element type a class of forward tensors is stored in
        # (hipvsr.engine.RefineNetEngine.STORAGE_CLASSES); {} = the path's own layout
        self.storage = {}

    def set_storage(self, storage):
        """Storage-class overrides of the bf16-storage path (see RefineNetEngine); takes effect at the next forward."""
        self.storage = dict(storage or {})
        self._eng = None
        return self

    def set_compute_dtype(self, dtype):
        """'f32' or 'bf16'; takes effect at the next forward (the engine and its packed weights are rebuilt)."""
        if dtype not in ('f32', 'bf16'):
            raise ValueError(f"compute dtype must be 'f32' or 'bf16', got {dtype!r}")
        self.compute_dtype = dtype
        return self

    def set_gate_memory(self, mode):
        """'store', 'recompute' or 'auto' (see above); takes effect at the next forward."""
        if mode not in ('store', 'recompute', 'auto'):
            raise ValueError(f"gate memory plan must be 'store', 'recompute' or 'auto', got {mode!r}")
        self.gate_memory = mode
        return self

    def _engine(self):
        dev = self.in_block.conv.weight.device
        if self._eng is None or self._eng.ops.device != dev or self._eng.dtype != self.compute_dtype:
            if dev.type != 'cuda':
                raise RuntimeError(f'RefineNet (HIP) needs its parameters on a HIP device; they are on {dev}. '
                                   'There is no CPU path in this package.')
            from hipvsr.engine import RefineNetEngine
            from hipvsr.hip_ops import HipOps
            ops = HipOps(dev)
            if os.environ.get('RNH_CHECK', '0') == '1':
                # debugging mode: every big launch is held against float64 right behind the launch (hipvsr/check_ops.py) - a checker around the
                # HIP path, not a path of its own
                from hipvsr.check_ops import CheckedOps
                ops = CheckedOps(ops)
            self._eng = RefineNetEngine(self.cfg, ops, dtype=self.compute_dtype,
                                        storage=self.storage if self.compute_dtype == 'bf16' else None)
        self._eng.gate_memory = self.gate_memory
        return self._eng

    def forward(self, inputs, pos_codes):
        params = [p for _, p in self.named_parameters()]
        O_all = RefineNetFn.apply(self, list(inputs), pos_codes, *params)
        S, _, TN = O_all.shape[:3]
        N = inputs[0].shape[0]
        T = TN // N
        # inference shortcut (not in the reference's module; its predictor consumes outputs[-1] only,
        # acdc_vsr_refinenet_predictor.py:62): with `net.last_group_only = True` and no gradient required, the other
        # 3*S - 1 output groups are not computed and are returned as None
        only_last = bool(getattr(self, 'last_group_only', False)) and not (torch.is_grad_enabled() and any(p.requires_grad for p in params))
        groups = []
        for s in range(S):
            for br in range(3):
                if only_last and not (s == S - 1 and br == 2):
                    groups.append(None)
                    continue
                groups.append([O_all[s, br, i * N:(i + 1) * N].permute(0, 3, 1, 2) for i in range(T)])
        out = HipOutputs(groups)
        out.packed, out.ops = O_all, self._engine().ops
        return out
