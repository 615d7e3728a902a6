"""Which FORM every big launch of a training / inference step takes at a shape - decided ONCE per (shape, mode), in one place.

Until round 6 the decision was spread over the engine's forward, its backward, the memory plan and bench.py's FLOP accounting, each asking the
environment and the device again (a sample's bits changed with batch size, card and graph mode; bench's accounting ignored the capture
fallback).  ``RefineNetEngine.resolve_forms`` now returns one ``Forms`` record; the forward, the backward (through the context), ``memory_plan``,
the packing of the weights (only the forms that will be launched) and bench.py's ``config.forms`` all read that record.

The A/B switches of the product are environment variables (``SWITCHES``: name -> default); ``env_overrides()`` lists the ones set to something
else - bench.py prints them and refuses a headline line while any is set (``--allow-overrides``).
"""
import os

# every RNH_* variable that selects a form / schedule (A/B runs, diagnosis), with the value that means "the product's own choice"
SWITCHES = {
    'RNH_DTYPE': 'f32', 'RNH_GATES': 'auto',
    'RNH_WINO': '1', 'RNH_WINO_COLS': '128', 'RNH_WINO_DGRAD': '1', 'RNH_WINO_REFINE': '1', 'RNH_WINO_REFINE2': '1', 'RNH_WINO_UP': '1',
    'RNH_WINO_WGRAD': '1', 'RNH_XCOL': '1', 'RNH_XCOL_M': '1', 'RNH_R1_SPLIT': '1', 'RNH_R2_WGRAD_SPLIT': '1', 'RNH_LSTM_TILE': None,
    'RNH_WINO44': '1', 'RNH_WINO44_MIN': '1', 'RNH_WINO44_REFINE': '1', 'RNH_WINO44_REFINE2': '0', 'RNH_WINO44_REFINE_DGRAD': '1', 'RNH_WINO44_UP': '1',
    'RNH_WINO44_DGRAD': '1', 'RNH_WINO44_WGRAD': '0', 'RNH_WINO44F_WGRAD': '1', 'RNH_WINO44F_V': '1', 'RNH_UP_F16': '1',
    'RNH_PAIR': '1', 'RNH_FUSE_GATES_BWD': '1', 'RNH_FUSE_ANY': None, 'RNH_DEFER_WGRAD': '1', 'RNH_ASIDE': '1', 'RNH_ASIDE_OFF': None,
    'RNH_ASIDE_CAPTURE': '1', 'RNH_ASIDE_DELAY': None, 'RNH_LSTM_STREAMS': 'layer', 'RNH_SHARED_STREAMS': '1', 'RNH_STREAM_TOUCH': None,
    'RNH_DIRECT': '1', 'RNH_DIRECT_PS': '1', 'RNH_BF16_KC': None, 'RNH_WGRAD_TILE': None, 'RNH_WGRAD_NSPLIT': None, 'RNH_GRAPH_DP': '0',
    'RNH_POISON': None, 'RNH_CHECK': '0', 'RNH_LIB': None, 'RNH_BF16_PERSIST': None,
}

SLOT_FRACTION = 0.08          # the transformed h' of the F(4x4) cells get a slot per frame while that takes at most this share of the card
VKEEP_FRACTION = 0.12         # ... and every stage keeps its images for the weight gradients (rnh_wino44f_wgrad_v) while all of them together take at most this


def env_overrides():
    """['NAME=value', ...] of the switches that are set to something other than the product's choice (sorted)."""
    out = []
    for k, dflt in SWITCHES.items():
        v = os.environ.get(k)
        if v is not None and v != dflt:
            out.append(f'{k}={v}')
    return sorted(out)


def wino44_launch_ok(plan, B, H, W, dst_channels=0):
    """Can the launch (plan, B images of H x W) run in Winograd form F(4x4, 3x3) (rnh_wino44_cell / rnh_wino44_conv)?  The plan must be eligible
    (plans.py: plan.wino44), the images whole 4x4 tiles, and the kernels' own limits hold: fewer than 2^27 pixels per launch, fewer than 2^31
    elements in a destination of ``dst_channels`` channels (the cell: its state, addressed in bytes).  No size condition: measured against the
    F(2x2) kernel, the step is faster at every launch size tried (profiles/r05_zb_*); RNH_WINO44=0 switches the form off, RNH_WINO44_MIN=n asks
    for launches of at least n workgroups, RNH_WINO44=force ignores that minimum (A/B runs)."""
    from . import lib as L
    mode = os.environ.get('RNH_WINO44', '1')
    if mode == '0' or not getattr(plan, 'wino44', False) or (H & 3) or (W & 3):
        return False
    if plan.epilogue == L.EPI_LSTM:
        dst_channels = max(dst_channels, plan.Cout)
    if B * H * W >= 2 ** 27 or B * H * W * dst_channels >= 2 ** 31:
        return False
    wgs = -(-(B * (H // 4) * (W // 4)) // 32) * max(len(plan.colmap) // 64, 1)
    return mode == 'force' or wgs >= int(os.environ.get('RNH_WINO44_MIN', '1'))


class Forms:
    """The resolved forms of one step at (N, H, W, F) - see RefineNetEngine.resolve_forms."""
    __slots__ = ('N', 'H', 'W', 'F', 'T', 'dtype', 'need_grad', 'last_only', 'capturing', 'cells44', 'capture_fallback', 'ring', 'refine_fwd44',
                 'refine_dgrad44', 'refine2_fwd44', 'refine2_dgrad44', 'up44', 'cell_dgrad44', 'gates_bwd44', 'cell_dgrad_fused', 'cell_wgrad44f', 'refine1_wgrad44f', 'refine2_wgrad44f', 'up_wgrad44f', 'wgrad_v', 'refine1_wgrad_v', 'recompute', 'paired', 'plans44', 'names')

    def uses44(self, plan):
        """Is this plan launched in F(4x4, 3x3) form in this step?"""
        return id(plan) in self.plans44

    def describe(self):
        """A JSON-able record for bench.py's ``config.forms`` / logs: the form of every big launch class, the gate plan, pairing, overrides."""
        d = dict(self.names)
        d.update(gates='stored' if not self.recompute else f'recomputed in {self.recompute} stage(s)', paired=bool(self.paired),
                 graph_capture=bool(self.capturing), env_overrides=env_overrides())
        return d
