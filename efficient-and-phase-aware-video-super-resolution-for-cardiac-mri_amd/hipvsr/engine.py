"""RefineNet forward and hand-written backward, scheduled over an ``ops`` backend.

Forward follows reference src/model/nets/refine_net.py:61-135 (frame loop, stage loop, grad / no-grad
frames, three output groups per stage, in-place feature update); backward is what ``loss.backward()``
(reference src/runner/trainers/acdc_vsr_refinenet_trainer.py:46) makes autograd do for that graph, written
out by hand so that every step is one of the kernels of include/refinenet_hip.h:

* frames are stored frame-major, image index b = frame * N + n, NHWC;
* everything without a temporal dependence is batched over frames: the input block over all F frames, the
  refine block over all F-4 windows, the upsampler over 3 branches x T frames, the LSTM weight gradients
  over the T supervised frames;
* no-grad ("update") frames (refine_net.py:74-93, :179-183) simply do not appear in the backward: their
  hidden states enter as constants;
* ``refine_block.prelu.weight`` never receives a gradient (quirk Q1, refine_net.py:150-155).

The ``ops`` object is ``hipvsr.hip_ops.HipOps`` in the product.  tests/ substitutes a torch implementation of
the same interface to check this scheduling logic against the oracle on machines without a GPU.
"""
import os
from collections import OrderedDict

from . import lib as L
from .forms import VKEEP_FRACTION, SLOT_FRACTION, Forms, wino44_launch_ok
from .plans import Dst, NetPlans, Src


def _span_src(store, k0, n):
    t, o = store.span(k0, n)
    return Src(t, img_off=o)


class Context:
    """What the forward keeps for the backward."""
    __slots__ = ('N', 'H', 'W', 'F', 'T', 'x_all', 'P4', 'stages', 'packed_dgrad', 'tail_bf16', 'recompute', 'forms')

    def __init__(self):
        self.stages = []
        self.packed_dgrad = False
        self.recompute = 0


class FrameStore:
    """N images for every frame k in [lo, hi) of one per-frame tensor of a stage (a hidden state, a cell state, the features).

    Liveness (reference src/model/nets/refine_net.py:61-135 keeps every frame's tensors alive through autograd; what the hand-written
    backward reads again is less): of a ConvLSTM layer's h only the frames around the T supervised ones come back - as the second
    K source of the cell's weight gradient and, for the top layer, as refine windows -, of c the supervised frames and their
    predecessors, of the features the supervised frames.  So a store is allocated in up to three pieces - the frames before
    ``keep`` = [klo, khi), the frames of ``keep``, the frames behind - and ``release()`` drops the outer two as soon as the stage's forward
    has passed them (no copy, no event: the pieces are separate allocations, every access below is per frame or inside ``keep``).
    ``ring`` > 0: the frames outside ``keep`` share ``ring`` slots in processing order (``step`` = +1: ascending k) instead of having a piece
    each - only for tensors whose frame k is read by nobody but the same chain's next cell, i.e. the cell state: the two launches
    are ordered by their stream, so slot reuse needs no event."""

    def __init__(self, ops, N, lo, hi, keep, shape, dtype, ring=0, step=1, alloc=True):
        self.N, self.lo, self.hi, self.step, self.R = N, lo, hi, step, ring
        self.segs = []                                           # (k0, k1, tensor, kept)
        self.ring = None
        if not alloc:                                            # the caller hands the pieces over (put)
            self.klo, self.khi = keep if keep is not None else (lo, lo)
            return
        if keep is None:
            keep = (lo, lo)
        klo, khi = max(keep[0], lo), min(keep[1], hi)
        if khi <= klo:
            klo = khi = lo
        self.klo, self.khi = klo, khi
        mk = lambda a, b: ops.empty((b - a) * N, *shape, dtype=dtype)               # noqa: E731
        if ring:
            self.ring = mk(0, ring) if (klo - lo) + (hi - khi) > 0 else None
            if khi > klo:
                self.segs.append((klo, khi, mk(klo, khi), True))
        else:
            self.ring = None
            for a, b, kept in ((lo, klo, False), (klo, khi, True), (khi, hi, False)):
                if b > a:
                    self.segs.append((a, b, mk(a, b), kept))

    def put(self, a, b, t):
        """Frames [a, b) are the tensor t (a piece produced elsewhere); kept if that is the ``keep`` range."""
        self.segs.append((a, b, t, (a, b) == (self.klo, self.khi)))

    def loc(self, k):
        """(tensor, image offset) of frame k."""
        for a, b, t, _ in self.segs:
            if a <= k < b:
                return t, (k - a) * self.N
        if self.ring is None or not self.lo <= k < self.hi:
            raise IndexError(f'frame {k} is not held (frames [{self.lo}, {self.hi}), kept [{self.klo}, {self.khi}))')
        idx = k - self.lo if self.step > 0 else self.hi - 1 - k
        return self.ring, (idx % self.R) * self.N

    def src(self, k, **kw):
        t, o = self.loc(k)
        return Src(t, img_off=o, **kw)

    def view(self, k):
        t, o = self.loc(k)
        return t[o:o + self.N]

    def span(self, k0, n):
        """(tensor, image offset) of the n consecutive frames from k0 - they must lie in one piece."""
        for a, b, t, _ in self.segs:
            if a <= k0 and k0 + n <= b:
                return t, (k0 - a) * self.N
        raise IndexError(f'frames [{k0}, {k0 + n}) do not lie in one piece of {[(a, b) for a, b, _, _ in self.segs]}')

    def frames(self, k0, k1):
        t, o = self.span(k0, k1 - k0)
        return t[o:o + (k1 - k0) * self.N]

    def pieces(self, k0=None, k1=None):
        """[(a, b)]: the frame ranges of [k0, k1) piece by piece."""
        k0, k1 = self.lo if k0 is None else k0, self.hi if k1 is None else k1
        return [(max(a, k0), min(b, k1)) for a, b, _, _ in self.segs if min(b, k1) > max(a, k0)]

    def release(self):
        """Drop everything outside ``keep``."""
        self.segs = [sg for sg in self.segs if sg[3]]
        self.ring = None
        self.lo, self.hi = self.klo, self.khi

    def nbytes(self):
        ts = [t for _, _, t, _ in self.segs] + ([self.ring] if self.ring is not None else [])
        return sum(t.numel() * t.element_size() for t in ts)


class RefineNetEngine:
    STORAGE_CLASSES = ('feat', 'h', 'r1', 'r', 'sb', 'ys')

    def __init__(self, cfg, ops, dtype='f32', storage=None):
        """dtype 'f32': everything fp32 (the reference's precision).  dtype 'bf16' (BASELINE.json configs[2]): the bf16-storage
        path - feature maps, hidden states, saved gates and the activation gradients between the big convolutions live in
        HBM as bf16 and all 3x3 convolutions run on bf16 MFMA with fp32 accumulators (rnh_conv_bf16 / rnh_wgrad_bf16);
        the cell state c and its gradient, the upsampler's inner feature map (which the collapsed tail kernels consume),
        the outputs, the loss and every parameter gradient stay fp32; the state_dict is fp32 either way.
        ``storage``: {class: 'f32' | 'bf16'} overrides of the element type a class of forward tensors is STORED in under dtype 'bf16'
        (STORAGE_CLASSES: the features across stages, the hidden states, refine conv1's / conv2's output, the upsampler's input sums, its
        inner maps) - the numerical ablation of tools/bf16_ablation.py; the MFMA operands are rounded to bf16 either way."""
        import torch
        if dtype not in ('f32', 'bf16'):
            raise ValueError(f"compute dtype must be 'f32' or 'bf16', got {dtype!r}")
        self.cfg, self.ops, self.dtype = cfg, ops, dtype
        self.bf16 = dtype == 'bf16'
        self.act = torch.bfloat16 if self.bf16 else torch.float32
        self.f32 = torch.float32
        storage = dict(storage or {})
        if set(storage) - set(self.STORAGE_CLASSES) or set(storage.values()) - {'f32', 'bf16'}:
            raise ValueError(f'storage overrides must map {self.STORAGE_CLASSES} to f32 / bf16, got {storage!r}')
        if storage and not self.bf16:
            raise ValueError("storage overrides only exist under dtype 'bf16'")
        self.storage = storage
        self.st_dt = {k: (torch.float32 if storage.get(k) == 'f32' else self.act) for k in self.STORAGE_CLASSES}
        self.plans = NetPlans(cfg, bf16=self.bf16)
        if self.bf16:
            # The upsampler's PixelShuffle convolutions in the forward contract in IEEE half (plan.f16w: rnh_pack_weights_f16 + the f16 MFMA form of
            # rnh_conv_bf16; the collapsed tail does the same inside rnh_uptail_fwd_bf16): the outputs are two linear maps away from these weights,
            # and with 8-bit weights in exactly these layers the PSNR at trained weights moved by up to 0.014 + 0.009 dB - a fixed perturbation of
            # the weights is a systematic shift, not noise (profiles/r06_a_bf16_psnr_ablation.txt, r06_b_bf16_weight_rounding.txt) - against
            # 0.002 + 0.001 with 11 bits.  Where the convolution's input is bf16 in whole 32-channel chunks (the form's condition); else bf16 as before.
            P = self.plans
            tail_bf16 = len(P.up) > 1 and storage.get('ys') != 'f32' and ops.uptail_bf16_supported(P.C, P.up[-1]['r'], cfg.out_channels)
            for i, u in enumerate(P.up):
                src_dt = (self.f32 if len(P.up) == 1 else self.st_dt['sb']) if i == 0 else (self.st_dt['ys'] if tail_bf16 else self.f32)
                u['fwd'].f16w = src_dt is torch.bfloat16 and P.C % 32 == 0 and os.environ.get('RNH_UP_F16', '1') != '0'
        self.hw = cfg.refine_window_size // 2
        if (3 if self.plans.pos else 2) * cfg.refine_window_size > L.MAX_SRC:
            raise ValueError(f'refine_window_size {cfg.refine_window_size} needs more than {L.MAX_SRC} conv sources')
        if cfg.num_features[0] != cfg.num_features[-1]:
            raise ValueError('num_features[0] must equal num_features[-1] (residual add, refine_net.py:102)')

    # ------------------------------------------------------------------------------------------------
    # activation memory ("activation-memory plan (recompute vs. store)")
    gate_memory = 'auto'          # 'store' | 'recompute' | 'auto' (RNH_GATES overrides): see recompute_gates
    AUTO_FRACTION = 0.80          # 'auto' stores the gates while the estimated peak of the step stays below this share of the device memory

    def memory_plan(self, N, H, W, F, recompute=False):
        """Estimated HBM bytes of one training step (forward with gradients + backward) at this shape: what every stage keeps for the
        backward (see FrameStore: the hidden / cell states around the supervised frames, the gates of the supervised frames unless
        they are recomputed, the upsampler's intermediates), the largest transient working set on top of it, and their sum.
        ``recompute``: how many stages (the first ones; True = all) recompute their gates instead of storing them."""
        cfg, P = self.cfg, self.plans
        U, S, hw, w = cfg.num_updated_frames, cfg.num_stages, self.hw, cfg.refine_window_size
        n_rc = S if recompute is True else int(recompute)
        T = F - 2 * U
        px, ea = N * H * W, (2 if self.bf16 else 4)
        nf, C, Cl = P.nf, P.C, P.Cl
        c1p = getattr(P, 'C1p', 0)
        tail_bf16 = self.bf16 and len(P.up) > 1
        e_sb = 4 if len(P.up) == 1 else ea
        e_y = ea if tail_bf16 else 4
        per = dict(h_lower=2 * sum(nf[:-1]) * (T + 1) * px * ea, c=2 * sum(nf) * (T + 1) * px * 4,
                   gates=0 if n_rc == S else 2 * sum(nf) * 4 * T * px * ea, feat=T * px * C * ea, r1=T * px * c1p * ea, sb=3 * T * px * C * e_sb)
        ys, scale = 0, 1
        for u in P.up[:-1]:                                     # the PixelShuffle stages in front of the collapsed tail keep their output
            scale *= u['r']
            ys += 3 * T * px * scale * scale * C * e_y
        per['ys'] = ys
        kept = S * sum(per.values()) - n_rc * per['gates'] + 2 * nf[-1] * px * ea * ((S - 1) * F + (U + T + hw))        # + the top layer's h, whole
        o_all = 2 * S * 3 * T * px * cfg.upscale_factor ** 2 * cfg.out_channels * 4               # outputs and their gradient
        fwd_t = (2 * sum(nf[:-1]) * (F - T - 1) * px * ea + 4 * sum(nf) * 2 * px * 4 + (F - 2 * hw) * px * Cl * ea +
                 max(U - hw, 1) * px * c1p * ea + 2 * F * px * C * ea)
        fm = self._conv_forms(N, H, W, F)                         # (the eager step's forms; a captured step at a ring shape runs F(2x2) cells: less)
        if fm.cells44:
            # the cells in F(4x4, 3x3) form read transformed inputs, 2.25 x 4 bytes per element: the features of every frame and the h' of every cell -
            # a slot per frame, or a ring of four where that would take more than 8 % of the card (resolve_forms)
            fwd_t += F * px * C * 9 + 2 * (min(F, fm.ring) if fm.ring else F) * sum(nf) * px * 9
        if fm.wgrad_v:
            # ... and kept by every stage until its weight gradients have run: they copy their x operand from these images (rnh_wino44f_wgrad_v)
            kept += S * (F * C + 2 * F * sum(nf)) * px * 9
            if fm.refine2_fwd44 and fm.refine2_wgrad44f:
                kept += S * T * px * 2 * Cl * 9                 # (and R1's hidden-state channels as refine conv2 read them)
            if fm.up_wgrad44f and fm.up44 and fm.up44[0]:
                kept += S * 3 * T * px * C * 9                  # (and the first PixelShuffle convolution's transformed input)
        bwd_t = (2 * sum(nf) * 4 * T * px * ea + 2 * sum(nf[:-1]) * T * px * ea + 3 * T * px * C * ea * (1 + scale * scale) + (T + 2 * hw) * px * c1p * ea +
                 4 * T * px * C * ea + (2 * sum(nf) * 6 * px * 4 if n_rc else 0))
        if fm.up44 and fm.up44[0]:
            fwd_t += 3 * T * px * C * 9                         # the transformed input of the first PixelShuffle convolution (beside the next stage's wavefront)
        if fm.refine_dgrad44:
            bwd_t += (T + 2 * hw) * px * getattr(P, 'r1_cols', 0) * 9      # the transformed dR1 of refine conv1's data gradient
        if fm.refine2_fwd44:
            fwd_t += (F - 2 * hw) * px * 2 * Cl * 9              # R1's hidden-state channels, transformed for refine conv2
        if fm.refine2_dgrad44:
            bwd_t += T * px * Cl * 9                            # the transformed dR of refine conv2's data gradient
        # the opt-in forms' scratch (both off by default): the transformed gate gradients of every chain; refine conv1's tile-major operands
        if fm.cell_dgrad44:
            bwd_t += 2 * sum(nf) * 4 * px * 9
        if (not self.bf16) and H % 4 == 0 and W % 16 == 0 and os.environ.get('RNH_WINO44F_WGRAD', '1') != '0':
            # the cell's weight gradient in F(4x4)-tile form (rnh_wino44f_wgrad): the K-split partial sums (its inputs are read in place)
            bwd_t += 64 * 36 * 128 * 256 * 4
        if os.environ.get('RNH_WINO44_WGRAD', '0') == '1' and fm.cells44 and P.pos and P.r1_wino:
            bwd_t += (2 * (T + 2 * hw) * Cl + T * getattr(P, 'r1_cols', 0)) * px * 9 + 64 * w * 36 * 128 * 128 * 4
        # the weight gradients of a stage run beside the next (earlier) stage's backward (engine.backward): until that stage's chains are
        # joined, the stage's dgates and hidden states stay alive although the stage itself has been released
        dgates = 2 * sum(nf) * 4 * T * px * ea
        held = dgates + per['h_lower'] + per['feat'] + 2 * nf[-1] * px * ea * (U + T + hw)
        last = sum(per.values()) - (per['gates'] if n_rc == S else 0) + 2 * nf[-1] * px * ea * (U + T + hw)      # what the last stage keeps
        peak = kept + o_all + max(fwd_t, bwd_t, bwd_t + held - last if S > 1 else 0)
        return dict(per_stage=per, recomputing_stages=n_rc, kept=kept, outputs=o_all, forward_transient=fwd_t, backward_transient=bwd_t,
                    held_for_weight_gradients=held, peak=peak)

    # ------------------------------------------------------------------------------------------------
    # forms (hipvsr/forms.py): ONE decision per (shape, mode), read by the forward, the backward, memory_plan, _pack and bench.py
    def resolve_forms(self, N, H, W, F, need_grad=True, last_only=False, capturing=False):
        """Which form every big launch of a step at this shape takes.  fp32 path: the ConvLSTM cells in Winograd form F(4x4, 3x3) (rnh_wino44_cell)
        wherever every cell plan of the net is eligible and the images are whole 4x4 tiles - all cells or none -, their transformed h' in a slot per
        frame, or in a ring of four where the slots would take more than SLOT_FRACTION of the card (BASELINE config 4 at N = 16: 62 GB; the
        device's TOTAL memory, not what is free right now: the decision must not change from one step to the next); under HIP-graph capture the
        ring cannot be captured (hipStreamEndCapture, DESIGN section 8 hazard 3), so a capture at a ring shape falls back to F(2x2) cells
        (``capture_fallback``: eager and graph steps differ in their low bits there, and only there).  Refine conv1's forward / data gradient
        and the PixelShuffle convolutions follow the cells where their own launches qualify.  The result is cached per argument tuple."""
        import copy
        f = copy.copy(self._conv_forms(N, H, W, F, need_grad, last_only, capturing))
        f.recompute = self.recompute_stages(N, H, W, F) if need_grad else 0           # (asks memory_plan, which reads the convolution forms)
        return f

    def _conv_forms(self, N, H, W, F, need_grad=True, last_only=False, capturing=False):
        """resolve_forms without the gate-memory plan (which itself depends on these forms through memory_plan)."""
        key = (N, H, W, F, bool(need_grad), bool(last_only), bool(capturing), os.environ.get('RNH_WINO44'), os.environ.get('RNH_WINO44_MIN'),
               os.environ.get('RNH_PAIR'), os.environ.get('RNH_FUSE_GATES_BWD'), os.environ.get('RNH_WINO44_WGRAD'), os.environ.get('RNH_WINO44_DGRAD'), os.environ.get('RNH_WINO44F_WGRAD'), os.environ.get('RNH_WINO44F_V'))
        cache = self.__dict__.setdefault('_forms_cache', {})
        if key in cache:
            return cache[key]
        ops, P, cfg = self.ops, self.plans, self.cfg
        U, hw, w = cfg.num_updated_frames, self.hw, cfg.refine_window_size
        T = F - 2 * U
        nf, Lr = P.nf, P.L
        f = Forms()
        f.N, f.H, f.W, f.F, f.T, f.dtype, f.need_grad, f.last_only, f.capturing = N, H, W, F, T, self.dtype, bool(need_grad), bool(last_only), bool(capturing)
        hip = hasattr(ops, 'wino44_cell')                        # (the CPU double of tests/ has no F(4x4) kernels: every launch in its plain form)
        cell_plans = [P.lstm[k][kind] for k in P.lstm for kind in ('full', 'first')]
        cells44 = (not self.bf16) and hip and all(wino44_launch_ok(pl, N, H, W) for pl in cell_plans)
        total = ops.total_memory() if hasattr(ops, 'total_memory') else 0
        slots = 2 * F * sum(nf) * N * H * W * 9               # a slot per frame: 2.25 x 4 bytes per element of every h'
        ring = cells44 and bool(total) and slots > SLOT_FRACTION * total
        f.capture_fallback = bool(cells44 and ring and capturing)
        if f.capture_fallback:
            cells44 = False
        f.cells44, f.ring = bool(cells44), (4 if cells44 and ring else 0)
        tiles32 = (N * (H // 4) * (W // 4)) % 32 == 0          # a frame = whole tile blocks (the window slots of refine conv1 are frames apart)
        r1 = cells44 and P.pos and P.r1_wino
        f.refine_fwd44 = bool(r1 and not f.ring and tiles32 and wino44_launch_ok(P.r1_fwd_h, (F - 2 * hw) * N, H, W, dst_channels=P.C1p))
        f.refine_dgrad44 = bool(r1 and need_grad and tiles32 and wino44_launch_ok(P.r1_dgrad_h, (T + 2 * hw) * N, H, W, dst_channels=P.Cl)
                                and wino44_launch_ok(P.r1_dgrad_h, T * N, H, W, dst_channels=P.Cl))
        r2 = cells44 and P.pos and getattr(P, 'r2_wino', False)
        f.refine2_fwd44 = bool(r2 and wino44_launch_ok(P.r2_fwd_h, (F - 2 * hw) * N, H, W, dst_channels=P.Cl) and wino44_launch_ok(P.r2_fwd_h, N, H, W, dst_channels=P.Cl))
        f.refine2_dgrad44 = bool(r2 and need_grad and wino44_launch_ok(P.r2_dgrad_h, T * N, H, W, dst_channels=P.C1p))
        nb = (1 if last_only and not need_grad else 3)
        n_up = len(P.up) - 1 if (hasattr(ops, 'uptail_fwd_supported') and ops.uptail_fwd_supported(P.up[-1]['r'], cfg.out_channels)) else len(P.up)
        f.up44 = [bool(cells44 and u['r'] == 2 and wino44_launch_ok(u['fwd'], nb * T * N, H * 2 ** i, W * 2 ** i)) for i, u in enumerate(P.up[:n_up])]
        fused = bool(cfg.memory and hasattr(ops, 'lstm_bwd_fusable') and
                     all(ops.lstm_bwd_fusable(P.lstm[k]['dgrad'], P.lstm[k]['cx'], P.lstm[k]['hd']) for k in P.lstm))
        f.cell_dgrad_fused = fused
        # the cell's data gradient in F(4x4, 3x3) form wherever the gate backward can write the transformed gate gradients itself (rnh_wino44_gates_bwd:
        # whole 8 x 4 blocks of tiles); with a transform launch of its own the form barely pays (RNH_WINO44_DGRAD=force asks for it anyway)
        gt = bool(hasattr(ops, 'wino44_gates_bwd_supported') and all(ops.wino44_gates_bwd_supported(H, W, P.lstm[k]['hd']) for k in P.lstm))
        f.cell_dgrad44 = bool(cells44 and need_grad and not fused and all(wino44_launch_ok(P.lstm[k]['dgrad'], N, H, W) for k in P.lstm) and
                              (gt or os.environ.get('RNH_WINO44_DGRAD') == 'force'))
        f.gates_bwd44 = bool(f.cell_dgrad44 and gt)
        f.recompute = None
        f.paired = bool(ops.pair_cells(N, H, W)) if hasattr(ops, 'pair_cells') else False
        f.plans44 = set()
        if cells44:
            f.plans44 |= {id(pl) for pl in cell_plans}
        if f.refine_fwd44:
            f.plans44.add(id(P.r1_fwd_h))
        if f.refine_dgrad44:
            f.plans44.add(id(P.r1_dgrad_h))
        if f.refine2_fwd44:
            f.plans44.add(id(P.r2_fwd_h))
        if f.refine2_dgrad44:
            f.plans44.add(id(P.r2_dgrad_h))
        f.plans44 |= {id(u['fwd']) for u, on in zip(P.up, f.up44) if on}
        if f.cell_dgrad44:
            f.plans44 |= {id(P.lstm[k]['dgrad']) for k in P.lstm}
        # the names bench.py prints (config.forms)
        pl0 = P.lstm[('forward', 0)]

        def conv_form(pl, on44):
            if self.bf16:
                return 'direct 3x3 on bf16 MFMA (rnh_conv_bf16)' + (', IEEE-half weights on the f16 MFMA form' if getattr(pl, 'f16w', False) else '')
            if on44:
                return 'Winograd F(4x4,3x3) on transformed inputs (rnh_wino44_*)'
            return 'Winograd F(2x2,3x3) (rnh_conv_wino)' if getattr(pl, 'wino', False) else 'implicit GEMM (rnh_conv_igemm)'
        cell = conv_form(pl0['full'], cells44)
        if cells44:
            cell += ', transformed h\' in a ring of 4' if f.ring else ', a transformed-h\' slot per frame'
        if f.capture_fallback:
            cell += " [capture fallback: the F(4x4) form's ring cannot be captured at this shape]"
        # the cell's weight gradient in F(4x4)-tile form with both transforms fused (rnh_wino44f_wgrad) where the kernel takes the call (HipOps.wgrad asks
        # rnh_wino44f_wgrad_supported per call: the cell's sources always qualify, the image must be whole quads of tiles)
        w44f_ok = bool(hip and not self.bf16 and need_grad and os.environ.get('RNH_WINO44F_WGRAD', '1') != '0' and H % 4 == 0 and W % 16 == 0 and
                       os.environ.get('RNH_WINO', '1') != '0' and os.environ.get('RNH_WINO_WGRAD', '1') != '0')
        allw = os.environ.get('RNH_WINO44F_WGRAD') == 'all'
        w44f = f.cell_wgrad44f = bool(w44f_ok and (allw or getattr(pl0['wgrad'], 'wino44f', False)))
        f.refine1_wgrad44f = bool(w44f_ok and P.pos and P.r1_wino and (allw or getattr(P.r1_wgrad_h, 'wino44f', False)) and os.environ.get('RNH_WINO44_WGRAD', '0') != '1')
        f.refine2_wgrad44f = bool(w44f_ok and P.pos and P.r2_wino and (allw or getattr(P.r2_wgrad_h, 'wino44f', False)) and os.environ.get('RNH_R2_WGRAD_SPLIT', '1') != '0')
        f.up_wgrad44f = bool(w44f_ok and n_up > 0 and P.C % 64 == 0 and (allw or getattr(P.up[0]['wgrad'], 'wino44f', False)))
        # ... with the x operand COPIED from the transformed images the forward's cells and refine conv1 read (rnh_wino44f_wgrad_v) instead of transformed
        # again - where every frame's image has a slot of its own (no ring) and keeping the images of ALL stages until the backward takes at most
        # VKEEP_FRACTION of the card (BASELINE config 2: 30.1 GB by this count, 10.5 %); the frame in front of the supervised ones must exist (U >= 1)
        vkeep = cfg.num_stages * (F * P.C + 2 * F * sum(nf)) * N * H * W * 9
        f.wgrad_v = bool(w44f and cells44 and not f.ring and U >= 1 and hasattr(ops, '_wgrad44f_v') and os.environ.get('RNH_WINO44F_V', '1') != '0' and
                         bool(total) and vkeep <= VKEEP_FRACTION * total and P.C % 32 == 0 and all(h % 32 == 0 for h in nf))
        f.refine1_wgrad_v = bool(f.wgrad_v and f.refine1_wgrad44f and f.refine_fwd44)
        vnote = ", x operand copied from the forward's transformed images (rnh_wino44f_wgrad_v)"
        f22w = 'Winograd F(2x2,3x3) tiles (rnh_wino_wgrad; pixel contraction where it does not take the call)'
        names = dict(cell=cell,
                     cell_dgrad=(conv_form(pl0['dgrad'], f.cell_dgrad44) + (' + the next frame\'s gate backward in its epilogue' if fused else '') +
                                 (', transformed gate gradients written by the gate backward (rnh_wino44_gates_bwd)' if f.gates_bwd44 else '')) if need_grad else None,
                     cell_wgrad=('bf16 MFMA over LDS-DMA rows (rnh_wgrad_bf16)' if self.bf16 else
                                 'Winograd F(4x4,3x3) tiles, both transforms fused (rnh_wino44f_wgrad)' + (vnote if f.wgrad_v else '') if w44f else f22w) if need_grad else None)
        if P.pos:
            r1p = P.r1_fwd_h if P.r1_wino else (P.r1_fwd_a if getattr(P, 'r1_split', False) else P.r1_fwd)
            names['refine1_fwd'] = conv_form(r1p, f.refine_fwd44) + (' on the cells\' transformed h\'' if f.refine_fwd44 else '')
            if getattr(P, 'r2_wino', False):
                names['refine2_fwd'] = conv_form(P.r2_fwd_h, f.refine2_fwd44)
            if need_grad:
                names['refine1_dgrad'] = conv_form(P.r1_dgrad_h if P.r1_wino else P.r1_dgrad, f.refine_dgrad44) + ', gather form'
                if getattr(P, 'r2_wino', False):
                    names['refine2_dgrad'] = conv_form(P.r2_dgrad_h, f.refine2_dgrad44)
                w44 = (not self.bf16) and cells44 and os.environ.get('RNH_WINO44_WGRAD', '0') == '1' and P.r1_wino
                names['refine1_wgrad'] = ('F(4x4)-tile Winograd (rnh_wino44_wgrad_*)' if w44 else ('bf16 MFMA over LDS-DMA rows (rnh_wgrad_bf16)' if self.bf16 else
                                          ('Winograd F(4x4,3x3) tiles, both transforms fused (rnh_wino44f_wgrad)' + (vnote if f.refine1_wgrad_v else '') if f.refine1_wgrad44f else f22w)))
        else:
            names['refine1_fwd'] = '1x1 ' + conv_form(P.r1_fwd, False)
        for i, u in enumerate(P.up[:n_up]):
            names[f'up{i + 1}_fwd'] = conv_form(u['fwd'], f.up44[i]) + ', PixelShuffle in the store'
        if n_up and need_grad and not self.bf16:
            names['up_wgrad'] = 'Winograd F(4x4,3x3) tiles, both transforms fused (rnh_wino44f_wgrad)' if f.up_wgrad44f else f22w
        names['tail'] = ('last PixelShuffle conv + final conv collapsed (rnh_uptail_*' + ('_bf16: f16 MFMA, IEEE-half composed weights)' if self.bf16 and n_up and
                         self.storage.get('ys') != 'f32' and ops.uptail_bf16_supported(P.C, P.up[-1]['r'], cfg.out_channels) else ')')) if n_up < len(P.up) else 'rnh_outconv_*'
        f.names = names
        cache[key] = f
        return f

    def cells_f4x4(self, N, H, W, F=None, capturing=False):
        """Do the ConvLSTM cells of a training step at this shape run in Winograd form F(4x4, 3x3)?  (resolve_forms; F defaults to the smallest legal sequence)"""
        return self._conv_forms(N, H, W, F if F is not None else 2 * self.cfg.num_updated_frames + 1, capturing=capturing).cells44

    def refine_f4x4(self, N, H, W, F, capturing=False):
        """Does refine conv1's forward over the hidden states run in F(4x4, 3x3) form (rnh_wino44_conv) at this shape?"""
        return self._conv_forms(N, H, W, F, capturing=capturing).refine_fwd44

    def refine_dgrad_f4x4(self, N, H, W, T, capturing=False):
        """Does refine conv1's data gradient over the hidden states run in F(4x4, 3x3) form at this shape (T supervised frames)?"""
        return self._conv_forms(N, H, W, T + 2 * self.cfg.num_updated_frames, capturing=capturing).refine_dgrad44

    def up_f4x4(self, N, H, W, F=None, capturing=False):
        """Does the first PixelShuffle convolution of the upsampler run in F(4x4, 3x3) form at this shape?"""
        fm = self._conv_forms(N, H, W, F if F is not None else 2 * self.cfg.num_updated_frames + 1, capturing=capturing)
        return bool(fm.up44 and fm.up44[0])

    def _mem(self, label):
        """RNH_MEMLOG=1: (label, allocated bytes) at the engine's stage boundaries, in self.memlog (calibration of memory_plan)."""
        if os.environ.get('RNH_MEMLOG') == '1' and hasattr(self.ops, 'mem_allocated'):
            self.__dict__.setdefault('memlog', []).append((label, self.ops.mem_allocated()))

    def _sum(self, label, t):
        """RNH_DBGSUM=1 (race hunting, tools/probes/flake_width16.py): (label, float64 sum of t as a device scalar) in self.dbgsum - launched
        on the current stream right where the engine stands, no synchronisation."""
        if os.environ.get('RNH_DBGSUM') == '1' and t is not None:
            self.__dict__.setdefault('dbgsum', []).append((label, t.detach().double().sum()))
        if os.environ.get('RNH_DBGKEEP') == '1' and t is not None:          # the tensor itself stays alive (no launch): inspected after the step
            self.__dict__.setdefault('dbgkeep', []).append((label, t))

    def recompute_stages(self, N, H, W, F):
        """How many stages (the first n of S) recompute their ConvLSTM gates in the backward - one more cell launch per cell and supervised
        frame of those stages - instead of reading gates the forward stored.  'store': 0, 'recompute': all, an integer: that many,
        'auto': the fewest with which the estimated peak of the step fits AUTO_FRACTION of the device memory (0 wherever the
        stored-gates step fits)."""
        S = self.cfg.num_stages
        mode = str(os.environ.get('RNH_GATES', self.gate_memory))
        if mode.isdigit():
            return min(int(mode), S)
        if mode not in ('store', 'recompute', 'auto'):
            raise ValueError(f"gate memory plan must be 'store', 'recompute', 'auto' or a number of stages, got {mode!r}")
        if mode != 'auto':
            return S if mode == 'recompute' else 0
        # the budget is what this process can still get: the device's memory, less what other engines of this process that are not ours to free,
        # and other processes on the card, hold (free memory + this process's reserved pool; never more than the device has)
        total = self.ops.memory_budget() if hasattr(self.ops, 'memory_budget') else (self.ops.total_memory() if hasattr(self.ops, 'total_memory') else None)
        if not total:
            return 0
        choice = S
        for n in range(S):
            if self.memory_plan(N, H, W, F, recompute=n)['peak'] <= self.AUTO_FRACTION * total:
                choice = n
                break
        key = (N, H, W, F, choice)
        if key not in self.__dict__.setdefault('_rc_logged', set()):
            self._rc_logged.add(key)
            if choice:
                import logging
                logging.getLogger(__name__).info('gate memory plan (auto): %d of %d stages recompute their gates at N=%d %dx%d F=%d (budget %.1f GiB)',
                                                 choice, S, N, H, W, F, total / 2**30)
        return choice

    def recompute_gates(self, N, H, W, F):
        """Does any stage recompute its gates at this shape?"""
        return self.recompute_stages(N, H, W, F) > 0

    # ------------------------------------------------------------------------------------------------
    def param_order(self):
        from .spec import state_dict_spec
        return list(state_dict_spec(self.cfg).keys())

    def _views(self, params):
        """Views of parameters that plans address under a key of their own (no copy): the last output channel of refine conv1 as a
        (window, C1, 3, 3) convolution over single frames (plans.xcol_m)."""
        P = self.plans
        if getattr(P, 'xcol_m', False) and P.r1x_key not in params:
            params = dict(params)
            w1 = params[P.r1_fwd.wkey]
            params[P.r1x_key] = w1[P.C1 - 1].view(self.cfg.refine_window_size, P.C1, 3, 3)
        return params

    def _pack(self, params, which, fm=None):
        """Re-lay the weights of the forward's (which = 'fwd') or the backward's plans - for a plan that has a Winograd F(4x4, 3x3) form only the form
        this step launches it in (``fm``: the step's resolved forms; None: every form the plan has)."""
        P = self.plans
        for pl in P.conv_plans():
            is_dgrad = pl.transposed
            if (which == 'fwd') == (not is_dgrad):
                kw = {}
                if fm is not None and getattr(pl, 'wino44', False):
                    kw = dict(f22=not fm.uses44(pl), f44=fm.uses44(pl))
                self.ops.pack(pl, params[pl.wkey], params[pl.bkey] if pl.bkey else None, **kw)

    # ------------------------------------------------------------------------------------------------
    def forward(self, params, inputs, pos_codes, need_grad, last_only=False):
        """forward_impl behind a guard: if it raises (an allocation that does not fit, a refused launch) while side streams and the helper stream
        still hold work on buffers that the unwinding stack is about to hand back to the allocator, everything in flight is drained first - a
        caller that catches the error and goes on must not compute on memory another stream still writes."""
        if need_grad and getattr(self.plans, 'inconv_bwd_error', None):
            raise ValueError(self.plans.inconv_bwd_error)
        try:
            return self.forward_impl(params, inputs, pos_codes, need_grad, last_only)
        except BaseException:
            if hasattr(self.ops, 'quiesce'):
                self.ops.quiesce()
            raise

    def forward_impl(self, params, inputs, pos_codes, need_grad, last_only=False):
        """inputs: list[F] of (N, Cin, H, W) tensors; pos_codes: (N, F, 1).
        Returns (O_all, ctx): O_all is (S, 3, T*N, sH, sW, Cout) with image index i*N + n.
        last_only (inference, reference predictor acdc_vsr_refinenet_predictor.py:62 consumes outputs[-1] only): the
        upsampler runs for the fused group of the last stage alone; the other 3*S - 1 groups of O_all stay unwritten."""
        ops, cfg, P = self.ops, self.cfg, self.plans
        U, S, hw, w = cfg.num_updated_frames, cfg.num_stages, self.hw, cfg.refine_window_size
        F = len(inputs)
        T = F - 2 * U
        if U == 0 or T <= 0 or U < hw:
            # reference: inputs[U:-U] is empty for U == 0 and refine_maps is over-run for U < w//2
            # (refine_net.py:66, :112) -> IndexError('list index out of range')
            raise IndexError('list index out of range')
        N, Cin, H, W = inputs[0].shape
        nf, Lr, C, Cl = P.nf, P.L, P.C, P.Cl
        s_up = cfg.upscale_factor
        TN = T * N

        ctx = Context()
        ctx.N, ctx.H, ctx.W, ctx.F, ctx.T = N, H, W, F, T
        # the form of every big launch of this step, decided once (hipvsr/forms.py): read here, by the backward (ctx.forms) and by _pack
        fm = ctx.forms = self.resolve_forms(N, H, W, F, need_grad, last_only, bool(ops.capturing()) if hasattr(ops, 'capturing') else False)
        ctx.recompute = n_rc = fm.recompute                    # stages 0 .. n_rc-1 store no gates
        params = self._views(params)
        x_all = ops.stack_inputs(inputs)                       # (F*N, H, W, Cin)
        ctx.x_all = x_all
        self._pack(params, 'fwd', fm)
        act, f32 = self.act, self.f32
        # the features of the F frames, in pieces: the backward reads the T supervised frames again (first K source of layer 0's
        # weight gradient), the update frames on both sides only feed this stage's forward
        sup = (U, U + T) if need_grad else None
        sdt = self.st_dt
        feat = FrameStore(ops, N, 0, F, sup, (H, W, C), sdt['feat'], alloc=False)
        for a, b in ([(0, U), (U, U + T), (U + T, F)] if need_grad else [(0, F)]):
            y = ops.inconv_fwd(x_all[a * N:b * N], params['in_block.conv.weight'], params['in_block.conv.bias'],
                               params['in_block.prelu.weight'])
            if self.bf16:
                y = ops.cast(y, sdt['feat'])                      # the input block's features cross into bf16 storage here
            feat.put(a, b, y)
        P4 = (ops.phase_plane(pos_codes, N, F, H, W, dtype=act, channels=P.pw) if self.bf16 else
              ops.phase_plane(pos_codes, N, F, H, W)) if P.pos else None
        # the tail kernels read their input in fp32 (csrc/uptail.hip) or, for the x4 / x8 nets' r = 2 tail, in bf16
        # (csrc/uptail_bf16.hip): then every inner feature map of the upsampler and its gradient are bf16 too.  With a single
        # PixelShuffle stage the tail's input is Sb, kept fp32
        tail_bf16 = self.bf16 and len(P.up) > 1 and self.storage.get('ys') != 'f32' and ops.uptail_bf16_supported(C, P.up[-1]['r'], cfg.out_channels)
        ctx.tail_bf16 = tail_bf16
        sb_dt = f32 if len(P.up) == 1 else sdt['sb']
        ctx.P4 = P4
        O_all = ops.empty(S, 3, TN, s_up * H, s_up * W, cfg.out_channels)

        aside_keep = []
        pair = fm.paired                                        # the two directions' cells of a layer in one launch

        def run_stage(s, feat):
            """Stage s of the forward; returns the features the next stage starts from.  A function of its own so that the stage's transients die
            with its locals before the next stage allocates."""
            st = dict(feat=feat)
            # ---- bidirectional ConvLSTM over the frames (refine_net.py:82-93) ----------------------------
            # One cell launch (N images) fills the chip for well under a millisecond, so its ramp-up and tail matter.
            # Cell (direction d, layer l, frame k) only needs (d, l-1, k) and (d, l, k-1): every (d, l) gets its own
            # stream, ordered along k by the stream and against the layer below by an event, and up to 2*L cells
            # of the layer/frame wavefront run concurrently.
            dirs = ('forward', 'backward')
            # In the last stage nothing reads the hidden states past the last refine window (the feature update after it is
            # dead, quirk Q5): the forward direction stops at frame U+T-1+hw, the backward one at U-hw, and the refine
            # block only computes the T supervised windows.  Outputs and gradients are unchanged.
            last = s == S - 1
            F_s = U + T + hw if last else F
            for d in dirs:
                fwd = d == 'forward'
                lo_d, hi_d = (0, F_s) if fwd else (F - F_s, F)
                # what the backward reads again: the supervised frames and, per direction, the frame in front of them (second K source of
                # the weight gradient, previous cell state); the top layer's h feeds the refine windows and stays whole
                keep = ((U - 1, U + T) if fwd else (U, U + T + 1)) if need_grad else None
                st[d] = dict(H=[FrameStore(ops, N, lo_d, hi_d, keep if l < Lr - 1 else (lo_d, hi_d), (H, W, hd), sdt['h']) for l, hd in enumerate(nf)],
                             C=[FrameStore(ops, N, lo_d, hi_d, keep, (H, W, hd), f32, ring=2, step=1 if fwd else -1) for hd in nf],
                             G=[ops.empty(TN, H, W, 4 * hd, dtype=act) for hd in nf] if need_grad and s >= n_rc else None)
            def cell_call(d, l, idx):
                """The conv() arguments of cell (direction d, layer l) at wavefront slot idx: (plan, sources, N, H, W, keyword arguments)."""
                k = idx if d == 'forward' else F - 1 - idx
                prev = None if idx == 0 else (k - 1 if d == 'forward' else k + 1)
                Hb, Cb, Gb = st[d]['H'], st[d]['C'], st[d]['G']
                grad_frame = Gb is not None and U <= k < U + T
                pl = P.lstm[(d, l)]
                xin = feat if l == 0 else Hb[l - 1]
                srcs = [xin.src(k)]
                if cfg.memory:
                    if prev is not None:
                        srcs.append(Hb[l].src(prev))
                        plan = pl['full']
                    else:
                        plan = pl['first']
                else:
                    srcs.append(xin.src(k))
                    plan = pl['full']
                return plan, srcs, N, H, W, dict(lstm=dict(
                    hd=pl['hd'], c_prev=Cb[l].view(prev) if prev is not None else None,
                    h_out=Hb[l].view(k), c_out=Cb[l].view(k),
                    gates_out=Gb[l][(k - U) * N:(k - U + 1) * N] if grad_frame else None))

            # The cells in Winograd form F(4x4, 3x3) (rnh_wino44_cell, csrc/conv_wino44.hip) where every cell plan of the net is packed for it
            # and the launch is large enough (HipOps.wino44_ok): the kernel reads its inputs in transform-domain form, written by a kernel of
            # its own - the features of every frame once, in front of the wavefront, and every cell's h' right behind the cell on the cell's
            # stream (the event the layer above waits for is recorded behind it): one transform serves both readers of an h'.
            use44 = st['use44'] = fm.cells44
            ref44, R44 = False, 1
            if use44:
                VF = ops.wino44_v(N, H, W, C, frames=F)
                for k in (range(F) if F_s == F else sorted(set(range(F_s)) | set(range(F - F_s, F)))):
                    ops.wino44_transform(feat.src(k), N, H, W, VF[k])
                # the transformed h' of a (direction, layer) live in a ring of R44 slots (2.25 x the bytes of h each): slot idx % R44 is written
                # behind cell idx and read by cell idx + 1 of the layer (same stream) and by cell idx of the layer above - whose event the
                # layer waits for before it overwrites the slot R44 cells later (it rarely has to: the layers run in step)
                # - only where a slot per frame would take more than 8 % of the card (BASELINE config 4 at N = 16: 62 GB), and not under HIP-graph
                # capture: the layer's stream then waits on the stream above while that waits on it, and hipStreamEndCapture (ROCm 7.2) never
                # returns from two streams that reference each other (hazard 3)
                R44 = min(F_s, fm.ring) if fm.ring else F_s
                VH = {d: [ops.wino44_v(N, H, W, hd, frames=R44) for hd in nf] for d in dirs}
                read44 = {}
                # refine conv1's forward reads the top layer's h' of both directions (refine_net.py:170-181): in the same form
                # (rnh_wino44_conv) it takes the transformed h' the cells wrote - every frame's then, the slots in frame order, and a window's
                # frames whole tile blocks apart
                ref44 = fm.refine_fwd44
            slot44 = lambda d, l, idx: (idx if d == 'forward' else F_s - 1 - idx) if ref44 and l == Lr - 1 else idx % R44   # noqa: E731

            def cell44_call(d, l, idx):
                """(plan, transformed sources, lstm arguments) of cell (d, l) at wavefront slot idx in F(4x4, 3x3) form."""
                plan, srcs, _, _, _, kw = cell_call(d, l, idx)
                k = idx if d == 'forward' else F - 1 - idx
                vx = VF[k] if l == 0 else VH[d][l - 1][slot44(d, l - 1, idx)]
                vs = [vx] + ([VH[d][l][slot44(d, l, idx - 1)]] if cfg.memory and idx > 0 else [vx])[:len(srcs) - 1]
                return plan, vs, kw['lstm']

            def cell44(d, l, idx, launch=True):
                k = idx if d == 'forward' else F - 1 - idx
                if launch:
                    plan, vs, lstm = cell44_call(d, l, idx)
                    ops.wino44_cell(plan, vs, N, H, W, lstm)
                if l > 0 and idx + R44 < F_s:                                # (only with a ring: R44 < F_s)
                    read44[(d, l, idx)] = ops.record()                       # slot idx % R44 of the layer below has been read
                if l + 1 < Lr or (cfg.memory and idx + 1 < F_s) or ref44:   # somebody reads this h' in transformed form (ref44: only the top layer gets here)
                    if l + 1 < Lr and idx >= R44:
                        ops.wait(read44.pop((d, l + 1, idx - R44)))
                    ops.wino44_transform(st[d]['H'][l].src(k), N, H, W, VH[d][l][slot44(d, l, idx)])

            if pair:
                # small images (the reference YAML's 32 x 32 crops): a cell launch is a fraction of the chip, and the two directions' cells of a
                # layer at the same slot are independent and of equal geometry - ONE launch for both (ops.conv_pair), one stream per layer
                ops.fork(2 * Lr)                                # (2 L logical streams as below: logical stream l is layer l's HIP stream)
                for idx in range(F_s):
                    below = None
                    for l in range(Lr):
                        with ops.side(l):
                            if below is not None:
                                ops.wait(below)
                            if use44:
                                # (both directions' cells in one launch, then each direction's event / transform as behind a launch of its own)
                                ops.wino44_cell_pair([cell44_call(d, l, idx) for d in dirs], N, H, W)
                                for d in dirs:
                                    cell44(d, l, idx, launch=False)
                            else:
                                ops.conv_pair([cell_call(d, l, idx) for d in dirs])
                            below = ops.record() if l + 1 < Lr else None
                ops.join(2 * Lr)
            else:
                ops.fork(2 * Lr)
                for idx in range(F_s):
                    for di, d in enumerate(dirs):
                        below = None
                        for l in range(Lr):
                            with ops.side(di * Lr + l):
                                if below is not None:
                                    ops.wait(below)
                                if use44:
                                    cell44(d, l, idx)
                                else:
                                    plan, srcs, _, _, _, kw = cell_call(d, l, idx)
                                    ops.conv(plan, srcs, N, H, W, **kw)
                                below = ops.record() if l + 1 < Lr else None
                ops.join(2 * Lr)
            VT = None
            if use44:
                VT = {d: VH[d][-1] for d in dirs} if ref44 else None      # the top layer's transformed h', slot = frame - first frame of the direction
                if need_grad and fm.wgrad_v:
                    # the weight gradients of this stage copy their x operand from these images (rnh_wino44f_wgrad_v): row of (direction, layer, frame)
                    st['V44'] = dict(VF=VF, VH=VH, row=lambda d, l, k: slot44(d, l, k if d == 'forward' else F - 1 - k))
                del VF, VH
            self._mem(f'fwd stage {s}: wavefront done')
            for d in dirs:                                      # the wavefront has passed: only what the backward reads stays
                for l in range(Lr):
                    if l < Lr - 1:
                        st[d]['H'][l].release()
                    st[d]['C'][l].release()
            HF, HB = st['forward']['H'][-1], st['backward']['H'][-1]      # top layer: every frame of [lo, hi) in one piece

            # ---- phase-aware refine block over all windows (refine_net.py:157-185) -----------------------
            w0, nwin = (U - hw, T) if last else (0, F - 2 * hw)     # first window computed, number of windows
            R = ops.empty(nwin * N, H, W, Cl, dtype=sdt['r'])

            def win_srcs(a):
                out = []
                for j in range(w):
                    out += [HF.src(a + j), HB.src(a + j)]
                    if P.pos:
                        out.append(Src(P4, img_off=(a + j) * N))
                return out
            if P.pos:
                # conv1's output R1 comes back in the backward for the T supervised windows only (conv2's weight gradient): they get
                # a buffer of their own, the windows on both sides a transient one each
                wsegs = [(w0, U - hw), (U - hw, U - hw + T), (U - hw + T, w0 + nwin)] if need_grad else [(w0, w0 + nwin)]
                for a, b in wsegs:
                    if b <= a:
                        continue
                    nw = b - a
                    srcs = win_srcs(a)
                    R1 = ops.empty(nw * N, H, W, P.C1p, dtype=sdt['r1'])
                    hfs, hbs, p4s = HF.frames(a, b + w - 1), HB.frames(a, b + w - 1), P4[a * N:(b + w - 1) * N]   # the source frames of these windows
                    if P.r1_wino and VT is not None:
                        mtf = N * (H // 4) * (W // 4) // 32                 # tile blocks per frame
                        lo_b = F - F_s                                      # first frame the backward direction holds
                        ops.wino44_conv(P.r1_fwd_h, [(VT[d], (a + j - (0 if d == 'forward' else lo_b)) * mtf) for j in range(w) for d in dirs], nw * N, H, W,
                                        Dst(R1, P.r1_cols))
                        ops.refine_phase_bias(R1, p4s, params[P.r1_fwd_h.wkey], N, w, Cl, P.r1_cols)
                    elif P.r1_wino:
                        ops.conv(P.r1_fwd_h, [sc for sc in srcs if sc.t is not P4], nw * N, H, W, dsts=[Dst(R1, P.r1_cols)])
                        ops.refine_phase_bias(R1, p4s, params[P.r1_fwd_h.wkey], N, w, Cl, P.r1_cols)
                    elif P.xcol_m:
                        # bf16 path: 2*Cl columns in one launch; the last channel frame by frame (one small convolution over the source
                        # frames whose columns are the window slots, then a sum over the slots)
                        ops.conv(P.r1_fwd_a, srcs, nw * N, H, W, dsts=[Dst(R1, 2 * Cl)])
                        nfr = nw + w - 1
                        Z5 = ops.empty(nfr * N, H, W, 8)
                        ops.conv(P.r1x_fwd, [HF.src(a), HB.src(a), Src(P4, img_off=a * N)], nfr * N, H, W, dsts=[Dst(Z5, 8)])
                        ops.xcol_combine_m(Z5, params[P.r1_fwd.bkey], R1, N, w, 2 * Cl)
                    elif P.r1_split:
                        ops.conv(P.r1_fwd_a, srcs, nw * N, H, W, dsts=[Dst(R1, 2 * Cl)])
                        ops.conv(P.r1_fwd_b, srcs, nw * N, H, W, dsts=[Dst(R1, P.C1p - 2 * Cl, c0=2 * Cl)])
                    else:
                        ops.conv(P.r1_fwd, srcs, nw * N, H, W, dsts=[Dst(R1, P.r1_cols)])
                    if P.xcol:
                        ops.refine_xcol_fwd([hfs, hbs, p4s], params[P.r1_fwd.wkey], params[P.r1_fwd.bkey], R1, N, w, Cl)
                    ro = (a - w0) * N
                    if P.r2_wino:
                        # (the small launch stores, the Winograd launch accumulates: the read-modify-write of R then rides in the
                        # MFMA-bound kernel instead of doubling the traffic of the HBM-bound one; a + b = b + a, the bits are the same)
                        ops.conv(P.r2_fwd_x, [Src(R1, c0=2 * Cl, nch=P.C1p - 2 * Cl)], nw * N, H, W, dsts=[Dst(R, Cl, img_off=ro)])
                        if fm.refine2_fwd44:
                            # (F(4x4, 3x3) form on one transform of R1's 2 Cl hidden-state channels; the supervised windows' image also serves conv2's
                            # weight gradient in the backward)
                            VR1 = ops.wino44_v(nw * N, H, W, 2 * Cl)
                            ops.wino44_transform(Src(R1, nch=2 * Cl), nw * N, H, W, VR1[0])
                            ops.wino44_conv(P.r2_fwd_h, [(VR1[0], 0)], nw * N, H, W, Dst(R, Cl, accumulate=True, img_off=ro))
                            if need_grad and a == U - hw and fm.wgrad_v and fm.refine2_wgrad44f:
                                st['VR1'] = VR1
                            del VR1
                        else:
                            ops.conv(P.r2_fwd_h, [Src(R1, nch=2 * Cl)], nw * N, H, W, dsts=[Dst(R, Cl, accumulate=True, img_off=ro)])
                    else:
                        ops.conv(P.r2_fwd, [Src(R1)], nw * N, H, W, dsts=[Dst(R, Cl, img_off=ro)])
                    if need_grad and a == U - hw:
                        st['R1'] = R1                           # windows U-hw .. U-hw+T-1
                    del R1
            else:
                ops.conv(P.r1_fwd, win_srcs(w0), nwin * N, H, W, dsts=[Dst(R, Cl)])

            # ---- three output groups through the upsampler (refine_net.py:100-113, :194-205) --------------
            fc = feat.frames(U, U + T)
            skip_up = last_only and not need_grad
            if skip_up and s < S - 1:
                nb = 0                                        # no output group of this stage is consumed
            elif skip_up:
                nb = 1                                        # the fused group only
                Sb = ops.empty(TN, H, W, C, dtype=sb_dt)
                ops.add(Sb, fc, R[(U - hw - w0) * N:(U - hw - w0 + T) * N])
            else:
                nb = 3
                Sb = ops.empty(3 * TN, H, W, C, dtype=sb_dt)
                ops.add(Sb[0:TN], fc, HF.frames(U, U + T))
                ops.add(Sb[TN:2 * TN], fc, HB.frames(U, U + T))
                ops.add(Sb[2 * TN:], fc, R[(U - hw - w0) * N:(U - hw - w0 + T) * N])
            Oview = O_all[s] if nb == 3 else O_all[s, 2:3]
            cur, h, wd, Ys = (Sb if nb else None), H, W, []
            fused_tail = ops.uptail_fwd_supported(P.up[-1]['r'], cfg.out_channels)
            # Nothing of the next stage reads this stage's outputs: the upsampler runs ASIDE (ops.aside: a helper stream behind the sums
            # above), beside the next stage's ConvLSTM wavefront, and the forward rejoins it at its end.  Its buffers are allocated here,
            # on the main stream (an allocation on another stream than the capture's origin fails under HIP-graph capture).
            tail_u = P.up[-1] if (nb and fused_tail) else None
            Yb, hh, ww = [], H, W
            for u in (P.up[:-1] if tail_u is not None else P.up) if nb else []:
                hh, ww = hh * u['r'], ww * u['r']
                Yb.append(ops.empty(nb * TN, hh, ww, C, dtype=sdt['ys'] if tail_bf16 else f32))
            if not need_grad and aside_keep:
                # inference keeps nothing for a backward: the previous stage's upsampler (long finished: a whole ConvLSTM wavefront and refine block
                # ago) is rejoined here, so that at most ONE stage's upsampler buffers are alive
                ops.rejoin()
                aside_keep.clear()
            # (the PixelShuffle convolutions in F(4x4, 3x3) form where the cells run in it: one transform of the input, allocated here like Yb)
            up44 = list(fm.up44[:len(Yb)]) if nb else []
            Vup = [ops.wino44_v(nb * TN, H * 2 ** i, W * 2 ** i, C)[0] if f44 else None for i, f44 in enumerate(up44)]
            with ops.aside('up_fwd'):
                for i, (u, Y) in enumerate(zip(P.up, Yb)):
                    if up44[i]:
                        ops.wino44_transform(Src(cur), nb * TN, h, wd, Vup[i])
                        ops.wino44_conv(u['fwd'], [(Vup[i], 0)], nb * TN, h, wd, ps=(Y, u['r']))
                    else:
                        ops.conv(u['fwd'], [Src(cur)], nb * TN, h, wd, ps=(Y, u['r']))
                    Ys.append(Y)
                    cur, h, wd = Y, h * u['r'], wd * u['r']
                if tail_u is not None:
                    # last PixelShuffle conv + final conv as one composed 5x5 convolution (csrc/uptail.hip): the
                    # r*r*C-channel tensor between them is never formed, neither here nor in the backward
                    r = tail_u['r']
                    ops.uptail_fwd(cur, params[tail_u['fwd'].wkey], params[tail_u['fwd'].bkey], params[P.last_w], params[P.last_b], r,
                                   Oview.reshape(nb * TN, h * r, wd * r, cfg.out_channels))
                elif nb:
                    ops.outconv_fwd(cur, params[P.last_w], params[P.last_b], out=Oview.reshape(nb * TN, h, wd, cfg.out_channels))
            aside_keep.append((Sb if nb else None, Yb, Vup))   # alive until the forward has rejoined the helper stream
            if tail_u is None and nb:
                Ys = Ys[:-1]                                  # the tail's output is not needed by the collapsed backward
            if need_grad:
                st['Sb'], st['Ys'] = Sb, Ys
                if fm.wgrad_v and fm.up_wgrad44f and nb == 3:
                    st['Vup'] = list(Vup)                       # the PixelShuffle convolutions' transformed inputs (None where a launch ran in another form)
                ctx.stages.append(st)

            # ---- feature update (refine_net.py:118-133), out of place ---------------------------------------
            if S > 1 and s < S - 1:
                nfeat = FrameStore(ops, N, 0, F, sup, (H, W, C), sdt['feat'])
                for lo_r, hi_r, other in ((0, hw, lambda a, b: HF.frames(a, b)), (hw, F - hw, lambda a, b: R[(a - hw) * N:(b - hw) * N]),
                                          (F - hw, F, lambda a, b: HB.frames(a, b))):
                    for a, b in nfeat.pieces(lo_r, hi_r):
                        ops.add(nfeat.frames(a, b), feat.frames(a, b), other(a, b))
                feat.release()                                # (the stage's entry keeps the supervised frames alive)
                feat = nfeat
            return feat

        for s in range(S):
            feat = run_stage(s, feat)
            self._mem(f'fwd stage {s}: end')
        ops.rejoin()                                              # the upsamplers of all stages have written their outputs
        aside_keep.clear()
        if need_grad:
            feat.release()
        return O_all, (ctx if need_grad else None)

    # ------------------------------------------------------------------------------------------------
    def backward(self, params, ctx, dO_all, flat=None):
        """backward_impl behind the same guard as forward."""
        try:
            return self.backward_impl(params, ctx, dO_all, flat)
        except BaseException:
            if hasattr(self.ops, 'quiesce'):
                self.ops.quiesce()
            raise

    def backward_impl(self, params, ctx, dO_all, flat=None):
        """dO_all: gradient of the loss w.r.t. O_all (same shape).  Returns an OrderedDict name -> gradient in
        state-dict order (None for parameters that take no part, quirk Q1).  If ``flat`` (a 1-D buffer with
        room for every parameter) is given the gradients are views into it (for a single all-reduce)."""
        ops, cfg, P = self.ops, self.cfg, self.plans
        U, S, hw, w = cfg.num_updated_frames, cfg.num_stages, self.hw, cfg.refine_window_size
        N, H, W, F, T = ctx.N, ctx.H, ctx.W, ctx.F, ctx.T
        nf, Lr, C, Cl = P.nf, P.L, P.C, P.Cl
        TN = T * N
        act = self.act
        fm = ctx.forms
        params = self._views(params)
        self._pack(params, 'bwd', fm)

        grads, touched = OrderedDict(), set()
        off = 0
        for name in self.param_order():
            p = params[name]
            if name == 'refine_block.prelu.weight':
                grads[name] = None
                if flat is not None:
                    off += p.numel()
                continue
            if flat is not None:
                grads[name] = flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
            else:
                grads[name] = ops.empty(*p.shape)

        def acc(name):
            a = name in touched
            touched.add(name)
            return a

        dfeat_next = None

        def stage_backward(s, dfeat_next):
            """The backward of stage s; returns the gradient w.r.t. the stage's input features on the supervised frames.  A function of its own so that
            every buffer of the stage - stored gates, dgates, state gradients - dies with its locals BEFORE the next (earlier) stage allocates
            (a loop body's rebinding `Gd = {...}` builds the new buffers while the old are alive: 30 GiB of overlap at config 4's peak)."""
            # Weight gradients feed nothing but the optimizer: every one of this stage's runs ASIDE (ops.aside: the helper stream, behind the
            # launch that produces its last operand), off the chain tail -> upsampler -> refine -> BPTT that the next stage waits for.
            # `hold`: what those launches read - alive until the next stage has rejoined the helper (which it does before it rewrites the
            # one buffer that outlives a stage, the refine block's gradient planes)
            ops.rejoin()
            hold = []
            st = ctx.stages[s]
            feat, Sb, Ys = st['feat'], st['Sb'], st['Ys']
            HF, HB = st['forward']['H'][-1], st['backward']['H'][-1]
            hold += [Sb, Ys, st.get('R1')]
            # ---- upsampler backward (3 branches x T frames at once) -----------------------------------------
            sH, sW = dO_all.shape[3], dO_all.shape[4]
            dO = dO_all[s].view(3 * TN, sH, sW, cfg.out_channels)
            # tail = last PixelShuffle conv + final conv: collapsed backward (csrc/uptail.hip) - the r*r*C-channel
            # gradient between them is never formed
            ut = P.up[-1]
            rt = ut['r']
            xin = Ys[len(P.up) - 2] if len(P.up) > 1 else Sb
            h_in, w_in = xin.shape[1], xin.shape[2]
            w2, b2, w3 = params[ut['wgrad'].wkey], params[ut['wgrad'].bkey], params[P.last_w]
            G = ops.uptail_compose(w2, w3, rt)
            a2 = acc(ut['wgrad'].wkey)
            acc(ut['wgrad'].bkey)
            a3 = acc(P.last_w)
            acc(P.last_b)
            M, Sd = ops.empty(P.tail_m.Cout, C, 3, 3), ops.empty(P.tail_m.Cout)
            if ops.uptail_xcorr_supported(C, rt, cfg.out_channels):
                D = None
            else:
                D = ops.uptail_expand(dO, rt, P.tail_dc)
            hold += [M, Sd, D, dO]
            with ops.aside('tail_w'):                    # the tail's weight gradients: cross-correlation (or the expanded GEMM) + contraction
                if D is None:
                    ops.uptail_xcorr(xin, dO, rt, out=(M, Sd))
                else:
                    ops.wgrad(P.tail_m, [Src(xin)], [Src(D)], 3 * TN, h_in, w_in, M, Sd, accumulate=False)
                ops.uptail_wcontract(M, Sd, w2, b2, w3, grads[ut['wgrad'].wkey], grads[ut['wgrad'].bkey], grads[P.last_w],
                                     grads[P.last_b], rt, a2, a3)
            dcur = ops.uptail_dgrad(dO, G, C, rt, dtype=act) if ctx.tail_bf16 else ops.uptail_dgrad(dO, G, C, rt)
            for ui in range(len(P.up) - 2, -1, -1):
                u = P.up[ui]
                r = u['r']
                xin = Ys[ui - 1] if ui > 0 else Sb
                h_in, w_in = xin.shape[1], xin.shape[2]
                ysrcs = [Src(dcur, scale=r, sub=(ij // r, ij % r)) for ij in range(r * r)]
                a = acc(u['wgrad'].wkey)
                acc(u['wgrad'].bkey)
                hold.append(dcur)
                vup = (st.get('Vup') or [None] * len(P.up))[ui] if fm.wgrad_v else None      # this convolution's input as the forward transformed it
                with ops.aside('up_w'):
                    ops.wgrad(u['wgrad'], [Src(xin)], ysrcs, 3 * TN, h_in, w_in, grads[u['wgrad'].wkey], grads[u['wgrad'].bkey],
                              accumulate=a, vsrcs=[(vup.view(1, -1), 0, 1)] if vup is not None else None, vN=3 * TN)
                dnext = ops.empty(3 * TN, h_in, w_in, C, dtype=act)
                ops.conv(u['dgrad'], ysrcs, 3 * TN, h_in, w_in, dsts=[Dst(dnext, C)])
                dcur = dnext
            dS = dcur
            hold.append(dS)
            dHf, dHb, dR = dS[0:TN], dS[TN:2 * TN], dS[2 * TN:3 * TN]
            dfeat = ops.empty(TN, H, W, C, dtype=act)
            ops.add(dfeat, dHf, dHb, dR)
            if dfeat_next is not None:                # feat[s+1] = feat[s] + R[s] on the supervised frames
                ops.add(dfeat, dfeat_next, accumulate=True)
                ops.add(dR, dfeat_next, accumulate=True)
            st['Sb'] = st['Ys'] = None
            self._mem(f'bwd stage {s}: upsampler done')
            self._sum(f'bwd {s} dS', dS)
            self._sum(f'bwd {s} dfeat0', dfeat)

            # ---- refine block backward on the T supervised windows --------------------------------------------
            xs = []
            for j in range(w):
                xs += [HF.src(U - hw + j), HB.src(U - hw + j)]
                if P.pos:
                    xs.append(Src(ctx.P4, img_off=(U - hw + j) * N))
            k1, b1 = P.r1_wgrad.wkey, P.r1_wgrad.bkey
            if P.pos:
                # the T middle frames are written by conv2's data gradient, the window halo on both sides stays zero
                dR1p = ops.halo_buffer('dR1p', ((T + 2 * hw) * N, H, W, P.C1p), act, hw * N, (hw + T) * N)
                if P.r2_wino:
                    if fm.refine2_dgrad44:
                        VdR = ops.wino44_v(TN, H, W, Cl)[0]
                        ops.wino44_transform(Src(dR), TN, H, W, VdR)
                        ops.wino44_conv(P.r2_dgrad_h, [(VdR, 0)], TN, H, W, Dst(dR1p, 2 * Cl, img_off=hw * N))
                        hold.append(VdR)
                    else:
                        ops.conv(P.r2_dgrad_h, [Src(dR)], TN, H, W, dsts=[Dst(dR1p, 2 * Cl, img_off=hw * N)])
                    # (the one real column of the rest: a 9-tap, Cl-channel stencil, HBM-bound - not a 64-column GEMM launch)
                    ops.conv_to_column(dR, params[P.r2_fwd.wkey], 2 * Cl, dR1p[hw * N:(hw + T) * N], 2 * Cl, yzero=P.C1p - P.C1)
                elif P.r1_split:
                    ops.conv(P.r2_dgrad_a, [Src(dR)], TN, H, W, dsts=[Dst(dR1p, 2 * Cl, img_off=hw * N)])
                    ops.conv(P.r2_dgrad_b, [Src(dR)], TN, H, W, dsts=[Dst(dR1p, P.C1p - 2 * Cl, c0=2 * Cl, img_off=hw * N)])
                else:
                    ops.conv(P.r2_dgrad, [Src(dR)], TN, H, W, dsts=[Dst(dR1p, P.C1p, img_off=hw * N)])
                a = acc(P.r2_wgrad.wkey)
                acc(P.r2_wgrad.bkey)
                a1 = acc(k1)
                acc(b1)
                ysrc = [Src(dR1p, nch=P.r1_cols, img_off=hw * N)]
                E = dbx = None
                if P.xcol_m:                                 # (the helper stream allocates nothing: its operands are made here)
                    E = ops.xcol_gather_m(dR1p[hw * N:(hw + T) * N], N, w, 2 * Cl, act)
                    dbx = ops.zeros(8)                       # (zeros: the launch below may run in accumulate mode)
                    hold += [E, dbx]
                with ops.aside('refine_w'):                  # the refine block's weight gradients (behind conv2's data gradient above)
                    if P.r2_wino and os.environ.get('RNH_R2_WGRAD_SPLIT', '1') != '0':
                        vr1 = st.get('VR1') if fm.wgrad_v else None          # (R1's 2 Cl channels as conv2's forward transformed them)
                        ops.wgrad(P.r2_wgrad_h, [Src(st['R1'], nch=2 * Cl)], [Src(dR)], TN, H, W, grads[P.r2_wgrad.wkey], grads[P.r2_wgrad.bkey],
                                  accumulate=a, vsrcs=[(vr1, 0, 1, 2 * Cl, 0)] if vr1 is not None else None, vN=TN)
                        ops.wgrad(P.r2_wgrad_x, [Src(st['R1'], c0=2 * Cl, nch=P.C1p - 2 * Cl)], [Src(dR)], TN, H, W, grads[P.r2_wgrad.wkey], None,
                                  accumulate=a)                     # (its own rows of the gradient: the same store / accumulate mode)
                    else:
                        ops.wgrad(P.r2_wgrad, [Src(st['R1'])], [Src(dR)], TN, H, W, grads[P.r2_wgrad.wkey],
                                  grads[P.r2_wgrad.bkey], accumulate=a)
                    a = a1
                    if P.xcol_m:
                        # 2*Cl columns against the window sources; the last channel's weights as the gradient of the per-frame convolution
                        # (the plan's row order: hidden-state sources first, then the phase planes)
                        ops.wgrad(P.r1_wgrad_a, [sc for i, sc in enumerate(xs) if i % 3 != 2] + [sc for i, sc in enumerate(xs) if i % 3 == 2],
                                  [Src(dR1p, nch=2 * Cl, img_off=hw * N)], TN, H, W, grads[k1], grads[b1], accumulate=a)
                        f0, nfr = (U - hw) * N, T + w - 1
                        ops.wgrad(P.r1x_wgrad, [HF.src(U - hw), HB.src(U - hw), Src(ctx.P4, img_off=f0)], [Src(E)], nfr * N, H, W,
                                  grads[k1][P.C1 - 1].view(w, P.C1, 3, 3), dbx[:w], accumulate=a)
                        ops.put_scalar(grads[b1][P.C1 - 1:P.C1], dbx[0:1], a)
                    elif P.r1_wino:
                        # hidden-state rows in Winograd form; the five phase-plane rows through the pixel-contraction kernel
                        v44 = st.get('V44') if fm.refine1_wgrad_v else None
                        vs_ = None
                        if v44 is not None:                  # (the top layer's transformed h' of both directions, rows in frame order)
                            vs_ = [(v44['VH'][d][-1], v44['row'](d, Lr - 1, U - hw + j), 1) for j in range(w) for d in ('forward', 'backward')]
                        ops.wgrad(P.r1_wgrad_h, [sc for sc in xs if sc.t is not ctx.P4], ysrc, TN, H, W, grads[k1], grads[b1], accumulate=a, vsrcs=vs_, vN=N)
                        if Cl % 64 == 0 and (P.r1_cols // 4) in (8, 16, 32, 64):
                            # the five phase-plane rows from border-class sums of the gradient (rnh_phase_wgrad)
                            lo, hi = (U - hw) * N, (U - hw + T + w - 1) * N
                            ops.refine_phase_wgrad(dR1p[hw * N:(hw + T) * N], ctx.P4[lo:hi], grads[k1], N, w, Cl, P.r1_cols, a)
                        else:
                            ops.wgrad(P.r1_wgrad_p, [sc for sc in xs if sc.t is ctx.P4], ysrc, TN, H, W, grads[k1], None, accumulate=a)
                    else:
                        ops.wgrad(P.r1_wgrad, xs, ysrc, TN, H, W, grads[k1], grads[b1], accumulate=a)
                    if P.xcol:
                        # the last channel's rows of the same two gradient tensors (rows the launches above do not map: they write disjoint
                        # elements), on the helper stream behind them - its operands (HF, HB, P4, dR1p) are held until the next rejoin
                        lo, hi = (U - hw) * N, (U - hw + T + w - 1) * N
                        ops.refine_xcol_wgrad([HF.frames(U - hw, U - hw + T + w - 1), HB.frames(U - hw, U - hw + T + w - 1), ctx.P4[lo:hi]],
                                              dR1p[hw * N:(hw + T) * N], grads[k1], grads[b1], N, w, Cl, a)
                gsrc = dR1p
                st['R1'] = None
            else:
                gsrc = ops.zeros((T + 2 * hw) * N, H, W, Cl, dtype=act)
                ops.add(gsrc[hw * N:(hw + T) * N], dR)
                a = acc(k1)
                acc(b1)
                with ops.aside('refine_w'):
                    ops.wgrad(P.r1_wgrad, xs, [Src(dR)], TN, H, W, grads[k1], grads[b1], accumulate=a)
            # data gradient in gather form: frame f collects from the windows f+hw-j that used it in slot j
            if P.r1_wino:
                nm = P.r1_cols
                if fm.refine_dgrad44:
                    # F(4x4, 3x3) form (rnh_wino44_conv): ONE transform of the zero-padded dR1, the window slots = the same image a frame apart
                    nfr, mtf = T + 2 * hw, N * (H // 4) * (W // 4) // 32
                    Vg = ops.wino44_v(nfr * N, H, W, nm)[0]
                    ops.wino44_transform(Src(gsrc, nch=nm), nfr * N, H, W, Vg)
                    ops.wino44_conv(P.r1_dgrad_h, [(Vg, (2 * hw - j) * mtf) for j in range(w)], TN, H, W,
                                    [Dst(dHf, Cl, accumulate=True), Dst(dHb, Cl, accumulate=True)])
                    del Vg
                else:
                    ops.conv(P.r1_dgrad_h, [Src(gsrc, nch=nm, img_off=(2 * hw - j) * N) for j in range(w)], TN, H, W,
                             dsts=[Dst(dHf, Cl, accumulate=True), Dst(dHb, Cl, accumulate=True)])
                if Cl % 64 == 0:
                    ops.refine_xcol_dgrad(gsrc, params[k1], dHf, dHb, N, w, Cl)      # the last channel's contribution: a 45-tap stencil
                else:
                    ops.conv(P.r1_dgrad_x, [Src(gsrc, c0=nm, nch=P.C1p - nm, img_off=(2 * hw - j) * N) for j in range(w)], TN, H, W,
                             dsts=[Dst(dHf, Cl, accumulate=True), Dst(dHb, Cl, accumulate=True)])
            else:
                ops.conv(P.r1_dgrad, [Src(gsrc, img_off=(2 * hw - j) * N) for j in range(w)], TN, H, W,
                         dsts=[Dst(dHf, Cl, accumulate=True), Dst(dHb, Cl, accumulate=True)])

            self._mem(f'bwd stage {s}: refine done')
            self._sum(f'bwd {s} gsrc', gsrc)
            self._sum(f'bwd {s} dHf', dHf)
            self._sum(f'bwd {s} dHb', dHb)
            # ---- ConvLSTM back-propagation through time over the supervised frames ----------------------------
            # Same wavefront as the forward, reversed: cell (d, l, k) needs the input gradient of (d, l+1, k) (an
            # event) and the state gradients of its own next-processed frame (stream order).  Buffers that cross
            # streams are allocated here, before the fork.
            dirs = ('forward', 'backward')
            tops = {'forward': dHf, 'backward': dHb}
            Gd = {d: [ops.empty(TN, H, W, 4 * hd, dtype=act) for hd in nf] for d in dirs}
            DX = {d: [ops.empty(TN, H, W, P.lstm[(d, l)]['cx'], dtype=act) if l > 0 else None for l in range(Lr)] for d in dirs}
            dfeat_d = {d: ops.empty(TN, H, W, C, dtype=act) for d in dirs}          # layer-0 input gradients per direction
            # state gradients handed from a frame to the previous one of the same (direction, layer): two buffers each, used in
            # turn, allocated HERE on the main stream - an allocation inside a side-stream block would, under HIP-graph
            # capture, come from the graph's pool on a stream other than the capture's origin (the capture then fails)
            fused = fm.cell_dgrad_fused
            DCP = {d: [[ops.empty(N, H, W, hd) for _ in range(2)] for hd in nf] for d in dirs}
            DHP = {d: [[ops.empty(N, H, W, hd, dtype=act) for _ in range(2)] for hd in nf] for d in dirs} if cfg.memory and not fused else None
            TMP = None if cfg.memory else {d: [ops.empty(N, H, W, P.lstm[(d, l)]['cx'], dtype=act) for l in range(Lr)] for d in dirs}
            dh_next = {d: [None] * Lr for d in dirs}
            dc_next = {d: [None] * Lr for d in dirs}
            # gate recomputation (stages below ctx.recompute: the forward stored no gates): the cell's forward launch runs again, on the cell's own
            # stream right in front of the launch that consumes the gates, from the saved layer input, previous hidden and cell state
            # (all inside the stores' kept ranges) - the same kernel on the same operands, so the gates and therefore every gradient
            # are bit-identical to the stored-gates step; its h' / c' go to scratch
            RG = {d: [(ops.empty(N, H, W, 4 * hd, dtype=act), ops.empty(N, H, W, hd, dtype=act), ops.empty(N, H, W, hd)) for hd in nf]
                  for d in dirs} if st['forward']['G'] is None else None

            def gates_of(d, l, k):
                """The saved gates of cell (d, l) at supervised frame k (call inside the cell's ops.side block)."""
                sd = st[d]
                if sd['G'] is not None:
                    return sd['G'][l][(k - U) * N:(k - U + 1) * N]
                kp = k - (1 if d == 'forward' else -1)
                pl = P.lstm[(d, l)]
                xin = feat if l == 0 else sd['H'][l - 1]
                g, hs, cs = RG[d][l]
                lstm = dict(hd=pl['hd'], c_prev=sd['C'][l].view(kp), h_out=hs, c_out=cs, gates_out=g)
                if st.get('use44'):                               # (the form the forward ran: the same kernels on the same operands)
                    vx, vh = RV[d][l]
                    ops.wino44_transform(xin.src(k), N, H, W, vx)
                    if cfg.memory:
                        ops.wino44_transform(sd['H'][l].src(kp), N, H, W, vh)
                    ops.wino44_cell(pl['full'], [vx, vh if cfg.memory else vx], N, H, W, lstm)
                else:
                    ops.conv(pl['full'], [xin.src(k), sd['H'][l].src(kp) if cfg.memory else xin.src(k)], N, H, W, lstm=lstm)
                return g
            RV = {d: [(ops.wino44_v(N, H, W, P.lstm[(d, l)]['cx'])[0], ops.wino44_v(N, H, W, hd)[0]) for l, hd in enumerate(nf)]
                  for d in dirs} if RG is not None and st.get('use44') else None
            pair = fm.paired                                    # (as the forward: small images)
            ops.fork(2 * Lr, bank=1)
            if fused:
                # The gate backward of a frame rides in the epilogue of the data-gradient launch of the frame its chain processed just before
                # (conv(..., lstm_bwd=...): the recurrent state gradient never reaches memory, one launch per cell and frame instead of
                # two, and the HBM-bound gate math of some workgroups overlaps the MFMAs of others - as separate launches the two
                # kinds of kernels excluded each other from the CUs and alternated in lockstep across the streams).  That launch
                # needs the input gradient of the layer above for the NEXT frame of the chain, so layer l runs one frame behind layer
                # l + 1: a skewed wavefront over tau = frame index + (Lr - 1 - l).  Only the first frame of a chain still has a launch of
                # its own for the gate backward.
                evs = {}

                def bwd_cell(d, l, idx):
                    """Cell (d, l) at chain position idx: launches the chain head's own gate backward (idx == 0) on the current stream and returns
                    the conv() arguments of the data-gradient launch (with the next frame's gate backward in its epilogue)."""
                    step = 1 if d == 'forward' else -1
                    sd, top = st[d], tops[d]
                    Cb = sd['C']
                    k = U + T - 1 - idx if d == 'forward' else U + idx
                    fi, k2 = k - U, k - step
                    fi2, has_next = k2 - U, idx + 1 < T
                    pl = P.lstm[(d, l)]
                    hd, cx = pl['hd'], pl['cx']
                    dh_of = (lambda f: top[f * N:(f + 1) * N]) if l == Lr - 1 else (lambda f: DX[d][l + 1][f * N:(f + 1) * N])
                    c_at = lambda kk: Cb[l].view(kk) if 0 <= kk < F else None               # noqa: E731
                    if idx == 0:                                # head of the chain
                        ops.lstm_gates_bwd(dh_of(fi), None, gates_of(d, l, k), c_at(k2), c_at(k),
                                           Gd[d][l][fi * N:(fi + 1) * N], DCP[d][l][0] if has_next else None)
                    dxbuf = (DX[d][l] if l > 0 else dfeat_d[d])[fi * N:(fi + 1) * N]
                    bw = None
                    if has_next:
                        bw = dict(dh=dh_of(fi2), dc_next=DCP[d][l][idx & 1], gates=gates_of(d, l, k2), c_prev=c_at(k2 - step),
                                  c_next=c_at(k2), dgates=Gd[d][l][fi2 * N:(fi2 + 1) * N],
                                  dc_prev=DCP[d][l][(idx + 1) & 1] if idx + 2 < T else None, hd=hd, rec_dtype=act)
                    return pl['dgrad'], [Src(Gd[d][l][fi * N:(fi + 1) * N])], N, H, W, dict(dsts=[Dst(dxbuf, cx)], lstm_bwd=bw)

                for tau in range(T + Lr - 1):
                    for l in range(Lr - 1, -1, -1):
                        idx = tau - (Lr - 1 - l)
                        if not 0 <= idx < T:
                            continue
                        if pair:
                            # the two directions' launches of (layer, chain position) in one (ops.conv_pair), one stream per layer: what this
                            # launch reads from the layer above (its input gradients of frames idx and idx + 1, both directions) is waited for first
                            with ops.side(l):
                                if l < Lr - 1:
                                    ops.wait(evs[(l + 1, min(idx + 1, T - 1))])
                                ops.conv_pair([bwd_cell(d, l, idx) for d in dirs])
                                if l > 0:
                                    evs[(l, idx)] = ops.record()
                            continue
                        for di, d in enumerate(dirs):
                            with ops.side(di * Lr + l):
                                if l < Lr - 1:
                                    ops.wait(evs[(d, l + 1, min(idx + 1, T - 1))])
                                plan, srcs, _, _, _, kw = bwd_cell(d, l, idx)
                                ops.conv(plan, srcs, N, H, W, **kw)
                                if l > 0:
                                    evs[(d, l, idx)] = ops.record()
            def bwd_cell_unfused(d, l, idx, dx_above):
                """Separate launches (the fp32 path): the gate backward of cell (d, l) at chain position idx on the current stream; returns the
                conv() arguments of its data gradient and the buffer the layer below reads (None at the bottom)."""
                step = 1 if d == 'forward' else -1
                k = U + T - 1 - idx if d == 'forward' else U + idx
                sd, top = st[d], tops[d]
                Cb = sd['C']
                fi = k - U
                prevk = k - step
                prev_grad = U <= prevk < U + T
                pl = P.lstm[(d, l)]
                hd, cx = pl['hd'], pl['cx']
                dh = top[fi * N:(fi + 1) * N] if l == Lr - 1 else dx_above
                c_prev = Cb[l].view(prevk) if 0 <= prevk < F else None
                dg = Gd[d][l][fi * N:(fi + 1) * N]
                dcp = DCP[d][l][idx & 1] if prev_grad else None
                if fm.gates_bwd44:                       # (the gate backward writes the transformed gate gradients the F(4x4) data gradient reads)
                    ops.wino44_gates_bwd(dh, dc_next[d][l], gates_of(d, l, k), c_prev, Cb[l].view(k), dg, dcp, dh_next[d][l], VG[d][l])
                else:
                    ops.lstm_gates_bwd(dh, dc_next[d][l], gates_of(d, l, k), c_prev, Cb[l].view(k), dg, dcp, dh2=dh_next[d][l])
                dxbuf = (DX[d][l] if l > 0 else dfeat_d[d])[fi * N:(fi + 1) * N]
                dhp, tmp = None, None
                if cfg.memory:
                    dsts = [Dst(dxbuf, cx)]
                    if prev_grad:
                        dhp = DHP[d][l][idx & 1]
                        dsts.append(Dst(dhp, hd))
                else:
                    tmp = TMP[d][l]
                    dsts = [Dst(dxbuf, cx), Dst(tmp, cx)]
                dh_next[d][l], dc_next[d][l] = dhp, dcp
                return (pl['dgrad'], [Src(dg)], N, H, W, dict(dsts=dsts)), dxbuf, tmp

            # the fp32 path's data gradients in F(4x4, 3x3) form where the forward's cells ran in it: one transform of the frame's gate gradients
            # (4 hd channels) into a scratch image per (direction, layer), then rnh_wino44_conv with the transposed weights
            dg44 = fm.cell_dgrad44
            VG = {d: [ops.wino44_v(N, H, W, 4 * hd)[0] for hd in nf] for d in dirs} if dg44 else None

            def dgrad_launch(d, l, call):
                plan, srcs, _, _, _, kw = call
                if dg44:
                    if not fm.gates_bwd44:
                        ops.wino44_transform(srcs[0], N, H, W, VG[d][l])
                    ops.wino44_conv(plan, [(VG[d][l], 0)], N, H, W, kw['dsts'])
                else:
                    ops.conv(plan, srcs, N, H, W, **kw)

            for idx in range(T if not fused else 0):
                if pair:
                    # one stream per layer; the two directions' data gradients of (layer, chain position) in one launch
                    dx_above, above = {d: None for d in dirs}, None
                    for l in range(Lr - 1, -1, -1):
                        with ops.side(l):
                            if above is not None:
                                ops.wait(above)
                            cells = [bwd_cell_unfused(d, l, idx, dx_above[d]) for d in dirs]
                            if dg44:
                                for d, c in zip(dirs, cells):
                                    dgrad_launch(d, l, c[0])
                            else:
                                ops.conv_pair([c[0] for c in cells])
                            for d, (_, dxbuf, tmp) in zip(dirs, cells):
                                if tmp is not None:
                                    ops.add(dxbuf, tmp, accumulate=True)
                                dx_above[d] = dxbuf if l > 0 else None
                            above = ops.record() if l > 0 else None
                    continue
                for di, d in enumerate(dirs):
                    dx_above, above = None, None
                    for l in range(Lr - 1, -1, -1):
                        with ops.side(di * Lr + l):
                            if above is not None:
                                ops.wait(above)
                            call, dxbuf, tmp = bwd_cell_unfused(d, l, idx, dx_above)
                            dgrad_launch(d, l, call)
                            if tmp is not None:
                                ops.add(dxbuf, tmp, accumulate=True)
                            dx_above = dxbuf if l > 0 else None
                            above = ops.record() if l > 0 else None
            # weight gradients of the cells, batched over the T frames (each on its cell's stream).  Nothing of this stage's backward waits
            # for them: the main stream joins the chains as they stand HERE (the events below) and goes on with the upsampler / refine
            # backward of the next (earlier) stage - kernels that have the chip to themselves otherwise - while the weight gradients
            # run beside them on the side streams; what they read is handed to the caller, who keeps it until the next join that
            # covers them (RNH_DEFER_WGRAD=0: join right here, as until round 3)
            defer = os.environ.get('RNH_DEFER_WGRAD', '1') != '0'
            chains_done = []
            if defer:
                for i in range(2 * Lr):
                    with ops.side(i):
                        chains_done.append(ops.record())
            for di, d in enumerate(dirs):
                step = 1 if d == 'forward' else -1
                Hb = st[d]['H']
                for l in range(Lr):
                    with ops.side(l if pair else di * Lr + l):      # (the stream the cell's chain ran on)
                        pl = P.lstm[(d, l)]
                        xin = feat if l == 0 else Hb[l - 1]
                        second = _span_src(Hb[l], U - step, T) if cfg.memory else _span_src(xin, U, T)
                        wk, bk = pl['wgrad'].wkey, pl['wgrad'].bkey
                        a = acc(wk)
                        acc(bk)
                        v44 = st.get('V44') if fm.wgrad_v else None
                        vs_ = None
                        if v44 is not None:
                            def vspan(Vt, dd, ll, k0, row=v44['row']):
                                r0 = row(dd, ll, k0)
                                return (Vt, r0, (row(dd, ll, k0 + 1) - r0) if T > 1 else 1)
                            vx = (v44['VF'], U, 1) if l == 0 else vspan(v44['VH'][d][l - 1], d, l - 1, U)
                            vs_ = [vx, vspan(v44['VH'][d][l], d, l, U - step) if cfg.memory else vx]
                        ops.wgrad(pl['wgrad'], [_span_src(xin, U, T), second], [Src(Gd[d][l])], TN, H, W, grads[wk], grads[bk],
                                  accumulate=a, vsrcs=vs_, vN=N)
            if defer:
                for ev in chains_done:
                    ops.wait(ev)
            else:
                ops.join(2 * Lr)
            self._mem(f'bwd stage {s}: BPTT done')
            self._sum(f'bwd {s} dfeat_fwd', dfeat_d['forward'])
            self._sum(f'bwd {s} dfeat_bwd', dfeat_d['backward'])
            ops.add(dfeat, dfeat_d['forward'], dfeat_d['backward'], accumulate=True)
            self._sum(f'bwd {s} dfeat', dfeat)
            for d in dirs:                                          # (the weight gradients read neither the stored gates nor the cell states)
                st[d]['G'] = st[d]['C'] = None
            in_flight = (st, feat, Gd if defer else None, hold)    # what the weight-gradient launches still read: h, features, dgates, ...
            ctx.stages[s] = None
            return dfeat, in_flight

        pending = None
        for s in range(S - 1, -1, -1):
            # the previous stage's weight gradients sit in front of this stage's chains on the same side streams: once this stage's
            # chains have been joined (inside stage_backward) they are done, and what they read can go
            dfeat_next, pending_new = stage_backward(s, dfeat_next)
            pending = pending_new
        # ---- input block backward (supervised frames only, refine_net.py:66-67), beside the first stage's weight gradients -----------
        xc = ctx.x_all[U * N:(U + T) * N]
        if self.bf16:
            dfeat_next = ops.cast(dfeat_next, self.f32)           # back across the precision boundary of the input block
        ops.inconv_bwd(xc, params['in_block.conv.weight'], params['in_block.conv.bias'], params['in_block.prelu.weight'],
                       dfeat_next, grads['in_block.conv.weight'], grads['in_block.conv.bias'], grads['in_block.prelu.weight'],
                       accumulate=False)
        ops.join(2 * Lr)                                          # the first stage's weight gradients (the last ones launched)
        ops.rejoin()
        ops.fence()                                               # (one sink for a captured graph: see HipOps.fence)
        pending = None
        return grads
