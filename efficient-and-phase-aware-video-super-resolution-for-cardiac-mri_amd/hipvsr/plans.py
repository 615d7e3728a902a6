"""Backend-independent descriptions of the convolutions of RefineNet as implicit GEMMs.

A ``ConvPlan`` says how the K dimension (source segments x 16-channel chunks x taps) and the column
dimension of one convolution map onto the reference's OIHW weight tensor (reference
src/model/nets/refine_net.py: conv definitions at :149-154, :191, :199-205, :235-241).  ``WgradPlan`` does
the same for the rows / columns of a weight gradient.  The HIP backend turns the maps into device index
arrays for ``rnh_pack_weights`` / ``rnh_wgrad_reduce``; the test double in tests/ evaluates them with
torch ops, so that the maps themselves are checked against the oracle on CPU.
"""
import os
from dataclasses import dataclass, field
from typing import List, Optional

from . import lib as L

KC = 16   # channels per K step of the forward kernel


@dataclass
class Src:
    """Python-level mirror of rnh_src_t: channels [c0, c0+nch) of an NHWC tensor (B, H, W, C)."""
    t: object
    c0: int = 0
    nch: Optional[int] = None
    img_off: int = 0
    add: object = None
    scale: int = 1
    sub: tuple = (0, 0)


@dataclass
class Dst:
    """Python-level mirror of rnh_dst_t."""
    t: object
    ncols: int
    c0: int = 0
    accumulate: bool = False
    img_off: int = 0


@dataclass
class KSeg:
    nch: int          # channels of the source tensor slice (multiple of 4)
    nvalid: int       # how many of them correspond to real weight channels (<= nch)
    kbase: int        # first index along the weight's K axis (Cin if not transposed, Cout if transposed)
    kcoff: int = 0    # offset added to the column map for this segment


def pick_tile(ncols):
    if ncols <= 64:
        return L.TILE_256x64
    if ncols <= 128:
        return L.TILE_128x128
    if ncols <= 160:
        return L.TILE_128x160
    return L.TILE_128x128


def _pad_to(n, m):
    return (n + m - 1) // m * m


class ConvPlan:
    """K / column layout of one rnh_conv_igemm call against the OIHW weight ``wkey``."""

    def __init__(self, name, wkey, bkey, wshape, ksegs: List[KSeg], colmap: List[int], tile=None,
                 epilogue=L.EPI_STORE, transposed=False, kstride=1, wino=False, bf16=False, wino_cols=None):
        self.name, self.wkey, self.bkey = name, wkey, bkey
        self.Cout, self.Cin, kh, kw = wshape
        self.ntaps = kh * kw
        assert self.ntaps in (1, 9)
        self.ksegs, self.transposed, self.kstride, self.epilogue = ksegs, transposed, kstride, epilogue
        self.tile = pick_tile(len(colmap)) if tile is None else tile
        # bf16 = True: a plan of rnh_conv_bf16 (csrc/conv_bf16.hip): same K order (source -> 16-channel chunk -> tap), weights
        # packed to bf16 in natural k order, columns padded to 64 (column tiles of 128 where that divides, else of 64)
        self.bf16 = bool(bf16)
        self.nchunks = sum((sg.nch + KC - 1) // KC for sg in ksegs)
        wino = wino and not self.bf16
        # Winograd form (rnh_conv_wino, csrc/conv_wino.hip): K in steps of 4 channels; column blocks of 64 (16-channel chunks, two
        # workgroups per CU) or of 128 (wino_cols: 32-channel chunks, one 8-wave workgroup per CU - asked for per plan)
        self.wino = bool(wino) and self.ntaps == 9 and all(sg.nch % 16 == 0 and sg.nvalid == sg.nch for sg in ksegs)
        if wino_cols is None:                                   # the wider blocks wherever they cost no padding (RNH_WINO_COLS=64: never)
            wino_cols = 128 if len(colmap) % 128 == 0 and os.environ.get('RNH_WINO_COLS', '128') != '64' else 64
        self.wino_cols = 128 if self.wino and wino_cols == 128 and all(sg.nch % 32 == 0 for sg in ksegs) else 64
        self.Npad = _pad_to(len(colmap), self.wino_cols if self.wino else 64) if self.bf16 or self.wino else \
            _pad_to(len(colmap), L.TILE_COLS[self.tile])
        self.colmap = list(colmap) + [-1] * (self.Npad - len(colmap))
        self.kbase, self.knv, self.ktap, self.kcoff = [], [], [], []
        for sg in ksegs:
            assert sg.nch % 4 == 0 and 0 < sg.nvalid <= sg.nch
            for ch in range((sg.nch + KC - 1) // KC):
                for t in range(self.ntaps):
                    self.kbase.append(sg.kbase + ch * KC * kstride)
                    self.knv.append(max(0, min(KC, sg.nvalid - ch * KC)))
                    self.ktap.append(t)
                    self.kcoff.append(sg.kcoff)
        self.nk = len(self.kbase)
        if self.wino:
            self.wkbase, self.wknv, self.wkcoff = [], [], []
            for sg in ksegs:
                for q in range(sg.nch // 4):
                    self.wkbase.append(sg.kbase + q * 4 * kstride)
                    self.wknv.append(max(0, min(4, sg.nvalid - q * 4)))
                    self.wkcoff.append(sg.kcoff)
            self.wns = len(self.wkbase)

    def __repr__(self):
        return f'ConvPlan({self.name}, nk={self.nk}, Npad={self.Npad}, tile={self.tile})'


@dataclass
class XSeg:
    nch: int
    nvalid: int
    ci_base: int


@dataclass
class YSeg:
    nch: int
    nvalid: int
    co_base: int
    stride: int = 1


def pick_wgrad_tile(nrows, ncols):
    """(MI, NI): one wave computes 32*MI rows x 32*NI columns of dW."""
    # padded area / relative efficiency of the tile (bytes loaded per MFMA, accumulators in flight)
    cands = [((4, 4), 1.15), ((4, 2), 1.0), ((2, 4), 1.0), ((2, 2), 0.92), ((4, 1), 0.85)]
    cost = lambda t: _pad_to(nrows, 32 * t[0][0]) * _pad_to(ncols, 32 * t[0][1]) / t[1]     # noqa: E731
    return min(cands, key=cost)[0]


class WgradPlan:
    """Row (forward input channel) / column (output channel) layout of one rnh_conv_wgrad call."""

    def __init__(self, name, wkey, bkey, wshape, xsegs: List[XSeg], ysegs: List[YSeg], tile=None, bf16=False):
        self.name, self.wkey, self.bkey = name, wkey, bkey
        self.Cout, self.Cin, kh, kw = wshape
        self.ntaps = kh * kw
        self.xsegs, self.ysegs = xsegs, ysegs
        nrows, ncols = sum(sg.nch for sg in xsegs), sum(sg.nch for sg in ysegs)
        self.mi, self.ni = pick_wgrad_tile(nrows, ncols) if tile is None else tile
        self.tile = (self.mi << 4) | self.ni
        import os
        if os.environ.get('RNH_WGRAD_TILE'):              # experiments only: "MI,NI,Dcode"
            mi, ni, dc = (int(v) for v in os.environ['RNH_WGRAD_TILE'].split(','))
            self.mi, self.ni, self.tile = mi, ni, (dc << 8) | (mi << 4) | ni
        rowmap, xgrp = [], []
        for si, sg in enumerate(xsegs):
            assert sg.nch % 4 == 0
            rowmap += [sg.ci_base + c if c < sg.nvalid else -1 for c in range(sg.nch)]
            xgrp += [(si << 16) | c for c in range(0, sg.nch, self.mi)]          # one entry per lane slot
        colmap, ygrp = [], []
        for si, sg in enumerate(ysegs):
            assert sg.nch % 4 == 0
            colmap += [sg.co_base + c * sg.stride if c < sg.nvalid else -1 for c in range(sg.nch)]
            ygrp += [(si << 16) | c for c in range(0, sg.nch, self.ni)]
        self.xcols_pad = _pad_to(len(rowmap), 32 * self.mi)
        self.ycols_pad = _pad_to(len(colmap), 32 * self.ni)
        self.rowmap = rowmap + [-1] * (self.xcols_pad - len(rowmap))
        self.colmap = colmap + [-1] * (self.ycols_pad - len(colmap))
        self.xgrp = xgrp + [-1] * (self.xcols_pad // self.mi - len(xgrp))
        self.ygrp = ygrp + [-1] * (self.ycols_pad // self.ni - len(ygrp))
        # rnh_wgrad_bf16 (csrc/wgrad_bf16.hip): workgroup tiles of 64 rows x 64 columns
        self.bf16 = bool(bf16)
        self.xrows_pad64, self.ycols_pad64 = _pad_to(len(rowmap), 64), _pad_to(len(colmap), 64)
        self.rowmap64 = rowmap + [-1] * (self.xrows_pad64 - len(rowmap))
        self.colmap64 = colmap + [-1] * (self.ycols_pad64 - len(colmap))

    def nsplit_bf16(self, nitems):
        """Pixel-range splits of rnh_wgrad_bf16: tiles x splits = about two workgroups per CU (the kernel keeps two resident:
        one's staging latency hides behind the other's MFMAs)."""
        tiles = (self.xrows_pad64 // 64) * (self.ycols_pad64 // 64)
        # floor, not ceil: 22 tiles x 24 splits = 528 workgroups on 512 slots ran as two rounds, the second one nearly empty (refine conv1's
        # weight gradient 2.03 -> see profiles/ARCHIVE/r04_k; the ConvLSTM's 8 tiles x 64 were exact already)
        return int(max(1, min(nitems, 256, 512 // tiles)))

    def nsplit(self, npix):
        import os
        if os.environ.get('RNH_WGRAD_NSPLIT'):            # experiments only
            return int(os.environ['RNH_WGRAD_NSPLIT'])
        items = (self.xcols_pad // (32 * self.mi)) * (self.ycols_pad // (32 * self.ni)) * self.ntaps
        # Every wave does the same amount of work, so the launch should fill the chip exactly once: 256 CUs x the
        # workgroups a CU holds (2 for the 128-accumulator tiles, 3 for the 64x64 one) and no partial second round.
        # resident waves on the chip: 1024 SIMDs x (1 for the 256-accumulator tile, 2 for 128, 3 for 64 accumulators)
        acc = self.mi * self.ni
        resident = 1024 * (1 if acc >= 16 else 3 if (self.mi, self.ni) == (2, 2) else 2)
        n = int(max(1, min(resident // items, 256, (npix + 63) // 64)))
        # the kernel packs whole ranges per XCD when their number is a multiple of 8
        return n // 8 * 8 if n >= 8 else n


# --------------------------------------------------------------------------------------------------------
# the plans of one RefineNet configuration
# --------------------------------------------------------------------------------------------------------
def lstm_colmap(hd):
    """Column n = tile*128 + gate*32 + j  <->  reference output channel gate*hd + tile*32 + j."""
    cm = []
    for tl in range((hd + 31) // 32):
        for g in range(4):
            for j in range(32):
                hc = tl * 32 + j
                cm.append(g * hd + hc if hc < hd else -1)
    return cm


def lstm_colmap64(hd):
    """64-column gate groups: column n = group*64 + tile*32 + half*16 + c  <->  gate (2*tile + half), channel group*16 + c."""
    cm = []
    for grp in range((hd + 15) // 16):
        for tile in range(2):
            for half in range(2):
                for c in range(16):
                    hc = grp * 16 + c
                    cm.append((2 * tile + half) * hd + hc if hc < hd else -1)
    return cm


def ps_colmap(cq, r):
    """Column n = (i*r + j)*cq + c  <->  nn.PixelShuffle input channel c*r*r + i*r + j."""
    return [(n % cq) * r * r + n // cq for n in range(cq * r * r)]


def r4(n):
    return _pad_to(n, 4)


class NetPlans:
    """All plans of one RefineNet configuration (``cfg`` has the reference constructor kwargs as attributes)."""

    def __init__(self, cfg, bf16=False):
        """bf16 = True: the plans of the bf16-storage path (rnh_conv_bf16 / rnh_wgrad_bf16): no Winograd forms, no side
        paths (refine conv1 runs with all 2*Cl + 1 columns and the phase planes as 8-channel K sources), channel counts
        in multiples of 8."""
        import functools
        self.cfg = cfg
        self.bf16 = bf = bool(bf16)
        ConvPlan_ = functools.partial(ConvPlan, bf16=bf)
        WgradPlan_ = functools.partial(WgradPlan, bf16=bf)
        nf = list(cfg.num_features)
        self.nf, self.C, self.Cl, self.L = nf, nf[0], nf[-1], len(nf)
        for c in nf:
            if c % (8 if bf else 4):
                raise ValueError(f'num_features must be multiples of {8 if bf else 4} for the HIP path')
        # rnh_inconv_prelu_bwd (four output channels per thread, include/refinenet_hip.h) serves 4, 8, 16, 32, 64, 128 or 256 features; the forward
        # kernel serves every multiple of 4.  A constraint of the BACKWARD only: a net of another width is planned (inference, the CPU double) and
        # refused by the first forward that is asked to keep what a backward needs (hipvsr.engine.RefineNetEngine.forward, need_grad=True)
        self.inconv_bwd_error = None
        if nf[0] > 256 or 256 % (nf[0] // 4):
            self.inconv_bwd_error = (f'num_features[0] = {nf[0]}: the backward of the HIP path\'s input block takes 4, 8, 16, 32, 64, 128 or 256 '
                                     f'features (inference runs at every multiple of 4)')
        pw = 8 if bf else 4                       # channels of a phase plane (p, 0, ..., 0)
        self.pw = pw
        self.lstm = {}
        for d in ('forward', 'backward'):
            for l, hd in enumerate(nf):
                cx = nf[0] if l == 0 else nf[l - 1]
                cin = cx + hd if cfg.memory else 2 * cx
                wk, bk = f'{d}_lstm_block.cell_list.{l}.conv.weight', f'{d}_lstm_block.cell_list.{l}.conv.bias'
                ws = (4 * hd, cin, 3, 3)
                second = hd if cfg.memory else cx
                ltile = int(os.environ.get('RNH_LSTM_TILE', L.TILE_128x128_G))       # experiments: 0 = 128x128, 2 = 256x64
                wino = os.environ.get('RNH_WINO', '1') != '0' and ltile == L.TILE_128x128_G and not bf
                # the cell as 128-column blocks (the four gates of 32 hidden channels: gate layout lstm_colmap) where cx, hd % 32 == 0
                wcols = 128 if wino and os.environ.get('RNH_WINO_COLS', '128') != '64' and cx % 32 == 0 and second % 32 == 0 and hd % 32 == 0 else 64
                lcm = lstm_colmap64(hd) if ltile in (L.TILE_128x128, L.TILE_256x64) or (wino and wcols == 64) else lstm_colmap(hd)
                def mk(lcm_, wino_):
                    full_ = ConvPlan_(f'{d}{l}.fwd', wk, bk, ws, [KSeg(cx, cx, 0), KSeg(second, second, cx)], lcm_,
                                     tile=ltile, epilogue=L.EPI_LSTM, wino=wino_, wino_cols=wcols)
                    first_ = ConvPlan_(f'{d}{l}.fwd0', wk, bk, ws, [KSeg(cx, cx, 0)], lcm_, tile=ltile,
                                      epilogue=L.EPI_LSTM, wino=wino_, wino_cols=wcols) if cfg.memory else full_
                    return full_, first_
                full, first = mk(lcm, wino)
                if wino and not (full.wino and first.wino):          # not eligible: the implicit-GEMM kernel and its gate layout
                    wino, lcm = False, lstm_colmap(hd)
                    full, first = mk(lcm, False)
                for pl_ in (full, first):                           # hidden channels per column group of the gate layout
                    pl_.gate_group = 16 if len(lcm) == 64 * ((hd + 15) // 16) and lcm == lstm_colmap64(hd) else 32
                    # eligible for the F(4x4, 3x3) form of the cell (rnh_wino44_cell: csrc/conv_wino44.hip)?  fp32, whole 16-channel chunks,
                    # an even number of them, hidden channels in sixteens; HipOps.wino44_ok decides per call (image size, launch size)
                    pl_.wino44 = (not bf and wino and hd % 16 == 0 and all(sg.nch % 16 == 0 and sg.nvalid == sg.nch for sg in pl_.ksegs)
                                  and sum(sg.nch for sg in pl_.ksegs) % 32 == 0 and os.environ.get('RNH_WINO44', '1') != '0')
                dgrad = ConvPlan_(f'{d}{l}.dgrad', wk, None, ws, [KSeg(4 * hd, 4 * hd, 0)], list(range(cin)), transposed=True,
                                 wino=os.environ.get('RNH_WINO_DGRAD', '1') != '0' and wino)
                # the data gradient in F(4x4, 3x3) form (rnh_wino44_conv on the transformed gate gradients): 177 against 281-303 us at BASELINE config 2.
                # With a transform launch of its own over the 4 hd = 256 gate-gradient channels (107 us) that was barely a gain (round 5: step 265.8 ->
                # 263.6 ms, an opt-in); since round 6 the gate backward writes the transformed image itself (rnh_wino44_gates_bwd) and the form is the
                # engine's choice wherever that launch serves the shape (hipvsr/forms.py; RNH_WINO44_DGRAD=0: off, =force: also with the separate
                # transform launch); 0.3 GB of scratch per chain
                dgrad.wino44 = bool(full.wino44) and dgrad.wino and (4 * hd) % 32 == 0 and os.environ.get('RNH_WINO44_DGRAD', '1') != '0'
                wgrad = WgradPlan_(f'{d}{l}.wgrad', wk, bk, ws, [XSeg(cx, cx, 0), XSeg(second, second, cx)],
                                  [YSeg(4 * hd, 4 * hd, 0)])
                # the weight gradient in F(4x4)-tile Winograd form with both transforms fused (rnh_wino44f_wgrad, round 6) where the kernel takes the call
                # (fp32, whole 32-channel row blocks and 64-channel column blocks, H % 4 == 0, W % 16 == 0: checked per call)
                wgrad.wino44f = not bf
                self.lstm[(d, l)] = dict(full=full, first=first, dgrad=dgrad, wgrad=wgrad, cx=cx, hd=hd, second=second)

        Cl, w = self.Cl, cfg.refine_window_size
        self.pos = bool(cfg.positional_encoding)
        self.xcol = self.r1_wino = self.r1_split = self.r2_wino = self.xcol_m = False
        if self.pos:
            C1 = 2 * Cl + 1
            self.C1, self.C1p = C1, (_pad_to(C1, 8) if bf else r4(C1))
            ws1, ws2 = (C1, w * C1, 3, 3), (Cl, C1, 3, 3)
            k1, b1 = 'refine_block.body.conv1.weight', 'refine_block.body.conv1.bias'
            k2, b2 = 'refine_block.body.conv2.weight', 'refine_block.body.conv2.bias'
            segs, xsegs = [], []
            for j in range(w):
                segs += [KSeg(Cl, Cl, j * C1), KSeg(Cl, Cl, j * C1 + Cl), KSeg(pw, 1, j * C1 + 2 * Cl)]
                xsegs += [XSeg(Cl, Cl, j * C1), XSeg(Cl, Cl, j * C1 + Cl), XSeg(pw, 1, j * C1 + 2 * Cl)]
            # conv1 writes C1p channels (the pad channels have zero weights and bias => zeros).  C1 = 2*Cl + 1 is one more
            # than a whole number of 32-column MFMA tiles when Cl % 32 == 0: that channel then takes the side path
            # (HipOps.refine_xcol_fwd / _wgrad) and the GEMMs run on r1_cols = 2*Cl columns
            self.xcol = C1 % 32 == 1 and Cl % 32 == 0 and os.environ.get('RNH_XCOL', '1') != '0' and not bf
            self.r1_cols = C1 - 1 if self.xcol else self.C1p
            self.r1_fwd = ConvPlan_('refine1.fwd', k1, b1, ws1, segs,
                                   list(range(C1 - 1)) if self.xcol else list(range(C1)) + [-1] * (self.C1p - C1))
            # Winograd split of the two big refine convolutions (only together with the side path, which leaves 2*Cl
            # columns / K channels): the hidden-state sources go through rnh_conv_wino, the 4-channel phase planes (forward)
            # and the odd channel of dR1 (data gradient) through the implicit GEMM, accumulating into the same output
            self.r1_wino = self.xcol and (2 * Cl) % 128 == 0 and os.environ.get('RNH_WINO', '1') != '0' and \
                os.environ.get('RNH_WINO_REFINE', '1') != '0'
            if self.r1_wino:
                hsegs = [sg for sg in segs if sg.nch == Cl]
                psegs = [sg for sg in segs if sg.nch == 4]
                self.r1_fwd_h = ConvPlan_('refine1.fwd.h', k1, b1, ws1, hsegs, list(range(C1 - 1)), wino=True)
                self.r1_fwd_p = ConvPlan_('refine1.fwd.p', k1, None, ws1, psegs, list(range(C1 - 1)))
                # refine conv1's forward over the hidden states in F(4x4, 3x3) form (rnh_wino44_conv) on the transformed h' of the top ConvLSTM layer
                self.r1_fwd_h.wino44 = Cl % 16 == 0 and (len(hsegs) * Cl) % 32 == 0 and os.environ.get('RNH_WINO44', '1') != '0' and \
                    os.environ.get('RNH_WINO44_REFINE', '1') != '0'
                self.r1_wgrad_h = WgradPlan_('refine1.wgrad.h', k1, b1, ws1, [sg for sg in xsegs if sg.nch == Cl], [YSeg(C1 - 1, C1 - 1, 0)])
                self.r1_wgrad_h.wino44w = not bf and Cl == 64          # F(4x4)-tile form on materialised transforms (rnh_wino44_wgrad_*): opt-in, RNH_WINO44_WGRAD=1
                self.r1_wgrad_h.wino44f = not bf                        # F(4x4)-tile form, both transforms fused (rnh_wino44f_wgrad): 5.54 -> 4.53 ms at BASELINE config 2
                self.r1_wgrad_p = WgradPlan_('refine1.wgrad.p', k1, None, ws1, [sg for sg in xsegs if sg.nch == 4], [YSeg(C1 - 1, C1 - 1, 0)])
                self.r1_dgrad_h = ConvPlan_('refine1.dgrad.h', k1, None, ws1, [KSeg(C1 - 1, C1 - 1, 0, kcoff=j * C1) for j in range(w)],
                                           list(range(2 * Cl)), transposed=True, wino=True)
                # ... and its data gradient over the hidden states (one transform of the zero-padded dR1, the five window slots as five sources)
                self.r1_dgrad_h.wino44 = self.r1_fwd_h.wino44 and (C1 - 1) % 16 == 0 and (w * (C1 - 1)) % 32 == 0 and \
                    os.environ.get('RNH_WINO44_REFINE_DGRAD', '1') != '0'
                self.r1_dgrad_x = ConvPlan_('refine1.dgrad.x', k1, None, ws1,
                                           [KSeg(self.C1p - C1 + 1, 1, C1 - 1, kcoff=j * C1) for j in range(w)], list(range(2 * Cl)),
                                           transposed=True)
            # bf16 path: the 2*Cl + 1 columns of conv1 (and of conv2's data gradient) as a 2*Cl-column launch (128-column tiles
            # where 2*Cl % 128 == 0) plus an 8-column launch for the last channel, instead of one launch padded to 192
            self.r1_split = bf and (2 * Cl) % 64 == 0 and os.environ.get('RNH_R1_SPLIT', '1') != '0'
            if self.r1_split:
                tail = [C1 - 1] + [-1] * (self.C1p - C1)
                self.r1_fwd_a = ConvPlan_('refine1.fwd.a', k1, b1, ws1, segs, list(range(C1 - 1)))
                self.r1_fwd_b = ConvPlan_('refine1.fwd.b', k1, b1, ws1, segs, tail)
                self.r2_dgrad_a = ConvPlan_('refine2.dgrad.a', k2, None, ws2, [KSeg(Cl, Cl, 0)], list(range(C1 - 1)), transposed=True)
                self.r2_dgrad_b = ConvPlan_('refine2.dgrad.b', k2, None, ws2, [KSeg(Cl, Cl, 0)], tail, transposed=True)
            # round 3: instead of the 8-column launch .b (a 64-column kernel over K = 5 x 136 channels for ONE real column) the last
            # channel of conv1 goes frame by frame: slot j of that channel is a convolution of source frame k + j that does not depend
            # on the window, so ONE small convolution over the F' source frames with the J = w slots as columns (weight = the VIEW
            # w1[2*Cl].view(w, C1, 3, 3), key r1x_key) and a sum over the slots replace it - 1/4.4 of the MFMA work - and the weight
            # gradient of that channel is one small rnh_wgrad_bf16 into the same view of the gradient (the main one keeps 2*Cl columns)
            self.xcol_m = self.r1_split and os.environ.get('RNH_XCOL_M', '1') != '0'
            if self.xcol_m:
                self.r1x_key = k1 + '[last channel as (w, C1, 3, 3)]'
                fsegs = [KSeg(Cl, Cl, 0), KSeg(Cl, Cl, Cl), KSeg(pw, 1, 2 * Cl)]
                self.r1x_fwd = ConvPlan_('refine1.fwd.x', self.r1x_key, None, (w, C1, 3, 3), fsegs, list(range(w)))
                self.r1x_wgrad = WgradPlan_('refine1.wgrad.x', self.r1x_key, None, (w, C1, 3, 3),
                                           [XSeg(Cl, Cl, 0), XSeg(Cl, Cl, Cl), XSeg(pw, 1, 2 * Cl)], [YSeg(8, w, 0)])
                # rows: the 2*w hidden-state sources first, the w phase planes behind them (NOT slot by slot: rnh_wgrad_bf16 works on 64-row
                # tiles, and an 8-channel plane in front of a hidden-state source shifts it off the tile grid - every 128-byte pixel
                # line of that source is then fetched by two row tiles: 2.10 -> see profiles/ARCHIVE/r04_k for the launch at config 2)
                self.r1_wgrad_a = WgradPlan_('refine1.wgrad.a', k1, b1, ws1,
                                            [sg for i, sg in enumerate(xsegs) if i % 3 != 2] + [sg for i, sg in enumerate(xsegs) if i % 3 == 2],
                                            [YSeg(C1 - 1, C1 - 1, 0)])
            # conv2 (C1 -> Cl channels) the same way: its 2*Cl hidden-state input channels in Winograd form, the phase channel (and
            # the pad channels behind it) through the implicit GEMM, accumulating; its data gradient as a 2*Cl-column Winograd
            # launch plus a launch for the columns of the last channel
            self.r2_wino = self.r1_wino and os.environ.get('RNH_WINO_REFINE2', '1') != '0'
            if self.r2_wino:
                self.r2_fwd_h = ConvPlan_('refine2.fwd.h', k2, b2, ws2, [KSeg(2 * Cl, 2 * Cl, 0)], list(range(Cl)), wino=True)
                self.r2_fwd_x = ConvPlan_('refine2.fwd.x', k2, None, ws2, [KSeg(self.C1p - 2 * Cl, C1 - 2 * Cl, 2 * Cl)], list(range(Cl)))
                self.r2_dgrad_h = ConvPlan_('refine2.dgrad.h', k2, None, ws2, [KSeg(Cl, Cl, 0)], list(range(2 * Cl)), transposed=True, wino=True)
                # conv2's hidden-state part and its data gradient in F(4x4, 3x3) form (rnh_wino44_conv; one transform of R1's 2 Cl channels / of dR each; the
                # transformed R1 then also serves conv2's weight gradient, rnh_wino44f_wgrad_v): an OPT-IN (RNH_WINO44_REFINE2=1) - parity green, but the step is
                # the same with it, 238.8 / 239.1 against 238.2 / 239.3 ms at BASELINE config 2 (profiles/r06_y_*): the launches are small and run beside others
                r2_44 = (not bf) and Cl % 64 == 0 and os.environ.get('RNH_WINO44', '1') != '0' and os.environ.get('RNH_WINO44_REFINE2', '0') == '1'
                self.r2_fwd_h.wino44 = r2_44
                self.r2_dgrad_h.wino44 = r2_44
                self.r2_dgrad_x = ConvPlan_('refine2.dgrad.x', k2, None, ws2, [KSeg(Cl, Cl, 0)],
                                           list(range(2 * Cl, C1)) + [-1] * (self.C1p - C1), transposed=True)
                # ... and its weight gradient (round 4; until then one pixel-contraction GEMM over all C1p rows, 2 % of the fp32 step): the 2*Cl
                # hidden-state rows in Winograd form, the row of the phase channel (and the pad rows behind it) through the pixel contraction
                self.r2_wgrad_h = WgradPlan_('refine2.wgrad.h', k2, b2, ws2, [XSeg(2 * Cl, 2 * Cl, 0)], [YSeg(Cl, Cl, 0)])
                self.r2_wgrad_h.wino44f = not bf                    # (rnh_wino44f_wgrad: 1.81 -> 1.55 ms at BASELINE config 2)
                self.r2_wgrad_x = WgradPlan_('refine2.wgrad.x', k2, None, ws2, [XSeg(self.C1p - 2 * Cl, C1 - 2 * Cl, 2 * Cl)], [YSeg(Cl, Cl, 0)])
            self.r2_fwd = ConvPlan_('refine2.fwd', k2, b2, ws2, [KSeg(self.C1p, C1, 0)], list(range(Cl)))
            self.r2_dgrad = ConvPlan_('refine2.dgrad', k2, None, ws2, [KSeg(Cl, Cl, 0)],
                                     list(range(C1)) + [-1] * (self.C1p - C1), transposed=True)
            self.r2_wgrad = WgradPlan_('refine2.wgrad', k2, b2, ws2, [XSeg(self.C1p, C1, 0)], [YSeg(Cl, Cl, 0)])
            self.r1_wgrad = WgradPlan_('refine1.wgrad', k1, b1, ws1, xsegs,
                                      [YSeg(C1 - 1, C1 - 1, 0)] if self.xcol else [YSeg(self.C1p, C1, 0)])
            self.r1_dgrad = ConvPlan_('refine1.dgrad', k1, None, ws1, [KSeg(self.C1p, C1, 0, kcoff=j * C1) for j in range(w)],
                                     list(range(2 * Cl)), transposed=True)
        else:
            ws1 = (Cl, w * 2 * Cl, 1, 1)
            k1, b1 = 'refine_block.body.conv1.weight', 'refine_block.body.conv1.bias'
            segs, xsegs = [], []
            for j in range(w):
                segs += [KSeg(Cl, Cl, j * 2 * Cl), KSeg(Cl, Cl, j * 2 * Cl + Cl)]
                xsegs += [XSeg(Cl, Cl, j * 2 * Cl), XSeg(Cl, Cl, j * 2 * Cl + Cl)]
            self.r1_fwd = ConvPlan_('refine1.fwd', k1, b1, ws1, segs, list(range(Cl)))
            self.r1_wgrad = WgradPlan_('refine1.wgrad', k1, b1, ws1, xsegs, [YSeg(Cl, Cl, 0)])
            self.r1_dgrad = ConvPlan_('refine1.dgrad', k1, None, ws1, [KSeg(Cl, Cl, 0, kcoff=j * 2 * Cl) for j in range(w)],
                                     list(range(2 * Cl)), transposed=True)

        # upsampler: [(conv with PixelShuffle r)]* then the small last conv
        C, s = self.C, cfg.upscale_factor
        self.up = []
        if s == 3:
            rs = [3]
        else:
            rs = [2] * {2: 1, 4: 2, 8: 3}[s]
        for i, r in enumerate(rs):
            wk, bk = f'out_block.conv{i + 1}.weight', f'out_block.conv{i + 1}.bias'
            ws = (r * r * C, C, 3, 3)
            fwd = ConvPlan_(f'up{i + 1}.fwd', wk, bk, ws, [KSeg(C, C, 0)], ps_colmap(C, r), epilogue=L.EPI_PS,
                           wino=os.environ.get('RNH_WINO', '1') != '0' and os.environ.get('RNH_WINO_UP', '1') != '0')
            # the PixelShuffle convolution's forward in F(4x4, 3x3) form (rnh_wino44_conv with the shuffle in its store) on one transform of its input
            fwd.wino44 = (not bf) and fwd.wino and C % 32 == 0 and os.environ.get('RNH_WINO44', '1') != '0' and os.environ.get('RNH_WINO44_UP', '1') != '0'
            dgrad = ConvPlan_(f'up{i + 1}.dgrad', wk, None, ws, [KSeg(C, C, ij) for ij in range(r * r)], list(range(C)),
                             transposed=True, kstride=r * r,
                             wino=os.environ.get('RNH_WINO', '1') != '0' and os.environ.get('RNH_WINO_UP', '1') != '0')
            wgrad = WgradPlan_(f'up{i + 1}.wgrad', wk, bk, ws, [XSeg(C, C, 0)], [YSeg(C, C, ij, r * r) for ij in range(r * r)])
            wgrad.wino44f = not bf                                   # (rnh_wino44f_wgrad on the r*r gathered sub-pixel planes of the output gradient)
            self.up.append(dict(r=r, fwd=fwd, dgrad=dgrad, wgrad=wgrad))
        self.last_w, self.last_b = f'out_block.conv{len(rs) + 1}.weight', f'out_block.conv{len(rs) + 1}.bias'
        # collapsed tail backward (csrc/uptail.hip): wgrad of the last PixelShuffle conv's input against the expanded
        # output gradient D, whose (out_channels * (r+2)^2) columns replace the conv's r*r*C
        rt, Co = rs[-1], cfg.out_channels
        nd2 = Co * (rt + 2) * (rt + 2)
        self.tail_dc = _pad_to(nd2, 8) if bf else r4(nd2)
        self.tail_m = WgradPlan_('uptail.M', None, None, (nd2, C, 3, 3), [XSeg(C, C, 0)], [YSeg(self.tail_dc, nd2, 0)])

    def conv_plans(self):
        out = []
        for v in self.lstm.values():
            out += [v['full'], v['first'], v['dgrad']]
        out += [self.r1_fwd, self.r1_dgrad]
        if getattr(self, 'r1_split', False):
            out += [self.r1_fwd_a, self.r2_dgrad_a, self.r2_dgrad_b] + ([self.r1x_fwd] if self.xcol_m else [self.r1_fwd_b])
        if self.r1_wino:
            out += [self.r1_fwd_h, self.r1_fwd_p, self.r1_dgrad_h, self.r1_dgrad_x]
        if self.pos:
            out += [self.r2_fwd_h, self.r2_fwd_x, self.r2_dgrad_h, self.r2_dgrad_x] if self.r2_wino else [self.r2_fwd, self.r2_dgrad]
        for u in self.up:
            out += [u['fwd'], u['dgrad']]
        if getattr(self, 'r1_split', False):                   # replaced by their .a / .b halves: never launched, not packed
            out = [p for p in out if p is not self.r1_fwd and p is not self.r2_dgrad]
        seen, uniq = set(), []
        for p in out:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        return uniq
