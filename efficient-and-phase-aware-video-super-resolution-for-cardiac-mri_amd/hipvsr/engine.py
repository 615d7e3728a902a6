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
from .plans import Dst, NetPlans, Src


class Context:
    """What the forward keeps for the backward."""
    __slots__ = ('N', 'H', 'W', 'F', 'T', 'x_all', 'P4', 'stages', 'packed_dgrad', 'tail_bf16')

    def __init__(self):
        self.stages = []
        self.packed_dgrad = False


class RefineNetEngine:
    def __init__(self, cfg, ops, dtype='f32'):
        """dtype 'f32': everything fp32 (the reference's precision).  dtype 'bf16' (BASELINE.json configs[2]): the bf16-storage
        path - feature maps, hidden states, saved gates and the activation gradients between the big convolutions live in
        HBM as bf16 and all 3x3 convolutions run on bf16 MFMA with fp32 accumulators (rnh_conv_bf16 / rnh_wgrad_bf16);
        the cell state c and its gradient, the upsampler's inner feature map (which the collapsed tail kernels consume),
        the outputs, the loss and every parameter gradient stay fp32; the state_dict is fp32 either way."""
        import torch
        if dtype not in ('f32', 'bf16'):
            raise ValueError(f"compute dtype must be 'f32' or 'bf16', got {dtype!r}")
        self.cfg, self.ops, self.dtype = cfg, ops, dtype
        self.bf16 = dtype == 'bf16'
        self.act = torch.bfloat16 if self.bf16 else torch.float32
        self.f32 = torch.float32
        self.plans = NetPlans(cfg, bf16=self.bf16)
        self.hw = cfg.refine_window_size // 2
        if (3 if self.plans.pos else 2) * cfg.refine_window_size > L.MAX_SRC:
            raise ValueError(f'refine_window_size {cfg.refine_window_size} needs more than {L.MAX_SRC} conv sources')
        if cfg.num_features[0] != cfg.num_features[-1]:
            raise ValueError('num_features[0] must equal num_features[-1] (residual add, refine_net.py:102)')

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

    def _pack(self, params, which):
        P = self.plans
        for pl in P.conv_plans():
            is_dgrad = pl.transposed
            if (which == 'fwd') == (not is_dgrad):
                self.ops.pack(pl, params[pl.wkey], params[pl.bkey] if pl.bkey else None)

    # ------------------------------------------------------------------------------------------------
    def forward(self, params, inputs, pos_codes, need_grad, last_only=False):
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
        params = self._views(params)
        x_all = ops.stack_inputs(inputs)                       # (F*N, H, W, Cin)
        ctx.x_all = x_all
        self._pack(params, 'fwd')
        act, f32 = self.act, self.f32
        feat = ops.inconv_fwd(x_all, params['in_block.conv.weight'], params['in_block.conv.bias'],
                              params['in_block.prelu.weight'])
        if self.bf16:
            feat = ops.cast(feat, act)                            # the input block's features cross into bf16 storage here
        P4 = (ops.phase_plane(pos_codes, N, F, H, W, dtype=act, channels=P.pw) if self.bf16 else
              ops.phase_plane(pos_codes, N, F, H, W)) if P.pos else None
        # the tail kernels read their input in fp32 (csrc/uptail.hip) or, for the x4 / x8 nets' r = 2 tail, in bf16
        # (csrc/uptail_bf16.hip): then every inner feature map of the upsampler and its gradient are bf16 too.  With a single
        # PixelShuffle stage the tail's input is Sb, kept fp32
        tail_bf16 = self.bf16 and len(P.up) > 1 and ops.uptail_bf16_supported(C, P.up[-1]['r'], cfg.out_channels)
        ctx.tail_bf16 = tail_bf16
        sb_dt = f32 if len(P.up) == 1 else act
        ctx.P4 = P4
        O_all = ops.empty(S, 3, TN, s_up * H, s_up * W, cfg.out_channels)

        for s in range(S):
            st = dict(feat=feat)
            # ---- bidirectional ConvLSTM over the frames (refine_net.py:82-93) ----------------------------
            # One cell launch (N images) fills the chip for well under a millisecond, so its ramp-up and tail matter.
            # Cell (direction d, layer l, frame k) only needs (d, l-1, k) and (d, l, k-1): every (d, l) gets its own
            # stream, ordered along k by the stream and against the layer below by an event, and up to 2*L cells
            # of the layer/frame wavefront run concurrently.
            dirs = ('forward', 'backward')
            for d in dirs:
                st[d] = dict(H=[ops.empty(F * N, H, W, hd, dtype=act) for hd in nf], C=[ops.empty(F * N, H, W, hd) for hd in nf],
                             G=[ops.empty(TN, H, W, 4 * hd, dtype=act) for hd in nf] if need_grad else None)
            # In the last stage nothing reads the hidden states past the last refine window (the feature update after it is
            # dead, quirk Q5): the forward direction stops at frame U+T-1+hw, the backward one at U-hw, and the refine
            # block only computes the T supervised windows.  Outputs and gradients are unchanged.
            last = s == S - 1
            F_s = U + T + hw if last else F
            ops.fork(2 * Lr)
            for idx in range(F_s):
                for di, d in enumerate(dirs):
                    k = idx if d == 'forward' else F - 1 - idx
                    prev = None if idx == 0 else (k - 1 if d == 'forward' else k + 1)
                    Hb, Cb, Gb = st[d]['H'], st[d]['C'], st[d]['G']
                    grad_frame = need_grad and U <= k < U + T
                    below = None
                    for l in range(Lr):
                        with ops.side(di * Lr + l):
                            if below is not None:
                                ops.wait(below)
                            pl = P.lstm[(d, l)]
                            xin = feat if l == 0 else Hb[l - 1]
                            srcs = [Src(xin, img_off=k * N)]
                            if cfg.memory:
                                if prev is not None:
                                    srcs.append(Src(Hb[l], img_off=prev * N))
                                    plan = pl['full']
                                else:
                                    plan = pl['first']
                            else:
                                srcs.append(Src(xin, img_off=k * N))
                                plan = pl['full']
                            ops.conv(plan, srcs, N, H, W, lstm=dict(
                                hd=pl['hd'], c_prev=Cb[l][prev * N:(prev + 1) * N] if prev is not None else None,
                                h_out=Hb[l][k * N:(k + 1) * N], c_out=Cb[l][k * N:(k + 1) * N],
                                gates_out=Gb[l][(k - U) * N:(k - U + 1) * N] if grad_frame else None))
                            below = ops.record() if l + 1 < Lr else None
            ops.join(2 * Lr)
            Hf, Hbk = st['forward']['H'][-1], st['backward']['H'][-1]

            # ---- phase-aware refine block over all windows (refine_net.py:157-185) -----------------------
            w0, nwin = (U - hw, T) if last else (0, F - 2 * hw)     # first window computed, number of windows
            srcs = []
            for j in range(w):
                srcs += [Src(Hf, img_off=(w0 + j) * N), Src(Hbk, img_off=(w0 + j) * N)]
                if P.pos:
                    srcs.append(Src(P4, img_off=(w0 + j) * N))
            R = ops.empty(nwin * N, H, W, Cl, dtype=act)
            if P.pos:
                R1 = ops.empty(nwin * N, H, W, P.C1p, dtype=act)
                lo, hi = w0 * N, (w0 + nwin + w - 1) * N            # the source frames of these windows
                if P.r1_wino:
                    ops.conv(P.r1_fwd_h, [sc for sc in srcs if sc.t is not P4], nwin * N, H, W, dsts=[Dst(R1, P.r1_cols)])
                    ops.refine_phase_bias(R1, P4[lo:hi], params[P.r1_fwd_h.wkey], N, w, Cl, P.r1_cols)
                elif P.xcol_m:
                    # bf16 path: 2*Cl columns in one launch; the last channel frame by frame (one small convolution over the source
                    # frames whose columns are the window slots, then a sum over the slots)
                    ops.conv(P.r1_fwd_a, srcs, nwin * N, H, W, dsts=[Dst(R1, 2 * Cl)])
                    nfr = nwin + w - 1
                    Z5 = ops.empty(nfr * N, H, W, 8)
                    ops.conv(P.r1x_fwd, [Src(Hf, img_off=lo), Src(Hbk, img_off=lo), Src(P4, img_off=lo)], nfr * N, H, W, dsts=[Dst(Z5, 8)])
                    ops.xcol_combine_m(Z5, params[P.r1_fwd.bkey], R1, N, w, 2 * Cl)
                elif P.r1_split:
                    ops.conv(P.r1_fwd_a, srcs, nwin * N, H, W, dsts=[Dst(R1, 2 * Cl)])
                    ops.conv(P.r1_fwd_b, srcs, nwin * N, H, W, dsts=[Dst(R1, P.C1p - 2 * Cl, c0=2 * Cl)])
                else:
                    ops.conv(P.r1_fwd, srcs, nwin * N, H, W, dsts=[Dst(R1, P.r1_cols)])
                if P.xcol:
                    ops.refine_xcol_fwd([Hf[lo:hi], Hbk[lo:hi], P4[lo:hi]], params[P.r1_fwd.wkey], params[P.r1_fwd.bkey], R1, N, w, Cl)
                if P.r2_wino:
                    ops.conv(P.r2_fwd_h, [Src(R1, nch=2 * Cl)], nwin * N, H, W, dsts=[Dst(R, Cl)])
                    ops.conv(P.r2_fwd_x, [Src(R1, c0=2 * Cl, nch=P.C1p - 2 * Cl)], nwin * N, H, W, dsts=[Dst(R, Cl, accumulate=True)])
                else:
                    ops.conv(P.r2_fwd, [Src(R1)], nwin * N, H, W, dsts=[Dst(R, Cl)])
                st['R1'], st['w0'] = (R1 if need_grad else None), w0
            else:
                ops.conv(P.r1_fwd, srcs, nwin * N, H, W, dsts=[Dst(R, Cl)])

            # ---- three output groups through the upsampler (refine_net.py:100-113, :194-205) --------------
            fc = feat[U * N:(U + T) * N]
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
                ops.add(Sb[0:TN], fc, Hf[U * N:(U + T) * N])
                ops.add(Sb[TN:2 * TN], fc, Hbk[U * N:(U + T) * N])
                ops.add(Sb[2 * TN:], fc, R[(U - hw - w0) * N:(U - hw - w0 + T) * N])
            Oview = O_all[s] if nb == 3 else O_all[s, 2:3]
            cur, h, wd, Ys = (Sb if nb else None), H, W, []
            fused_tail = ops.uptail_fwd_supported(P.up[-1]['r'], cfg.out_channels)
            for ui, u in enumerate(P.up if nb else []):
                r = u['r']
                if fused_tail and ui == len(P.up) - 1:
                    # last PixelShuffle conv + final conv as one composed 5x5 convolution (csrc/uptail.hip): the
                    # r*r*C-channel tensor between them is never formed, neither here nor in the backward
                    ops.uptail_fwd(cur, params[u['fwd'].wkey], params[u['fwd'].bkey], params[P.last_w], params[P.last_b], r,
                                   Oview.reshape(nb * TN, h * r, wd * r, cfg.out_channels))
                    cur = None
                    break
                Y = ops.empty(nb * TN, h * r, wd * r, C, dtype=act if tail_bf16 else f32)
                ops.conv(u['fwd'], [Src(cur)], nb * TN, h, wd, ps=(Y, r))
                Ys.append(Y)
                cur, h, wd = Y, h * r, wd * r
            if cur is not None:
                ops.outconv_fwd(cur, params[P.last_w], params[P.last_b], out=Oview.reshape(nb * TN, h, wd, cfg.out_channels))
                Ys = Ys[:-1]                                  # the tail's output is not needed by the collapsed backward
            if need_grad:
                st['Sb'], st['Ys'] = Sb, Ys
                ctx.stages.append(st)

            # ---- feature update (refine_net.py:118-133), out of place ---------------------------------------
            if S > 1 and s < S - 1:
                nfeat = ops.empty(F * N, H, W, C, dtype=act)
                ops.add(nfeat[0:hw * N], feat[0:hw * N], Hf[0:hw * N])
                ops.add(nfeat[hw * N:(F - hw) * N], feat[hw * N:(F - hw) * N], R)
                ops.add(nfeat[(F - hw) * N:], feat[(F - hw) * N:], Hbk[(F - hw) * N:])
                feat = nfeat
        return O_all, (ctx if need_grad else None)

    # ------------------------------------------------------------------------------------------------
    def backward(self, params, ctx, dO_all, flat=None):
        """dO_all: gradient of the loss w.r.t. O_all (same shape).  Returns an OrderedDict name -> gradient in
        state-dict order (None for parameters that take no part, quirk Q1).  If ``flat`` (a 1-D buffer with
        room for every parameter) is given the gradients are views into it (for a single all-reduce)."""
        ops, cfg, P = self.ops, self.cfg, self.plans
        U, S, hw, w = cfg.num_updated_frames, cfg.num_stages, self.hw, cfg.refine_window_size
        N, H, W, F, T = ctx.N, ctx.H, ctx.W, ctx.F, ctx.T
        nf, Lr, C, Cl = P.nf, P.L, P.C, P.Cl
        TN = T * N
        act = self.act
        params = self._views(params)
        self._pack(params, 'bwd')

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
        for s in range(S - 1, -1, -1):
            st = ctx.stages[s]
            feat, Sb, Ys = st['feat'], st['Sb'], st['Ys']
            Hf, Hbk = st['forward']['H'][-1], st['backward']['H'][-1]
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
            if ops.uptail_xcorr_supported(C, rt, cfg.out_channels):
                M, Sd = ops.uptail_xcorr(xin, dO, rt)
            else:
                D = ops.uptail_expand(dO, rt, P.tail_dc)
                M = ops.empty(P.tail_m.Cout, C, 3, 3)
                Sd = ops.empty(P.tail_m.Cout)
                ops.wgrad(P.tail_m, [Src(xin)], [Src(D)], 3 * TN, h_in, w_in, M, Sd, accumulate=False)
            a2 = acc(ut['wgrad'].wkey)
            acc(ut['wgrad'].bkey)
            a3 = acc(P.last_w)
            acc(P.last_b)
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
                ops.wgrad(u['wgrad'], [Src(xin)], ysrcs, 3 * TN, h_in, w_in, grads[u['wgrad'].wkey], grads[u['wgrad'].bkey],
                          accumulate=a)
                dnext = ops.empty(3 * TN, h_in, w_in, C, dtype=act)
                ops.conv(u['dgrad'], ysrcs, 3 * TN, h_in, w_in, dsts=[Dst(dnext, C)])
                dcur = dnext
            dS = dcur
            dHf, dHb, dR = dS[0:TN], dS[TN:2 * TN], dS[2 * TN:3 * TN]
            dfeat = ops.empty(TN, H, W, C, dtype=act)
            ops.add(dfeat, dHf, dHb, dR)
            if dfeat_next is not None:                # feat[s+1] = feat[s] + R[s] on the supervised frames
                ops.add(dfeat, dfeat_next, accumulate=True)
                ops.add(dR, dfeat_next, accumulate=True)
            st['Sb'] = st['Ys'] = None

            # ---- refine block backward on the T supervised windows --------------------------------------------
            xs = []
            for j in range(w):
                xs += [Src(Hf, img_off=(U - hw + j) * N), Src(Hbk, img_off=(U - hw + j) * N)]
                if P.pos:
                    xs.append(Src(ctx.P4, img_off=(U - hw + j) * N))
            k1, b1 = P.r1_wgrad.wkey, P.r1_wgrad.bkey
            if P.pos:
                # the T middle frames are written by conv2's data gradient, the window halo on both sides stays zero
                dR1p = ops.halo_buffer('dR1p', ((T + 2 * hw) * N, H, W, P.C1p), act, hw * N, (hw + T) * N)
                if P.r2_wino:
                    ops.conv(P.r2_dgrad_h, [Src(dR)], TN, H, W, dsts=[Dst(dR1p, 2 * Cl, img_off=hw * N)])
                    ops.conv(P.r2_dgrad_x, [Src(dR)], TN, H, W, dsts=[Dst(dR1p, P.C1p - 2 * Cl, c0=2 * Cl, img_off=hw * N)])
                elif P.r1_split:
                    ops.conv(P.r2_dgrad_a, [Src(dR)], TN, H, W, dsts=[Dst(dR1p, 2 * Cl, img_off=hw * N)])
                    ops.conv(P.r2_dgrad_b, [Src(dR)], TN, H, W, dsts=[Dst(dR1p, P.C1p - 2 * Cl, c0=2 * Cl, img_off=hw * N)])
                else:
                    ops.conv(P.r2_dgrad, [Src(dR)], TN, H, W, dsts=[Dst(dR1p, P.C1p, img_off=hw * N)])
                a = acc(P.r2_wgrad.wkey)
                acc(P.r2_wgrad.bkey)
                ops.wgrad(P.r2_wgrad, [Src(st['R1'], img_off=(U - hw - st['w0']) * N)], [Src(dR)], TN, H, W, grads[P.r2_wgrad.wkey],
                          grads[P.r2_wgrad.bkey], accumulate=a)
                a = acc(k1)
                acc(b1)
                ysrc = [Src(dR1p, nch=P.r1_cols, img_off=hw * N)]
                if P.xcol_m:
                    # 2*Cl columns against the window sources; the last channel's weights as the gradient of the per-frame convolution
                    ops.wgrad(P.r1_wgrad_a, xs, [Src(dR1p, nch=2 * Cl, img_off=hw * N)], TN, H, W, grads[k1], grads[b1], accumulate=a)
                    E = ops.xcol_gather_m(dR1p[hw * N:(hw + T) * N], N, w, 2 * Cl, act)
                    f0, nfr = (U - hw) * N, T + w - 1
                    dbx = ops.zeros(8)                                   # (zeros: the launch below may run in accumulate mode)
                    ops.wgrad(P.r1x_wgrad, [Src(Hf, img_off=f0), Src(Hbk, img_off=f0), Src(ctx.P4, img_off=f0)], [Src(E)], nfr * N, H, W,
                              grads[k1][P.C1 - 1].view(w, P.C1, 3, 3), dbx[:w], accumulate=a)
                    ops.put_scalar(grads[b1][P.C1 - 1:P.C1], dbx[0:1], a)
                elif P.r1_wino:
                    # hidden-state rows in Winograd form; the five phase-plane rows through the pixel-contraction kernel
                    ops.wgrad(P.r1_wgrad_h, [sc for sc in xs if sc.t is not ctx.P4], ysrc, TN, H, W, grads[k1], grads[b1], accumulate=a)
                    if Cl % 64 == 0 and (P.r1_cols // 4) in (8, 16, 32, 64):
                        # the five phase-plane rows from border-class sums of the gradient (rnh_phase_wgrad)
                        lo, hi = (U - hw) * N, (U - hw + T + w - 1) * N
                        ops.refine_phase_wgrad(dR1p[hw * N:(hw + T) * N], ctx.P4[lo:hi], grads[k1], N, w, Cl, P.r1_cols, a)
                    else:
                        ops.wgrad(P.r1_wgrad_p, [sc for sc in xs if sc.t is ctx.P4], ysrc, TN, H, W, grads[k1], None, accumulate=a)
                else:
                    ops.wgrad(P.r1_wgrad, xs, ysrc, TN, H, W, grads[k1], grads[b1], accumulate=a)
                if P.xcol:
                    lo, hi = (U - hw) * N, (U - hw + T + w - 1) * N
                    ops.refine_xcol_wgrad([Hf[lo:hi], Hbk[lo:hi], ctx.P4[lo:hi]], dR1p[hw * N:(hw + T) * N], grads[k1], grads[b1], N, w,
                                          Cl, a)
                gsrc = dR1p
                st['R1'] = None
            else:
                gsrc = ops.zeros((T + 2 * hw) * N, H, W, Cl, dtype=act)
                ops.add(gsrc[hw * N:(hw + T) * N], dR)
                a = acc(k1)
                acc(b1)
                ops.wgrad(P.r1_wgrad, xs, [Src(dR)], TN, H, W, grads[k1], grads[b1], accumulate=a)
            # data gradient in gather form: frame f collects from the windows f+hw-j that used it in slot j
            if P.r1_wino:
                nm = P.r1_cols
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
            fused = cfg.memory and all(ops.lstm_bwd_fusable(P.lstm[(d, l)]['dgrad'], P.lstm[(d, l)]['cx'], P.lstm[(d, l)]['hd'])
                                       for d in dirs for l in range(Lr))
            DCP = {d: [[ops.empty(N, H, W, hd) for _ in range(2)] for hd in nf] for d in dirs}
            DHP = {d: [[ops.empty(N, H, W, hd, dtype=act) for _ in range(2)] for hd in nf] for d in dirs} if cfg.memory and not fused else None
            TMP = None if cfg.memory else {d: [ops.empty(N, H, W, P.lstm[(d, l)]['cx'], dtype=act) for l in range(Lr)] for d in dirs}
            dh_next = {d: [None] * Lr for d in dirs}
            dc_next = {d: [None] * Lr for d in dirs}
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
                for tau in range(T + Lr - 1):
                    for di, d in enumerate(dirs):
                        step = 1 if d == 'forward' else -1
                        sd, top = st[d], tops[d]
                        Cb, Gb = sd['C'], sd['G']
                        for l in range(Lr - 1, -1, -1):
                            idx = tau - (Lr - 1 - l)
                            if not 0 <= idx < T:
                                continue
                            k = U + T - 1 - idx if d == 'forward' else U + idx
                            fi, k2 = k - U, k - step
                            fi2, has_next = k2 - U, idx + 1 < T
                            pl = P.lstm[(d, l)]
                            hd, cx = pl['hd'], pl['cx']
                            dh_of = (lambda f: top[f * N:(f + 1) * N]) if l == Lr - 1 else (lambda f, l=l: DX[d][l + 1][f * N:(f + 1) * N])
                            c_at = lambda kk, l=l: Cb[l][kk * N:(kk + 1) * N] if 0 <= kk < F else None
                            with ops.side(di * Lr + l):
                                if idx == 0:                        # head of the chain
                                    if l < Lr - 1:
                                        ops.wait(evs[(d, l + 1, 0)])
                                    ops.lstm_gates_bwd(dh_of(fi), None, Gb[l][fi * N:(fi + 1) * N], c_at(k2), c_at(k),
                                                       Gd[d][l][fi * N:(fi + 1) * N], DCP[d][l][0] if has_next else None)
                                dxbuf = (DX[d][l] if l > 0 else dfeat_d[d])[fi * N:(fi + 1) * N]
                                bw = None
                                if has_next:
                                    if l < Lr - 1:
                                        ops.wait(evs[(d, l + 1, idx + 1)])
                                    bw = dict(dh=dh_of(fi2), dc_next=DCP[d][l][idx & 1], gates=Gb[l][fi2 * N:(fi2 + 1) * N], c_prev=c_at(k2 - step),
                                              c_next=c_at(k2), dgates=Gd[d][l][fi2 * N:(fi2 + 1) * N],
                                              dc_prev=DCP[d][l][(idx + 1) & 1] if idx + 2 < T else None, hd=hd, rec_dtype=act)
                                ops.conv(pl['dgrad'], [Src(Gd[d][l][fi * N:(fi + 1) * N])], N, H, W, dsts=[Dst(dxbuf, cx)], lstm_bwd=bw)
                                if l > 0:
                                    evs[(d, l, idx)] = ops.record()
            for idx in range(T if not fused else 0):
                for di, d in enumerate(dirs):
                    step = 1 if d == 'forward' else -1
                    k = U + T - 1 - idx if d == 'forward' else U + idx
                    sd, top = st[d], tops[d]
                    Hb, Cb, Gb = sd['H'], sd['C'], sd['G']
                    fi = k - U
                    prevk = k - step
                    prev_grad = U <= prevk < U + T
                    dx_above, above = None, None
                    for l in range(Lr - 1, -1, -1):
                        with ops.side(di * Lr + l):
                            if above is not None:
                                ops.wait(above)
                            pl = P.lstm[(d, l)]
                            hd, cx = pl['hd'], pl['cx']
                            dh = top[fi * N:(fi + 1) * N] if l == Lr - 1 else dx_above
                            c_prev = Cb[l][prevk * N:(prevk + 1) * N] if 0 <= prevk < F else None
                            dg = Gd[d][l][fi * N:(fi + 1) * N]
                            dcp = DCP[d][l][idx & 1] if prev_grad else None
                            ops.lstm_gates_bwd(dh, dc_next[d][l], Gb[l][fi * N:(fi + 1) * N], c_prev, Cb[l][k * N:(k + 1) * N], dg,
                                               dcp, dh2=dh_next[d][l])
                            dxbuf = (DX[d][l] if l > 0 else dfeat_d[d])[fi * N:(fi + 1) * N]
                            dhp = None
                            if cfg.memory:
                                dsts = [Dst(dxbuf, cx)]
                                if prev_grad:
                                    dhp = DHP[d][l][idx & 1]
                                    dsts.append(Dst(dhp, hd))
                                ops.conv(pl['dgrad'], [Src(dg)], N, H, W, dsts=dsts)
                            else:
                                tmp = TMP[d][l]
                                ops.conv(pl['dgrad'], [Src(dg)], N, H, W, dsts=[Dst(dxbuf, cx), Dst(tmp, cx)])
                                ops.add(dxbuf, tmp, accumulate=True)
                            dh_next[d][l], dc_next[d][l] = dhp, dcp
                            dx_above = dxbuf if l > 0 else None
                            above = ops.record() if l > 0 else None
            # weight gradients of the cells, batched over the T frames (each on its cell's stream)
            for di, d in enumerate(dirs):
                step = 1 if d == 'forward' else -1
                Hb = st[d]['H']
                for l in range(Lr):
                    with ops.side(di * Lr + l):
                        pl = P.lstm[(d, l)]
                        xin = feat if l == 0 else Hb[l - 1]
                        second = Src(Hb[l], img_off=(U - step) * N) if cfg.memory else Src(xin, img_off=U * N)
                        wk, bk = pl['wgrad'].wkey, pl['wgrad'].bkey
                        a = acc(wk)
                        acc(bk)
                        ops.wgrad(pl['wgrad'], [Src(xin, img_off=U * N), second], [Src(Gd[d][l])], TN, H, W, grads[wk], grads[bk],
                                  accumulate=a)
            ops.join(2 * Lr)
            ops.add(dfeat, dfeat_d['forward'], dfeat_d['backward'], accumulate=True)
            st['forward'] = st['backward'] = None
            dfeat_next = dfeat
            ctx.stages[s] = None

        # ---- input block backward (supervised frames only, refine_net.py:66-67) --------------------------------
        xc = ctx.x_all[U * N:(U + T) * N]
        if self.bf16:
            dfeat_next = ops.cast(dfeat_next, self.f32)           # back across the precision boundary of the input block
        ops.inconv_bwd(xc, params['in_block.conv.weight'], params['in_block.conv.bias'], params['in_block.prelu.weight'],
                       dfeat_next, grads['in_block.conv.weight'], grads['in_block.conv.bias'], grads['in_block.prelu.weight'],
                       accumulate=False)
        return grads
