"""RNH_CHECK=1: every big launch of the HIP path is held, right behind the launch, against a float64 restatement of what that launch is meant to
compute, and the first one outside its tolerance raises ``CheckError`` naming the plan, the shape and the form it ran in.

A DEBUGGING MODE ("a debug mode that compares every HIP kernel against the CPU restatement"): the tool the
hand-counted-``s_waitcnt`` hunts of docs/HISTORY.md lacked - a wrong packed weight, a stale operand ring or a mis-addressed tile shows up at the launch
that produced it, not as a loss that is off in the fifth digit three stages later.  It is a checker, not a path: nothing here can produce a result the
step uses - ``CheckedOps`` wraps a ``HipOps``, lets IT launch, then recomputes the launch's outputs from the launch's inputs with plain tensor algebra
in float64 (on the device, through ATen) and compares; it cannot drive the engine by itself, refuses HIP-graph capture, slows a step by orders of
magnitude and keeps a copy of every tensor that went through rnh_wino44_transform.  Imported only when RNH_CHECK=1 (src/model/nets/refine_net.py) and by
tests/ (tests/torch_ops.py shares ``gather_src`` / ``effective_weight`` with it).  It does not import oracle/.

Checked: conv() in every epilogue (plain store with segments / accumulation, PixelShuffle store, the fused ConvLSTM gates; of the fused gate backward
the data-gradient columns), conv_pair(), wino44_cell() / wino44_cell_pair(), wino44_conv(), wgrad(), lstm_gates_bwd() / wino44_gates_bwd().  Passed through unchecked: the
small element-wise / side-path / tail kernels (tests/ hold them against their references one by one).
"""
import torch
import torch.nn.functional as F

from . import lib as L


class CheckError(RuntimeError):
    pass


def gather_src(s, B):
    """(B, H, W, nch) NHWC slice described by a plans.Src."""
    t = s.t[s.img_off:s.img_off + B]
    if s.add is not None:
        t = t + s.add[s.img_off:s.img_off + B]
    if s.scale > 1:
        t = t[:, s.sub[0]::s.scale, s.sub[1]::s.scale]
    nch = t.shape[-1] - s.c0 if s.nch is None else s.nch
    return t[..., s.c0:s.c0 + nch]


def effective_weight(plan, w):
    """[Npad][Ktot][kh][kw] weight equivalent to what rnh_pack_weights builds for ``plan`` (the index conventions of include/refinenet_hip.h:
    column n of the GEMM = output channel colmap[n] (+ a segment's kcoff), K row = input channel kbase + kk * kstride; a transposed plan = the
    data gradient: roles swapped, taps flipped)."""
    kh = 3 if plan.ntaps == 9 else 1
    ktot = sum(sg.nch for sg in plan.ksegs)
    weff = torch.zeros(plan.Npad, ktot, kh, kh, dtype=w.dtype, device=w.device)
    cols = [(n, cm) for n, cm in enumerate(plan.colmap) if cm >= 0]
    if not cols:
        return weff
    nn = torch.tensor([n for n, _ in cols], device=w.device)
    cm = torch.tensor([c for _, c in cols], device=w.device)
    koff = 0
    for sg in plan.ksegs:
        if sg.nvalid:
            kidx = torch.tensor([sg.kbase + kk * plan.kstride for kk in range(sg.nvalid)], device=w.device)
            cme = cm + sg.kcoff
            if plan.transposed:
                blk = torch.flip(w[kidx][:, cme], dims=(2, 3)).transpose(0, 1)          # [col][k][kh][kw] = flip(w[kidx, cme])
            else:
                blk = w[cme][:, kidx]
            weff[nn[:, None], (koff + torch.arange(sg.nvalid, device=w.device))[None, :]] = blk
        koff += sg.nch
    return weff


def _conv64(x, weff, bias, pad):
    """float64 convolution of NCHW x with [N][K][kh][kw] weights by im2col + matmul (MIOpen has no float64 convolution), a few images at a time."""
    Bn, K, H, W = x.shape
    kh = weff.shape[-1]
    wm = weff.reshape(weff.shape[0], -1)
    step = max(1, int(2 ** 27 // max(K * kh * kh * H * W, 1)))
    out = []
    for i in range(0, Bn, step):
        cols = F.unfold(x[i:i + step], kh, padding=pad)                                # (b, K*kh*kh, H*W)
        y = torch.matmul(wm, cols)
        if bias is not None:
            y = y + bias.view(1, -1, 1)
        out.append(y.view(y.shape[0], -1, H, W))
    return torch.cat(out, 0)


class CheckedOps:
    """A HipOps whose big launches are checked against float64 (see the module docstring).  Everything not overridden is the wrapped object's."""

    def __init__(self, inner):
        self.inner = inner
        self._w = {}               # id(plan) -> (plan, w, b): the fp32 parameters the plan was last packed from
        self._vsrc = []            # records of rnh_wino44_transform: (first byte, bytes, images, channels, copy of the source images)
        self.checked = 0           # launches held against float64 so far

    def __getattr__(self, name):
        return getattr(self.inner, name)

    # ---- helpers --------------------------------------------------------------------------------------------------------------
    def _no_capture(self):
        if self.inner.capturing():
            raise CheckError('RNH_CHECK=1 cannot run under HIP-graph capture (it reads results back): use the eager step')

    @staticmethod
    def _round(t, plan):
        """The values the matrix cores see: bf16 operands in the bf16-storage path."""
        return t.float().bfloat16().double() if getattr(plan, 'bf16', False) else t.double()

    def _weights(self, plan):
        if id(plan) not in self._w:
            raise CheckError(f'{plan.name}: launched without a pack() in this process')
        _, w, b = self._w[id(plan)]
        weff = effective_weight(plan, w.detach().float())
        if getattr(plan, 'f16w', False):
            weff = weff.half()
        elif getattr(plan, 'bf16', False):
            weff = weff.bfloat16()
        bp = None
        if b is not None and not plan.transposed and plan.bkey is not None:
            bp = torch.zeros(plan.Npad, dtype=torch.float64, device=w.device)
            cm = torch.tensor(plan.colmap, device=w.device)
            bp[cm >= 0] = b.detach().double()[cm[cm >= 0]]
        return weff.double(), bp

    def _fail(self, what, plan, B, H, W, form, name, got, want, tol):
        err = (got.double() - want).abs()
        i = int(err.argmax())
        raise CheckError(f'RNH_CHECK: {what} {plan.name} (B={B}, {H}x{W}, form: {form}): {name} is off by {float(err.flatten()[i]):.3e} at flat index {i} '
                         f'(got {float(got.flatten()[i]):.6g}, float64 says {float(want.flatten()[i]):.6g}; tolerance {tol:.3e}, largest reference value '
                         f'{float(want.abs().max()):.4g})')

    def _cmp(self, what, plan, B, H, W, form, name, got, want, loose=1.0):
        """fp32 result: 1e-4 of the largest reference value (Winograd transforms, MFMA accumulation order, v_exp / v_rcp activations); a bf16 result: two
        bf16 ulps of the element on top."""
        self.checked += 1
        scale = float(want.abs().max())
        tol = loose * (1e-4 * scale + 1e-6)
        err = (got.double() - want).abs()
        if got.dtype == torch.bfloat16:
            over = err - (2.0 ** -7 * want.abs() + tol)
        else:
            over = err - tol
        if bool(torch.isnan(got.float()).any()) or float(over.max()) > 0:
            self._fail(what, plan, B, H, W, form, name, got, want, tol)

    def _form(self, plan, f44=False):
        if f44:
            return 'Winograd F(4x4,3x3) (rnh_wino44_*)'
        if getattr(plan, 'bf16', False):
            return 'bf16 MFMA (rnh_conv_bf16)' + (', f16 weights' if getattr(plan, 'f16w', False) else '')
        return 'Winograd F(2x2,3x3) (rnh_conv_wino)' if getattr(plan, 'wino', False) else 'implicit GEMM (rnh_conv_igemm)'

    # ---- weights ------------------------------------------------------------------------------------------------------------------
    def pack(self, plan, w, b=None, **forms):
        self._w[id(plan)] = (plan, w, b)
        return self.inner.pack(plan, w, b, **forms)

    # ---- reference of one convolution launch ------------------------------------------------------------------------------------------
    def _conv_ref(self, plan, xs, B):
        """float64 GEMM result (B, H, W, Npad) of the plan over the NHWC sources xs (already gathered)."""
        weff, bp = self._weights(plan)
        x = torch.cat([self._round(t, plan) for t in xs], dim=-1).permute(0, 3, 1, 2)
        if x.shape[1] != weff.shape[1]:
            raise CheckError(f'{plan.name}: the sources hold {x.shape[1]} channels, the plan {weff.shape[1]}')
        return _conv64(x, weff, bp, 1 if plan.ntaps == 9 else 0).permute(0, 2, 3, 1)

    def _before(self, dsts, B):
        return [d.t[d.img_off:d.img_off + B, ..., d.c0:d.c0 + d.ncols].double().clone() if d.accumulate else None for d in (dsts or [])]

    def _check_store(self, what, plan, B, H, W, form, y, dsts, before):
        col = 0
        for i, (d, b0) in enumerate(zip(dsts, before)):
            got = d.t[d.img_off:d.img_off + B, ..., d.c0:d.c0 + d.ncols]
            want = y[..., col:col + d.ncols] + (b0 if b0 is not None else 0)
            self._cmp(what, plan, B, H, W, form, f'destination {i} (columns {col}..{col + d.ncols - 1}{", accumulated" if d.accumulate else ""})', got, want)
            col += d.ncols

    def _check_ps(self, what, plan, B, H, W, form, y, ps):
        t, r = ps
        cq = t.shape[-1]
        v = y[..., :cq * r * r].reshape(B, H, W, r, r, cq).permute(0, 1, 3, 2, 4, 5).reshape(B, H * r, W * r, cq)     # column = (i*r + j)*cq + c
        self._cmp(what, plan, B, H, W, form, 'the pixel-shuffled destination', t, v)

    def _check_lstm(self, what, plan, B, H, W, form, y, lstm):
        hd = lstm['hd']
        cm = torch.tensor(plan.colmap, dtype=torch.long, device=y.device)
        keep = cm >= 0
        v = torch.zeros(B, H, W, 4 * hd, dtype=torch.float64, device=y.device)
        v[..., cm[keep]] = y[..., keep]
        v = v.reshape(B, H, W, 4, hd)
        gi, gf, go, gg = torch.sigmoid(v[..., 0, :]), torch.sigmoid(v[..., 1, :]), torch.sigmoid(v[..., 2, :]), torch.tanh(v[..., 3, :])
        cp = lstm.get('c_prev')
        cn = gf * cp.double() + gi * gg if cp is not None else gi * gg
        if lstm.get('gates_out') is not None:
            self._cmp(what, plan, B, H, W, form, 'the stored gates (i, f, o, g)', lstm['gates_out'], torch.cat([gi, gf, go, gg], dim=-1))
        self._cmp(what, plan, B, H, W, form, "the cell state c'", lstm['c_out'], cn)
        self._cmp(what, plan, B, H, W, form, "the hidden state h'", lstm['h_out'], go * torch.tanh(cn))

    # ---- rnh_conv_igemm / rnh_conv_wino / rnh_conv_bf16 ---------------------------------------------------------------------------------
    def conv(self, plan, srcs, B, H, W, dsts=None, ps=None, lstm=None, lstm_bwd=None):
        self._no_capture()
        xs = [gather_src(s, B).clone() for s in srcs]          # (a destination may alias a source's tensor: what the launch READ)
        cprev = lstm['c_prev'].clone() if lstm is not None and lstm.get('c_prev') is not None else None
        before = self._before(dsts, B)
        self.inner.conv(plan, srcs, B, H, W, dsts=dsts, ps=ps, lstm=lstm, lstm_bwd=lstm_bwd)
        self._check_conv('conv', plan, xs, B, H, W, dsts, ps, dict(lstm, c_prev=cprev) if lstm is not None else None, before, self._form(plan),
                         only_first=lstm_bwd is not None)

    def _check_conv(self, what, plan, xs, B, H, W, dsts, ps, lstm, before, form, only_first=False):
        y = self._conv_ref(plan, xs, B)
        if lstm is not None:
            self._check_lstm(what, plan, B, H, W, form, y, lstm)
        elif ps is not None:
            self._check_ps(what, plan, B, H, W, form, y, ps)
        else:
            self._check_store(what, plan, B, H, W, form, y, dsts[:1] if only_first else dsts, before)

    def conv_pair(self, calls):
        self._no_capture()
        pre = []
        for pl, srcs, B, H, W, kw in calls:
            lstm = kw.get('lstm')
            pre.append(([gather_src(s, B).clone() for s in srcs], self._before(kw.get('dsts'), B),
                        lstm['c_prev'].clone() if lstm is not None and lstm.get('c_prev') is not None else None))
        self.inner.conv_pair(calls)
        for (pl, srcs, B, H, W, kw), (xs, before, cprev) in zip(calls, pre):
            lstm = kw.get('lstm')
            self._check_conv('conv (paired launch)', pl, xs, B, H, W, kw.get('dsts'), kw.get('ps'), dict(lstm, c_prev=cprev) if lstm is not None else None, before,
                             self._form(pl), only_first=kw.get('lstm_bwd') is not None)

    # ---- the F(4x4, 3x3) forms: the launches read TRANSFORMED inputs, so the images behind every transform are remembered --------------------------
    def wino44_transform(self, src, B, H, W, out):
        self._no_capture()
        self.inner.wino44_transform(src, B, H, W, out)
        x = gather_src(src, B).clone()
        lo, nb = out.data_ptr(), out.numel() * out.element_size()
        self._vsrc = [r for r in self._vsrc if r[0] + r[1] <= lo or r[0] >= lo + nb]       # whatever lived in these bytes is gone
        self._vsrc.append((lo, nb, B, x.shape[-1], x))

    def _images(self, v, first_block, need, H, W, nch):
        """The ``need`` images whose transform starts ``first_block`` tile blocks (32 tiles x 36 positions x nch floats) into the buffer v."""
        p = v.data_ptr() + first_block * 32 * 36 * nch * 4
        bpi = (H // 4) * (W // 4) * 36 * nch * 4
        got = []
        while need > 0:
            rec = next((r for r in self._vsrc if r[0] <= p < r[0] + r[1] and r[3] == nch), None)
            if rec is None or (p - rec[0]) % bpi:
                raise CheckError(f'RNH_CHECK: a launch reads transformed images at {p:#x} that no rnh_wino44_transform of this process wrote ({nch} channels)')
            i0 = (p - rec[0]) // bpi
            take = min(need, rec[2] - i0)
            if take <= 0:
                raise CheckError('RNH_CHECK: a launch reads past the images of a transform')
            got.append(rec[4][i0:i0 + take])
            need -= take
            p += take * bpi
        return torch.cat(got, 0) if len(got) > 1 else got[0]

    def wino44_cell(self, plan, vsrcs, B, H, W, lstm):
        self._no_capture()
        xs = [self._images(v, 0, B, H, W, sg.nch) for v, sg in zip(vsrcs, plan.ksegs)]
        cprev = lstm['c_prev'].clone() if lstm.get('c_prev') is not None else None
        self.inner.wino44_cell(plan, vsrcs, B, H, W, lstm)
        self._check_conv('cell', plan, xs, B, H, W, None, None, dict(lstm, c_prev=cprev), None, self._form(plan, True))

    def wino44_cell_pair(self, calls, B, H, W):
        self._no_capture()
        pre = [([self._images(v, 0, B, H, W, sg.nch) for v, sg in zip(vs, pl.ksegs)], ls['c_prev'].clone() if ls.get('c_prev') is not None else None)
               for pl, vs, ls in calls]
        self.inner.wino44_cell_pair(calls, B, H, W)
        for (pl, vs, ls), (xs, cprev) in zip(calls, pre):
            self._check_conv('cell (paired launch)', pl, xs, B, H, W, None, None, dict(ls, c_prev=cprev), None, self._form(pl, True))

    def wino44_conv(self, plan, vsrcs, B, H, W, dst=None, ps=None):
        self._no_capture()
        xs = [self._images(v, off, B, H, W, sg.nch) for (v, off), sg in zip(vsrcs, plan.ksegs)]
        dsts = None if dst is None else (list(dst) if isinstance(dst, (list, tuple)) else [dst])
        before = self._before(dsts, B)
        self.inner.wino44_conv(plan, vsrcs, B, H, W, dst, ps=ps) if ps is not None else self.inner.wino44_conv(plan, vsrcs, B, H, W, dst)
        self._check_conv('convolution', plan, xs, B, H, W, dsts, ps, None, before, self._form(plan, True))

    # ---- weight gradients -----------------------------------------------------------------------------------------------------------------
    def wgrad(self, plan, xsrcs, ysrcs, B, H, W, dw, db=None, accumulate=False, vsrcs=None, vN=None):
        self._no_capture()
        dw0 = dw.double().clone() if accumulate else None
        db0 = db.double().clone() if accumulate and db is not None else None
        self.inner.wgrad(plan, xsrcs, ysrcs, B, H, W, dw, db, accumulate=accumulate, vsrcs=vsrcs, vN=vN)   # (held against the RAW sources below)
        x = torch.cat([self._round(gather_src(s, B), plan) for s in xsrcs], dim=-1).permute(0, 3, 1, 2)
        dy = torch.cat([self._round(gather_src(s, B), plan) for s in ysrcs], dim=-1).permute(0, 3, 1, 2)
        kh = 3 if plan.ntaps == 9 else 1
        g = torch.zeros(dy.shape[1], x.shape[1] * kh * kh, dtype=torch.float64, device=dw.device)
        step = max(1, int(2 ** 27 // max(x.shape[1] * kh * kh * H * W, 1)))
        for i in range(0, B, step):
            cols = F.unfold(x[i:i + step], kh, padding=kh // 2)                            # (b, K*kh*kh, H*W)
            g += torch.einsum('bnp,bkp->nk', dy[i:i + step].reshape(dy[i:i + step].shape[0], dy.shape[1], -1), cols)
        g = g.view(dy.shape[1], x.shape[1], kh, kh)
        bsum = dy.sum(dim=(0, 2, 3))
        want_w = dw0.clone() if accumulate else dw.double().clone()                        # (elements the plan does not map keep what they held)
        want_b = (db0.clone() if accumulate else db.double().clone()) if db is not None else None
        cols_ = [(j, co) for j, co in enumerate(plan.colmap[:dy.shape[1]]) if co >= 0]
        rows_ = [(i, ci) for i, ci in enumerate(plan.rowmap[:x.shape[1]]) if ci >= 0]
        if cols_ and rows_:
            jj, co = (torch.tensor(v, device=dw.device) for v in zip(*cols_))
            ii, ci = (torch.tensor(v, device=dw.device) for v in zip(*rows_))
            blk = g[jj][:, ii]
            want_w[co[:, None], ci[None, :]] = blk + (dw0[co[:, None], ci[None, :]] if accumulate else 0)
            if db is not None:
                want_b[co] = bsum[jj] + (db0[co] if accumulate else 0)
        form = 'bf16 MFMA (rnh_wgrad_bf16)' if getattr(plan, 'bf16', False) else 'rnh_wino_wgrad / rnh_conv_wgrad'
        self._cmp('weight gradient', plan, B, H, W, form, 'dW', dw, want_w, loose=10.0)
        if db is not None:
            self._cmp('weight gradient', plan, B, H, W, form, 'db', db, want_b, loose=10.0)

    # ---- gate backward ----------------------------------------------------------------------------------------------------------------------
    def lstm_gates_bwd(self, dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, dh2=None):
        self._no_capture()
        self.inner.lstm_gates_bwd(dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, dh2=dh2)
        self._check_gates_bwd('rnh_lstm_gates_bwd', dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, dh2)

    def wino44_gates_bwd(self, dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, dh2, v):
        """The gate backward that also writes the transformed gate gradients: its dgates / dc_prev are checked here, its transformed image by the
        F(4x4) data gradient that reads it (recorded, like a rnh_wino44_transform, as the image of what float64 says the gate gradients are)."""
        self._no_capture()
        self.inner.wino44_gates_bwd(dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, dh2, v)
        want = self._check_gates_bwd('rnh_wino44_gates_bwd', dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, dh2)
        lo, nb = v.data_ptr(), v.numel() * v.element_size()
        self._vsrc = [r for r in self._vsrc if r[0] + r[1] <= lo or r[0] >= lo + nb]
        self._vsrc.append((lo, nb, dh.shape[0], want.shape[-1], want.float()))

    def _check_gates_bwd(self, who, dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, dh2):
        hd = dh.shape[-1]
        d = dh.double() + (dh2.double() if dh2 is not None else 0)
        g = gates.double()
        gi, gf, go, gg = (g[..., k * hd:(k + 1) * hd] for k in range(4))
        th = torch.tanh(c_next.double())
        dct = d * go * (1 - th * th) + (dc_next.double() if dc_next is not None else 0)
        cp = c_prev.double() if c_prev is not None else torch.zeros_like(d)
        want = torch.cat([dct * gg * gi * (1 - gi), dct * cp * gf * (1 - gf), d * th * go * (1 - go), dct * gi * (1 - gg * gg)], dim=-1)

        class _P:                                           # (no plan: a name for the message)
            name = who
        B, H, W = dh.shape[:3]
        self._cmp('gate backward', _P, B, H, W, 'element-wise', 'dgates', dgates, want)
        if dc_prev is not None:
            self._cmp('gate backward', _P, B, H, W, 'element-wise', 'dc_prev', dc_prev, dct * gf)
        return want
