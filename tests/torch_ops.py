"""TEST DOUBLE: a plain-PyTorch (CPU or GPU) implementation of the ``ops`` interface of hipvsr.engine.

It exists only so that (a) the engine's forward/backward *scheduling* and the index maps of
hipvsr.plans can be checked against the oracle on machines without a GPU, and (b) every HIP kernel can be
compared, through the C ABI, against the semantics written here on random descriptors.  It lives under
tests/ and is never importable from the product package.
"""
import os

import torch
import torch.nn.functional as F

from hipvsr import lib as L
from hipvsr.check_ops import effective_weight, gather_src      # noqa: F401  (shared with the RNH_CHECK=1 checker; re-exported)
from hipvsr.plans import ConvPlan, Dst, WgradPlan


class TorchOps:
    name = 'torch-double'

    def __init__(self, device='cpu'):
        self.device = torch.device(device)
        self._w = {}

    def fork(self, n, bank=0):
        pass

    def join(self, n):
        pass

    def record(self):
        return None

    def wait(self, ev):
        pass

    def side(self, i):
        import contextlib
        return contextlib.nullcontext()

    def aside(self, tag=''):
        import contextlib
        return contextlib.nullcontext()

    def rejoin(self):
        pass

    def fence(self):
        pass

    def empty(self, *shape, dtype=torch.float32):
        return torch.full(shape, float('nan'), dtype=dtype, device=self.device)   # poison: catches unwritten reads

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def halo_buffer(self, key, shape, dtype, lo, hi):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    # bf16-storage path: arithmetic in fp32 on operands ROUNDED to bf16 where the HIP kernels round them (the MFMA
    # operands; every store into a bf16 tensor rounds once more through copy_)
    @staticmethod
    def _r(t, on):
        return t.float().bfloat16().float() if on else t.float()

    def cast(self, t, dtype):
        return t if t.dtype == dtype else t.to(dtype)

    def stack_inputs(self, inputs):
        x = torch.stack([t.to(self.device, torch.float32) for t in inputs], dim=0)
        Fr, N, Cin, H, W = x.shape
        return x.permute(0, 1, 3, 4, 2).reshape(Fr * N, H, W, Cin).contiguous()

    def pack(self, plan, w, b=None, **forms):                 # (forms: which kernel layouts HipOps packs - the double has one)
        weff = effective_weight(plan, w.detach())
        if getattr(plan, 'f16w', False):                      # (the f16 MFMA form: IEEE-half weights, the bf16 inputs convert exactly)
            weff = weff.half().float()
        elif getattr(plan, 'bf16', False):
            weff = weff.bfloat16().float()
        bp = None
        if b is not None and not plan.transposed:
            bp = torch.zeros(plan.Npad, dtype=torch.float32, device=self.device)
            for n, cm in enumerate(plan.colmap):
                if cm >= 0:
                    bp[n] = b[cm]
        self._w[id(plan)] = (weff, bp)

    @staticmethod
    def lstm_bwd_fusable(plan, cx, hd):
        # the same rule as HipOps (the double follows the product's scheduling decisions); RNH_FUSE_ANY=1 lets the CPU tests drive
        # the fused schedule through nets of any width
        import os
        if os.environ.get('RNH_FUSE_GATES_BWD', '1') == '0':
            return False
        if os.environ.get('RNH_FUSE_ANY') == '1':
            return plan.ntaps == 9 and plan.epilogue == L.EPI_STORE
        tile = 64 if plan.Npad % 128 else 128
        return bool(getattr(plan, 'bf16', False) and plan.ntaps == 9 and plan.epilogue == L.EPI_STORE and plan.Npad == tile and
                    cx + hd <= tile and cx % 8 == 0 and hd % 8 == 0)

    def pair_cells(self, N, H, W):
        return os.environ.get('RNH_PAIR') == '1'                 # (the double has no launches to save: pairing only when a test asks for it)

    def conv_pair(self, calls):
        for pl, srcs, B, H, W, kw in calls:
            self.conv(pl, srcs, B, H, W, **kw)

    def conv(self, plan, srcs, B, H, W, dsts=None, ps=None, lstm=None, lstm_bwd=None):
        if lstm_bwd is not None:
            # the data gradient with the gate backward of the chain's next frame behind it: dh_rec = the hd columns after the input
            # gradient, rounded to rec_dtype (the element type the unfused path stores it in)
            hd, b = lstm_bwd['hd'], lstm_bwd
            rec = torch.empty(B, H, W, hd, dtype=b['rec_dtype'], device=self.device)
            self.conv(plan, srcs, B, H, W, dsts=[dsts[0], Dst(rec, hd)])
            self.lstm_gates_bwd(b['dh'], b.get('dc_next'), b['gates'], b.get('c_prev'), b['c_next'], b['dgates'], b.get('dc_prev'), dh2=rec)
            return
        weff, bp = self._w[id(plan)]
        bf = getattr(plan, 'bf16', False)
        x = torch.cat([self._r(gather_src(s, B), bf) for s in srcs], dim=-1).permute(0, 3, 1, 2)
        assert x.shape[1] == weff.shape[1], (plan.name, x.shape, weff.shape)
        y = F.conv2d(x, weff, bp if plan.bkey is not None else None, padding=1 if plan.ntaps == 9 else 0)
        y = y.permute(0, 2, 3, 1)                     # (B, H, W, Npad)
        if plan.epilogue == L.EPI_STORE:
            col = 0
            for d in dsts:
                tgt = d.t[d.img_off:d.img_off + B, ..., d.c0:d.c0 + d.ncols]
                val = y[..., col:col + d.ncols]
                if d.accumulate:
                    tgt.copy_(tgt.float() + val)
                else:
                    tgt.copy_(val)
                col += d.ncols
        elif plan.epilogue == L.EPI_PS:
            t, r = ps
            cq = t.shape[-1]
            v = y[..., :cq * r * r].reshape(B, H, W, r, r, cq)          # column = (i*r + j)*cq + c
            t.copy_(v.permute(0, 1, 3, 2, 4, 5).reshape(B, H * r, W * r, cq))
        else:
            hd = lstm['hd']
            # undo the plan's gate layout (plans.lstm_colmap / lstm_colmap64 / lstm_colmap8): column n holds reference output channel colmap[n]
            cm = torch.tensor(plan.colmap, dtype=torch.long, device=y.device)
            keep = cm >= 0
            v = torch.empty(B, H, W, 4 * hd, dtype=y.dtype, device=y.device)
            v[..., cm[keep]] = y[..., keep]
            v = v.reshape(B, H, W, 4, hd)
            gi, gf, go = torch.sigmoid(v[..., 0, :]), torch.sigmoid(v[..., 1, :]), torch.sigmoid(v[..., 2, :])
            gg = torch.tanh(v[..., 3, :])
            cp = lstm.get('c_prev')
            cn = gf * cp + gi * gg if cp is not None else gi * gg
            lstm['c_out'].copy_(cn)
            lstm['h_out'].copy_(go * torch.tanh(cn))
            if lstm.get('gates_out') is not None:
                lstm['gates_out'].copy_(torch.cat([gi, gf, go, gg], dim=-1))

    def wgrad(self, plan: WgradPlan, xsrcs, ysrcs, B, H, W, dw, db=None, accumulate=False, vsrcs=None, vN=None):
        bf = getattr(plan, 'bf16', False)
        x = torch.cat([self._r(gather_src(s, B), bf) for s in xsrcs], dim=-1).permute(0, 3, 1, 2)
        dy = torch.cat([self._r(gather_src(s, B), bf) for s in ysrcs], dim=-1).permute(0, 3, 1, 2)
        kh = 3 if plan.ntaps == 9 else 1
        w0 = torch.zeros(dy.shape[1], x.shape[1], kh, kh, dtype=torch.float32, device=self.device, requires_grad=True)
        with torch.enable_grad():
            F.conv2d(x.detach(), w0, padding=1 if kh == 3 else 0).backward(dy.detach())
        g = w0.grad
        # like the kernels' reductions: only the (co, ci) entries the plan maps are written (stored or accumulated)
        bsum = dy.sum(dim=(0, 2, 3))
        for j, co in enumerate(plan.colmap[:dy.shape[1]]):
            if co < 0:
                continue
            if db is not None:
                db[co] = db[co] + bsum[j] if accumulate else bsum[j]
            for i, ci in enumerate(plan.rowmap[:x.shape[1]]):
                if ci >= 0:
                    dw[co, ci] = dw[co, ci] + g[j, i] if accumulate else g[j, i]

    def inconv_fwd(self, x, w, b, slope):
        y = F.prelu(F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1), slope)
        return y.permute(0, 2, 3, 1).contiguous()

    def inconv_bwd(self, x, w, b, slope, dy, dw, db, dslope, accumulate=False):
        w_, b_, a_ = (t.detach().clone().requires_grad_(True) for t in (w, b, slope))
        with torch.enable_grad():
            y = F.prelu(F.conv2d(x.permute(0, 3, 1, 2), w_, b_, padding=1), a_)
            y.backward(dy.permute(0, 3, 1, 2))
        for tgt, g in ((dw, w_.grad), (db, b_.grad), (dslope, a_.grad)):
            if accumulate:
                tgt += g
            else:
                tgt.copy_(g)

    def outconv_fwd(self, x, w, b, out=None):
        y = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1)
        if out is None:
            return y.contiguous()
        out.copy_(y)
        return out

    def outconv_dgrad(self, dy, w):
        return F.conv_transpose2d(dy.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1).contiguous()

    def outconv_wgrad(self, x, dy, dw, db, accumulate=False):
        w_ = torch.zeros_like(dw).requires_grad_(True)
        with torch.enable_grad():
            F.conv2d(x.permute(0, 3, 1, 2), w_, padding=1).backward(dy.permute(0, 3, 1, 2))
        gb = dy.sum(dim=(0, 1, 2))
        if accumulate:
            dw += w_.grad
            db += gb
        else:
            dw.copy_(w_.grad)
            db.copy_(gb)

    # ---- collapsed upsampler tail (semantics of csrc/uptail.hip, written independently with torch ops) ----------
    @staticmethod
    def _xcol_windows(srcs, N, J, nwin):
        feats = []
        for i in range(nwin):
            parts = []
            for j in range(J):
                sl = slice((i + j) * N, (i + j + 1) * N)
                parts += [srcs[0][sl], srcs[1][sl], srcs[2][sl][..., :1]]
            feats.append(torch.cat(parts, dim=-1).permute(0, 3, 1, 2))
        return feats

    def refine_xcol_fwd(self, srcs, w1, b1, R1, N, J, cl):
        co, nwin = 2 * cl, R1.shape[0] // N
        for i, f in enumerate(self._xcol_windows(srcs, N, J, nwin)):
            R1[i * N:(i + 1) * N, ..., co] = F.conv2d(f, w1[co:co + 1], b1[co:co + 1], padding=1)[:, 0]
            R1[i * N:(i + 1) * N, ..., co + 1:co + 4] = 0

    def refine_phase_bias(self, R1, P4, w1, N, J, cl, ncols):
        nwin = R1.shape[0] // N
        for i in range(nwin):
            x = torch.cat([P4[(i + j) * N:(i + j + 1) * N, ..., :1] for j in range(J)], dim=-1).permute(0, 3, 1, 2)
            wp = w1[:ncols, [j * (2 * cl + 1) + 2 * cl for j in range(J)]]
            R1[i * N:(i + 1) * N, ..., :ncols] += F.conv2d(x, wp, None, padding=1).permute(0, 2, 3, 1)

    def refine_xcol_wgrad(self, srcs, dy, dw1, db1, N, J, cl, accumulate):
        co, nwin = 2 * cl, dy.shape[0] // N
        w0 = torch.zeros(1, dw1.shape[1], 3, 3, device=self.device, requires_grad=True)
        b0 = torch.zeros(1, device=self.device, requires_grad=True)
        with torch.enable_grad():
            for i, f in enumerate(self._xcol_windows(srcs, N, J, nwin)):
                F.conv2d(f.detach(), w0, b0, padding=1).backward(dy[i * N:(i + 1) * N, ..., co].unsqueeze(1))
        if accumulate:
            dw1[co] += w0.grad[0]
            db1[co] += b0.grad[0]
        else:
            dw1[co] = w0.grad[0]
            db1[co] = b0.grad[0]

    def refine_phase_wgrad(self, dy, P4, dw1, N, J, cl, ncols, accumulate):
        nwin = dy.shape[0] // N
        cs, c0 = 2 * cl + 1, 2 * cl
        g = torch.zeros(ncols, J, 3, 3, device=self.device)
        for i in range(nwin):
            x = torch.cat([P4[(i + j) * N:(i + j + 1) * N, ..., :1] for j in range(J)], dim=-1).permute(0, 3, 1, 2)
            w0 = torch.zeros(ncols, J, 3, 3, device=self.device, requires_grad=True)
            with torch.enable_grad():
                F.conv2d(x, w0, padding=1).backward(dy[i * N:(i + 1) * N, ..., :ncols].permute(0, 3, 1, 2))
            g += w0.grad
        for j in range(J):
            if accumulate:
                dw1[:ncols, j * cs + c0] += g[:, j]
            else:
                dw1[:ncols, j * cs + c0] = g[:, j]

    def conv_to_column(self, x, w, col, out, c0, yzero=0):
        d = F.conv_transpose2d(x.permute(0, 3, 1, 2), w[:, col:col + 1], padding=1).permute(0, 2, 3, 1)
        out[..., c0:c0 + 1] = d
        if yzero:
            out[..., c0 + 1:c0 + 1 + yzero] = 0

    def refine_xcol_dgrad(self, g, w1, dHf, dHb, N, J, cl):
        T = dHf.shape[0] // N
        cs, c = 2 * cl + 1, 2 * cl
        for f in range(T):
            for j in range(J):
                gj = g[(f + J - 1 - j) * N:(f + J - j) * N, ..., c:c + 1].permute(0, 3, 1, 2)
                wj = w1[c:c + 1, j * cs:j * cs + 2 * cl]                                  # (1, 2*cl, 3, 3)
                d = F.conv_transpose2d(gj, wj, padding=1).permute(0, 2, 3, 1)
                dHf[f * N:(f + 1) * N] += d[..., :cl]
                dHb[f * N:(f + 1) * N] += d[..., cl:]

    def xcol_combine_m(self, z, b1, R1, N, J, c0):
        nwin = R1.shape[0] // N
        for i in range(nwin):
            v = b1[c0] + sum(z[(i + j) * N:(i + j + 1) * N, ..., j] for j in range(J))
            R1[i * N:(i + 1) * N, ..., c0:c0 + 8] = 0
            R1[i * N:(i + 1) * N, ..., c0] = v.to(R1.dtype)

    def xcol_gather_m(self, dy, N, J, c, dtype):
        nwin, H, W = dy.shape[0] // N, dy.shape[1], dy.shape[2]
        E = torch.zeros((nwin + J - 1) * N, H, W, 8, device=self.device)
        for f in range(nwin + J - 1):
            for j in range(J):
                if 0 <= f - j < nwin:
                    E[f * N:(f + 1) * N, ..., j] = dy[(f - j) * N:(f - j + 1) * N, ..., c].float()
        return E.to(dtype)

    @staticmethod
    def put_scalar(dst, src, accumulate):
        if accumulate:
            dst.add_(src)
        else:
            dst.copy_(src)

    def uptail_fwd(self, y1, w2, b2, w3, b3, r, out):
        z = F.pixel_shuffle(F.conv2d(y1.float().permute(0, 3, 1, 2), w2, b2, padding=1), r)
        out.copy_(F.conv2d(z, w3, b3, padding=1).permute(0, 2, 3, 1))
        return out

    @staticmethod
    def uptail_fwd_supported(r, Co):
        return r in (2, 3) and Co == 1

    def uptail_compose(self, w2, w3, r):
        return dict(w2=w2, w3=w3, r=r)

    def uptail_bf16_supported(self, C1, r, Co):
        return C1 == 64 and r == 2 and Co == 1

    def uptail_dgrad(self, d_o, G, C1, r, dtype=torch.float32):
        dy2 = F.conv_transpose2d(d_o.permute(0, 3, 1, 2), G['w3'], padding=1)             # (B, Cq, rH, rW)
        dz = F.pixel_unshuffle(dy2, r)                                                      # channel c*r*r + i*r + j
        return F.conv_transpose2d(dz, G['w2'], padding=1).permute(0, 2, 3, 1).contiguous().to(dtype)

    def uptail_xcorr_supported(self, C1, r, Co):
        return Co == 1 and r in (2, 3) and C1 % 64 == 0

    def uptail_xcorr(self, y1, d_o, r, out=None):
        nd2 = (r + 2) * (r + 2)
        D = self.uptail_expand(d_o, r)[..., :nd2]                                        # (B, Hm, Wm, ND*ND)
        ypad = F.pad(y1.float(), (0, 0, 1, 1, 1, 1))
        Hm, Wm = y1.shape[1], y1.shape[2]
        M = torch.zeros(nd2, y1.shape[3], 3, 3, device=self.device)
        for ty in range(3):
            for tx in range(3):
                M[:, :, ty, tx] = torch.einsum('bhwd,bhwc->dc', D, ypad[:, ty:ty + Hm, tx:tx + Wm])
        if out is not None:
            out[0].copy_(M)
            out[1].copy_(D.sum(dim=(0, 1, 2)))
            return out
        return M, D.sum(dim=(0, 1, 2))

    def uptail_expand(self, d_o, r, Dc=None):
        B, Hh, Wh, Co = d_o.shape
        nd = r + 2
        dc = (Co * nd * nd + 3) // 4 * 4 if Dc is None else Dc
        pad = F.pad(d_o.permute(0, 3, 1, 2), (1, r, 1, r))                                  # index p + 1; zeros outside
        D = torch.zeros(B, Hh // r, Wh // r, dc, device=self.device)
        for co in range(Co):
            for dyi in range(nd):
                for dxi in range(nd):
                    D[..., co * nd * nd + dyi * nd + dxi] = pad[:, co, dyi::r, dxi::r][:, :Hh // r, :Wh // r]
        return D

    def uptail_wcontract(self, M, S, w2, b2, w3, dw2, db2, dw3, db3, r, acc2, acc3):
        Co, Cq = w3.shape[0], w3.shape[1]
        C1, nd = w2.shape[1], r + 2
        g2, gb2, g3, gb3 = torch.zeros_like(dw2), torch.zeros_like(db2), torch.zeros_like(dw3), torch.zeros_like(db3)
        Mv = M.reshape(Co, nd, nd, C1, 3, 3)
        Sv = S[:Co * nd * nd].reshape(Co, nd, nd)
        w2v = w2.reshape(Cq, r, r, C1, 3, 3)
        b2v = b2.reshape(Cq, r, r)
        for i in range(r):
            for j in range(r):
                for ty in (-1, 0, 1):
                    for tx in (-1, 0, 1):
                        Md, Sd = Mv[:, i - ty + 1, j - tx + 1], Sv[:, i - ty + 1, j - tx + 1]     # (Co, C1, 3, 3), (Co,)
                        w3t = w3[:, :, ty + 1, tx + 1]                                               # (Co, Cq)
                        g2.view(Cq, r, r, C1, 3, 3)[:, i, j] += torch.einsum('oc,oxyz->cxyz', w3t, Md)
                        gb2.view(Cq, r, r)[:, i, j] += torch.einsum('oc,o->c', w3t, Sd)
                        g3[:, :, ty + 1, tx + 1] += torch.einsum('cxyz,oxyz->oc', w2v[:, i, j], Md) + Sd[:, None] * b2v[None, :, i, j]
                gb3 += Sv[:, i + 1, j + 1]
        for tgt, g, a in ((dw2, g2, acc2), (db2, gb2, acc2), (dw3, g3, acc3), (db3, gb3, acc3)):
            if a:
                tgt += g
            else:
                tgt.copy_(g)

    def lstm_gates_bwd(self, dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, dh2=None):
        hd = dh.shape[-1]
        dh = dh.float()
        if dh2 is not None:
            dh = dh + dh2.float()
        gates = gates.float()
        gi, gf, go, gg = (gates[..., k * hd:(k + 1) * hd] for k in range(4))
        th = torch.tanh(c_next)
        dct = dh * go * (1 - th * th)
        if dc_next is not None:
            dct = dct + dc_next
        cp = c_prev if c_prev is not None else torch.zeros_like(dh)
        dgates.copy_(torch.cat([dct * gg * gi * (1 - gi), dct * cp * gf * (1 - gf), dh * th * go * (1 - go),
                                dct * gi * (1 - gg * gg)], dim=-1))
        if dc_prev is not None:
            dc_prev.copy_(dct * gf)

    def add(self, out, a, b=None, c=None, accumulate=False):
        v = a.float().clone()
        if b is not None:
            v = v + b.float()
        if c is not None:
            v = v + c.float()
        if accumulate:
            out.copy_(out.float() + v)
        else:
            out.copy_(v)
        return out

    def phase_plane(self, pos, N, Fr, H, W, dtype=torch.float32, channels=4):
        p = pos.reshape(N, Fr).to(self.device).t().reshape(Fr * N, 1, 1, 1).expand(Fr * N, H, W, 1)
        return torch.cat([p, torch.zeros(Fr * N, H, W, channels - 1, device=self.device)], dim=-1).contiguous().to(dtype)

    def loss(self, o, y, G, T, kind, eps, gscale=None, want_grad=False):
        per = y.numel() // T
        d = o.reshape(G, T, per) - y.reshape(1, T, per)
        if kind == L.LOSS_L1:
            val, der = d.abs(), torch.sign(d)
        else:
            r = torch.sqrt(d * d + eps)
            val, der = r, d / r
        loss = val.mean(dim=-1).reshape(G * T)
        d_o = (der * gscale.reshape(G, T, 1) / per).reshape(o.shape) if want_grad else None
        return loss, d_o
