"""HIP backend of the RefineNet engine: thin tensor-level wrappers over the C ABI (include/refinenet_hip.h).

PyTorch is used only for device memory (``torch.empty`` on the current device) and for the current HIP
stream; every computation below is one of the hand-written gfx950 kernels.  No fallback: tensors that are
not fp32 / contiguous / on a HIP device raise.
"""
import ctypes as C
import os

import torch

from . import lib as L
from .plans import ConvPlan, Dst, Src, WgradPlan


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def packed_view(tensors, device=None):
    """list[K] of equally shaped tensors -> (K, ...) fp32 tensor on ``device``: without a copy when the list items are
    consecutive contiguous slices of one buffer (what hipvsr.cine_cache hands over), through torch.stack otherwise."""
    t0 = tensors[0]
    dev = torch.device(device) if device is not None else t0.device
    n = t0.numel() * t0.element_size()
    # the same storage is required as well: two separate allocations may be neighbours in the caching allocator's pool
    base = t0.untyped_storage().data_ptr()
    if (t0.dtype == torch.float32 and t0.device == dev and n > 0
            and all(t.shape == t0.shape and t.dtype == t0.dtype and t.device == t0.device and t.is_contiguous()
                    and t.untyped_storage().data_ptr() == base
                    and t.data_ptr() == t0.data_ptr() + k * n for k, t in enumerate(tensors))):
        return t0.as_strided((len(tensors),) + tuple(t0.shape), (t0.numel(),) + tuple(t0.stride()))
    return torch.stack([t.to(dev, torch.float32) for t in tensors], dim=0)


# Side streams are a per-process resource: every HipOps instance of a device uses the SAME ones (role by role: LSTM bank and index, the
# helper).  A second engine in the same process - bench.py's bf16 case behind its fp32 case, a predictor beside a trainer - that made
# streams of its own ran its step 5 % slower than the same step in a fresh process (86.3 against 82.1 ms, single kernels equally fast:
# the later streams share hardware queues less favourably); with shared streams it does not.  Engines of one process issue their work from
# the same current stream and order every use of a side stream by events, so sharing them changes no dependency.
_SHARED_STREAMS = {}


def _shared_stream(device, role):
    key = (str(device), role)
    st = _SHARED_STREAMS.get(key)
    if st is None:
        st = _SHARED_STREAMS[key] = torch.cuda.Stream(device)
    return st


def touch_side_streams(device):
    """Use every shared side stream once, in the order a training step first uses them (forward LSTM layers, helper, backward LSTM layers).
    ROCm binds a stream to one of its 4 hardware queues at first use, and which streams share a queue decides what overlaps (DESIGN.md 4d d):
    a process that makes other streams first - RCCL's, when a data-parallel run initialises its process group - calls this BEFORE it does so, and
    keeps the pairing of the single-GPU run (measured under torchrun, one rank: 308.3 -> 306.2 ms fp32, 83.9 -> 83.1 ms bf16)."""
    device = torch.device(device)
    z = torch.zeros(64, dtype=torch.float32, device=device)
    for role in [('lstm', 0, 0), ('lstm', 0, 1), ('lstm', 0, 2), ('helper', 0), ('lstm', 1, 0), ('lstm', 1, 1), ('lstm', 1, 2)]:
        with torch.cuda.stream(_shared_stream(device, role)):
            z.add_(0)
    torch.cuda.synchronize(device)


class HipOps:
    name = 'hip'

    def __init__(self, device, direct=None):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise L.HipKernelError(f'the HIP backend needs a HIP device, got {device}')
        self.lib = L.load()
        self._maps = {}        # id(plan) -> dict of device int32 arrays
        self._packed = {}      # id(plan) -> (wp, biasp)
        self._maps44, self._packed44 = {}, {}      # the same for the F(4x4, 3x3) form of the ConvLSTM cell plans
        self._ws = {}
        self._ws_retired = []   # outgrown scratch buffers a captured HIP graph may still address (see _workspace)
        self.graph_captures = 0
        self._check_halo = os.environ.get('RNH_CHECK_HALO', '0') == '1'
        # side streams of the ConvLSTM wavefront, in two banks: the forward uses bank 0, the backward bank 1.  Under HIP-graph
        # capture a stream remembers the streams whose events it waited on; hipStreamEndCapture (ROCm 7.2) walks those
        # references recursively and never returns if two streams reference EACH OTHER.  The forward makes layer l's stream
        # wait on layer l-1's, the backward layer l's on layer l+1's: on the same streams, a captured training step
        # (hipvsr.graph.GraphedTrainStep) would form exactly such a cycle (found with rocgdb: 174 000 frames of
        # hip::Stream::EndCapture).  The 'one' mapping of RNH_LSTM_STREAMS (a stream waiting on its own event) has the
        # same problem and is for eager measurements only.
        self._banks = ([], [])
        self._helper, self._helper_used, self._fence_buf = None, False, None      # aside() / rejoin() / fence()
        self._side = self._banks[0]
        self._fork_n = 2
        self._zero_page = torch.zeros(64, dtype=torch.float32, device=self.device)      # what masked wgrad lanes read
        # RNH_DIRECT=0 selects the LDS-staged variant of rnh_conv_igemm (kept for A/B measurements)
        self.direct = (os.environ.get('RNH_DIRECT', '1') != '0') if direct is None else bool(direct)
        self.direct_ps = os.environ.get('RNH_DIRECT_PS', '1') != '0'
        self.wino_wgrad = os.environ.get('RNH_WINO', '1') != '0' and os.environ.get('RNH_WINO_WGRAD', '1') != '0'
        # experiment (DESIGN 4d d): the order in which the side streams are first USED decides which of them share a hardware queue (ROCm binds
        # a stream to one of GPU_MAX_HW_QUEUES = 4 queues); RNH_STREAM_TOUCH="H,F0,F1,F2,B0,B1,B2" submits one trivial launch on each, in that order
        if os.environ.get('RNH_STREAM_TOUCH'):
            for role in os.environ['RNH_STREAM_TOUCH'].split(','):
                st = _shared_stream(self.device, ('helper', 0)) if role == 'H' else _shared_stream(self.device, ('lstm', 0 if role[0] == 'F' else 1, int(role[1])))
                with torch.cuda.stream(st):
                    self._zero_page.add_(0)
            torch.cuda.synchronize(self.device)

    # ---- memory -------------------------------------------------------------------------------------
    def empty(self, *shape, dtype=torch.float32):
        if os.environ.get('RNH_POISON'):
            # debugging aid: every buffer starts as NaN, so a kernel that reads an element nobody wrote shows up in the
            # results instead of depending on what the allocator's block held before
            return torch.full(shape if not (len(shape) == 1 and isinstance(shape[0], (tuple, list))) else tuple(shape[0]),
                              float('nan'), dtype=dtype, device=self.device)
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def total_memory(self):
        """HBM bytes of the device (what the engine's 'auto' gate-memory plan is sized against)."""
        return torch.cuda.get_device_properties(self.device).total_memory

    def capturing(self):
        """Is the current stream being captured into a HIP graph?"""
        return torch.cuda.is_current_stream_capturing()

    def memory_budget(self):
        """What this process can still use of the device: free memory + the pool torch has reserved for this process (other processes on the card
        and everything this process holds outside torch's allocator are not ours to plan with); never more than the device has."""
        free, total = torch.cuda.mem_get_info(self.device)
        return min(total, free + torch.cuda.memory_reserved(self.device))

    def quiesce(self):
        """After an exception in the middle of a forward / backward: nothing may still run on the side streams or the helper stream when the buffers
        they use go back to the allocator.  Outside a graph capture the device is drained; the helper-stream bookkeeping starts afresh."""
        try:
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize(self.device)
        finally:
            self._helper_used = False

    def mem_allocated(self):
        return torch.cuda.memory_allocated(self.device)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, *ts, mixed=False):
        """fp32 (mixed=True: fp32 or bf16 - the entry points of the bf16-storage path), contiguous, on this device."""
        for t in ts:
            if t is None:
                continue
            ok = t.dtype == torch.float32 or (mixed and t.dtype == torch.bfloat16)
            if not ok or not t.is_contiguous() or t.device != self.device:
                raise L.HipKernelError(f'expected contiguous {"fp32 / bf16" if mixed else "fp32"} tensors on {self.device}, got {t.dtype} '
                                       f'contiguous={t.is_contiguous()} on {t.device}')

    def _i32(self, lst):
        return torch.tensor(lst, dtype=torch.int32, device=self.device)

    # ---- streams: the two LSTM directions run on side streams between fork() and join() ---------------------
    # The ConvLSTM wavefront asks for 2L logical streams, one per (direction, layer).  RNH_LSTM_STREAMS maps them onto
    # HIP streams: 'cell' = 2L streams, 'layer' = L streams (both directions of a layer share one: measured best at
    # BASELINE config 2, where every cell launch fills the chip and cross-stream event waits only cost latency),
    # 'dir' = 2, 'one' = 1.
    def _side_index(self, i):
        mode = os.environ.get('RNH_LSTM_STREAMS', 'layer')
        half = max(self._fork_n // 2, 1)
        return {'cell': i, 'layer': i % half, 'dir': i // half, 'one': 0}[mode]

    def fork(self, n, bank=0):
        self._side = self._banks[bank]
        self._fork_n = n
        n = 1 + max(self._side_index(i) for i in range(n))
        while len(self._side) < n:
            self._side.append(_shared_stream(self.device, ('lstm', bank, len(self._side))) if os.environ.get('RNH_SHARED_STREAMS', '1') != '0'
                              else torch.cuda.Stream(self.device))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        for st in self._side[:n]:
            st.wait_event(ev)

    def side(self, i):
        return torch.cuda.stream(self._side[self._side_index(i)])

    def record(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def wait(self, ev):
        torch.cuda.current_stream(self.device).wait_event(ev)

    def join(self, n):
        n = 1 + max(self._side_index(i) for i in range(n))
        cur = torch.cuda.current_stream(self.device)
        for st in self._side[:n]:
            ev = torch.cuda.Event()
            ev.record(st)
            cur.wait_event(ev)

    # ---- a helper stream for work that nothing on the critical path waits for (weight gradients, the upsampler of a finished stage) -----
    # aside(): the helper stream picks up behind everything the current stream has been given so far (so it may read what those
    # launches produce) and the launches inside the block go to it; rejoin(): the current stream waits for everything the helper has
    # been given.  Buffers the helper reads or writes must stay alive - and must not be rewritten by the current stream - until the
    # next rejoin().  RNH_ASIDE=0: the block runs on the current stream (A/B measurements).  Under HIP-graph capture the helper is an ordinary
    # branch of the captured graph (it allocates nothing).  History: mid-round 4 a captured training step with this branch reproduced the eager
    # gradients only intermittently on some boxes (5 of 12 runs on one MI355X, 0 of 80 on another; ~1 % off downstream of refine conv1's gradient
    # planes) and the branch was kept out of captures; the cause turned out to be the direct implicit-GEMM kernel's folded loop tails (stale operand
    # copies under cross-stream memory load, DESIGN.md 4d e) - with that fixed, 58 of 58 graph-vs-eager runs on four boxes agree bit for bit.
    # RNH_ASIDE_CAPTURE=0 keeps the blocks on the capturing stream.
    def aside(self, tag=''):
        if os.environ.get('RNH_ASIDE', '1') == '0' or (tag and tag in os.environ.get('RNH_ASIDE_OFF', '').split(',')) or \
                (torch.cuda.is_current_stream_capturing() and os.environ.get('RNH_ASIDE_CAPTURE', '1') == '0'):
            import contextlib
            return contextlib.nullcontext()
        if self._helper is None:
            self._helper = _shared_stream(self.device, ('helper', 0)) if os.environ.get('RNH_SHARED_STREAMS', '1') != '0' else torch.cuda.Stream(self.device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._helper.wait_event(ev)
        self._helper_used = True
        if os.environ.get('RNH_ASIDE_DELAY'):                 # race detector (tests): the helper starts late by that many spin cycles, so a
            with torch.cuda.stream(self._helper):             # launch elsewhere that does not wait for it - or rewrites what it reads - shows
                torch.cuda._sleep(int(os.environ['RNH_ASIDE_DELAY']))
        return torch.cuda.stream(self._helper)

    def rejoin(self):
        if self._helper_used:
            ev = torch.cuda.Event()
            ev.record(self._helper)
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._helper_used = False

    def fence(self):
        """One trivial launch on the current stream behind everything it has waited for.  A stream that joins side streams and then launches
        nothing leaves their last launches as additional SINK nodes of a captured HIP graph; the replays of such a graph did not always
        hold back the launching stream's later work (the optimizer step) until every sink had finished (ROCm 7.2; tools/probes/graph_dot.py)."""
        if self._fence_buf is None:
            self._fence_buf = torch.zeros(8, dtype=torch.float32, device=self.device)
        b = self._fence_buf
        L.check(self.lib.rnh_ew_add(_ptr(b[4:]), _ptr(b[:4]), None, None, 4, 0, self._stream()), 'rnh_ew_add(fence)')

    def _workspace(self, key, nfloats):
        """Scratch buffer ``key`` of the current stream, grown on demand.  A HIP graph captured through this object
        (hipvsr.graph) has the addresses of the buffers it used baked in: once a graph exists an outgrown buffer is
        retired, not freed, so that no replay can ever write into memory the allocator has handed to someone else."""
        key = (key, torch.cuda.current_stream(self.device).cuda_stream)      # one scratch buffer per stream
        t = self._ws.get(key)
        if t is None or t.numel() < nfloats:
            if t is not None and self.graph_captures:
                self._ws_retired.append(t)
            if os.environ.get('RNH_WS_GUARD') == '1':         # debugging aid: 256 KiB of sentinel behind (and in front of) every scratch buffer
                g = 65536
                full = self.empty(int(nfloats) + 2 * g)
                full.fill_(-12345.0)
                self.__dict__.setdefault('_ws_guard', {})[key] = (full, g, int(nfloats))
                t = full[g:g + int(nfloats)]
            else:
                t = self.empty(int(nfloats))
            self._ws[key] = t
        return t

    def check_ws_guards(self):
        """RNH_WS_GUARD=1: names of the scratch buffers whose guard bands have been written to (device sync)."""
        bad = []
        for key, (full, g, n) in getattr(self, '_ws_guard', {}).items():
            lo, hi = full[:g], full[g + n:]
            if bool((lo != -12345.0).any()) or bool((hi != -12345.0).any()):
                bad.append((key, int((lo != -12345.0).sum()), int((hi != -12345.0).sum()), n))
        return bad

    def halo_buffer(self, key, shape, dtype, lo, hi):
        """A buffer of ``shape`` whose leading-dimension slices outside [lo, hi) are zero and stay zero: allocated and zeroed
        once per (key, shape, dtype), handed out again on every call - the caller writes [lo, hi) only, and everything that
        reads it is ordered behind the previous use on the same stream.  (The refine block's gradient planes with their
        window halo: zeroing the halo of a fresh buffer was six 138 MB fill launches per step.)
        One buffer per key: a call with another shape (a partial last batch, another crop size) REPLACES the previous one
        instead of pinning both for the life of the process - unless a HIP graph has been captured through this object,
        whose replays still address the old buffer: then it is retired, not freed.  RNH_CHECK_HALO=1 (debug): every reuse
        first checks that the halo is still all zero (a device sync per call)."""
        k, sig = ('halo', key), (tuple(shape), dtype, lo, hi)
        ent = self._ws.get(k)
        if ent is not None and ent[0] != sig:
            if self.graph_captures:
                self._ws_retired.append(ent[1])
            ent = None
        if ent is None:
            ent = (sig, self.zeros(*shape, dtype=dtype))
            self._ws[k] = ent
        elif self._check_halo:
            t = ent[1]
            if bool(t[:lo].any()) or bool(t[hi:].any()):
                raise RuntimeError(f'halo_buffer {key!r}: the zero halo outside [{lo}, {hi}) has been written to')
        return ent[1]

    def stack_inputs(self, inputs):
        """list[F] of (N, Cin, H, W) -> (F*N, H, W, Cin) NHWC, frame-major (plumbing: one copy of the LR input)."""
        x = packed_view(inputs, self.device)                                                # (F, N, Cin, H, W)
        F, N, Cin, H, W = x.shape
        return x.permute(0, 1, 3, 4, 2).reshape(F * N, H, W, Cin).contiguous()

    # ---- weights ------------------------------------------------------------------------------------
    def _plan_maps(self, plan):
        m = self._maps.get(id(plan))
        if m is None:
            if isinstance(plan, ConvPlan):
                m = dict(kbase=self._i32(plan.kbase), knv=self._i32(plan.knv), ktap=self._i32(plan.ktap),
                         kcoff=self._i32(plan.kcoff), colmap=self._i32(plan.colmap))
                if plan.wino:
                    m.update(wkbase=self._i32(plan.wkbase), wknv=self._i32(plan.wknv), wkcoff=self._i32(plan.wkcoff))
            else:
                m = dict(rowmap=self._i32(plan.rowmap), colmap=self._i32(plan.colmap), xgrp=self._i32(plan.xgrp),
                         ygrp=self._i32(plan.ygrp))
                if plan.bf16:
                    m.update(rowmap64=self._i32(plan.rowmap64), colmap64=self._i32(plan.colmap64))
            m['_plan'] = plan          # keeps id(plan) unique while cached
            self._maps[id(plan)] = m
        return m

    def pack(self, plan: ConvPlan, w, b=None, f22=True, f44=None):
        """Re-lay the OIHW weight (and bias) of ``plan`` into the kernel's [nk][Npad][16] slabs (f22: the plan's own form - direct, F(2x2, 3x3) or
        bf16) and / or into the F(4x4, 3x3) kernels' layout (f44; None = wherever the plan is eligible for that form).  The engine asks for the one
        form the step launches the plan in (hipvsr/forms.py); a caller that packs a plan by hand gets both."""
        if f22:
            self._pack(plan, w, b)
        if f44 if f44 is not None else getattr(plan, 'wino44', False):
            self._pack44(plan, w, b)

    # ---- the ConvLSTM cell in Winograd form F(4x4, 3x3): rnh_wino44_* (csrc/conv_wino44.hip) ---------------------------------------------
    def _pack44(self, plan, w, b):
        from .plans import lstm_colmap64
        m = self._maps44.get(id(plan))
        if m is None:
            kch = [sg.kbase + c if c < sg.nvalid else -1 for sg in plan.ksegs for c in range(sg.nch)]
            # the cell's columns in blocks of 64 = the gates i, f | o, g of 16 hidden channels; any other plan's columns as they are, padded to 64
            cm = lstm_colmap64(plan.Cout // 4) if plan.epilogue == L.EPI_LSTM else list(plan.colmap) + [-1] * (-len(plan.colmap) % 64)
            kco = [sg.kcoff for sg in plan.ksegs for c in range(sg.nch)]
            m = self._maps44[id(plan)] = dict(kch=self._i32(kch), kcoff=self._i32(kco) if any(kco) else None, colmap=self._i32(cm), K=len(kch), Npad=len(cm), _plan=plan)
        K, Npad = m['K'], m['Npad']
        buf = self._packed44.get(id(plan))
        if buf is None:
            buf = self._packed44[id(plan)] = (self.empty(K // 8 * 36 * Npad * 8), self.empty(Npad))
        L.check(self.lib.rnh_wino44_pack_weights(_ptr(w), _ptr(b), _ptr(buf[0]), _ptr(buf[1]), _ptr(m['kch']), _ptr(m['kcoff']), _ptr(m['colmap']), K, Npad,
                                                 plan.Cout, plan.Cin, int(plan.transposed), self._stream()), f'rnh_wino44_pack_weights({plan.name})')

    def wino44_conv(self, plan, vsrcs, B, H, W, dst=None, ps=None):
        """A plain-store convolution in F(4x4, 3x3) form on transformed sources (rnh_wino44_conv): ``vsrcs`` = (tensor of wino44_transform images,
        first tile block) per K segment of the plan; ``dst``: a Dst or a list of them, as conv()'s dsts, or ``ps`` = (tensor (B, rH, rW, cq), r): the
        PixelShuffle fused into the store, as conv()'s ps."""
        if id(plan) not in self._packed44:
            raise L.HipKernelError(f'{plan.name}: weights were not packed for the F(4x4, 3x3) form')
        if len(vsrcs) != len(plan.ksegs) or len(vsrcs) > 16:
            raise L.HipKernelError(f'{plan.name}: {len(vsrcs)} sources for {len(plan.ksegs)} K segments')
        a = L.Wino44ConvArgs()
        need = int(self.lib.rnh_wino44_v_floats(B, H, W, 16)) // 16             # floats per channel of one launch's worth of tile blocks
        for i, ((v, boff), sg) in enumerate(zip(vsrcs, plan.ksegs)):
            self._chk(v)
            per_block = 36 * 32 * sg.nch
            if boff < 0 or boff * per_block + need * sg.nch > v.numel():
                raise L.HipKernelError(f'{plan.name}: transformed source {i} does not hold the launch\'s tile blocks')
            a.v[i], a.vchunks[i], a.vblock_off[i] = v.data_ptr(), sg.nch // 16, boff
        if ps is not None:
            t, r = ps
            self._chk(t)
            cq = t.shape[-1]
            if tuple(t.shape) != (B, H * r, W * r, cq) or plan.epilogue != L.EPI_PS:
                raise L.HipKernelError(f'{plan.name}: pixel-shuffle destination shape {tuple(t.shape)}')
            a.ps_r, a.ps_cq = r, cq
            a.dst[0].ptr, a.dst[0].C, a.dst[0].c0, a.dst[0].ncols = t.data_ptr(), cq, 0, cq
            dsts, nd = [], 1
        else:
            dsts = [dst] if isinstance(dst, Dst) else list(dst)
            nd = len(dsts)
        wp, bp = self._packed44[id(plan)]
        a.nsrc, a.B, a.H, a.W, a.Npad, a.ndst = len(vsrcs), B, H, W, self._maps44[id(plan)]['Npad'], nd
        a.wp, a.bias = wp.data_ptr(), (bp.data_ptr() if plan.bkey is not None else None)
        for i, d in enumerate(dsts):
            self._chk(d.t)
            if tuple(d.t.shape[1:3]) != (H, W) or d.img_off < 0 or d.img_off + B > d.t.shape[0] or d.c0 + d.ncols > d.t.shape[-1]:
                raise L.HipKernelError(f'{plan.name}: destination {i} geometry')
            a.dst[i].ptr, a.dst[i].C, a.dst[i].c0 = d.t.data_ptr(), d.t.shape[-1], d.c0
            a.dst[i].ncols, a.dst[i].accumulate, a.dst[i].img_off = d.ncols, int(d.accumulate), d.img_off
        L.check(self.lib.rnh_wino44_conv(C.byref(a), self._stream()), f'rnh_wino44_conv({plan.name})')

    def wino44_ok(self, plan, B, H, W, packed=True, dst_channels=0):
        """Does the cell call (plan, B, H, W) run in F(4x4, 3x3) form?  The plan must be eligible and packed for it and the images whole 4x4
        tiles.  No size condition: measured against the F(2x2) kernel on one box each, the step is faster at every launch size tried - BASELINE
        config 4 (8192 workgroups per cell launch) 1742 -> 1495 ms, config 2 (1024) 304 -> 277, config 5 (576) 247.9 -> 230.7, the same at N = 4 /
        N = 2 (288 / 144) 127.8 -> 119.8 / 66.1 -> 62.4, the reference YAML's 16 crops of 32 x 32 (128, graph replay) 42.4 -> 40.8
        (profiles/r05_zb_*).  RNH_WINO44=0 switches the form off; RNH_WINO44_MIN=n asks for launches of at least n workgroups (A/B runs).
        ``packed=False``: the question before the weights are packed (the engine's memory plan).  The kernels' own limits are part of the answer
        (the engine falls back to the F(2x2) launch instead of meeting a refusal): fewer than 2^27 pixels per launch, and fewer than 2^31 elements in
        a destination of ``dst_channels`` channels (the cell: its state, 4 bytes per element of hd channels)."""
        from .forms import wino44_launch_ok
        return bool((not packed or id(plan) in self._packed44) and wino44_launch_ok(plan, B, H, W, dst_channels))

    def wino44_v(self, B, H, W, nch, frames=1):
        """Buffer(s) for the transformed form of ``frames`` tensors (B, H, W, nch): a (frames, floats) tensor."""
        return self.empty(frames, int(self.lib.rnh_wino44_v_floats(B, H, W, nch)))

    def wino44_transform(self, s: Src, B, H, W, out):
        """out = B^T d B of the source's channels (rnh_wino44_transform); the source as conv() takes it (scale 1, no second operand)."""
        t = s.t
        self._chk(t, out)
        nch = t.shape[-1] - s.c0 if s.nch is None else s.nch
        if s.scale != 1 or s.add is not None or tuple(t.shape[1:3]) != (H, W) or s.img_off < 0 or s.img_off + B > t.shape[0]:
            raise L.HipKernelError('wino44_transform: a plain source of the output geometry')
        if out.numel() != int(self.lib.rnh_wino44_v_floats(B, H, W, nch)):
            raise L.HipKernelError('wino44_transform: output size')
        L.check(self.lib.rnh_wino44_transform(t.data_ptr() + s.img_off * H * W * t.shape[-1] * 4, t.shape[-1], s.c0, nch, B, H, W, _ptr(out), self._stream()),
                'rnh_wino44_transform')

    def _wino44_cell_args(self, plan, vsrcs, B, H, W, lstm):
        if id(plan) not in self._packed44:
            raise L.HipKernelError(f'{plan.name}: weights were not packed for the F(4x4, 3x3) form')
        if len(vsrcs) != len(plan.ksegs):
            raise L.HipKernelError(f'{plan.name}: {len(vsrcs)} sources for {len(plan.ksegs)} K segments')
        hd = lstm['hd']
        a = L.Wino44CellArgs()
        for i, (v, sg) in enumerate(zip(vsrcs, plan.ksegs)):
            self._chk(v)
            if v.numel() != int(self.lib.rnh_wino44_v_floats(B, H, W, sg.nch)):
                raise L.HipKernelError(f'{plan.name}: transformed source {i} size')
            a.v[i], a.vchunks[i] = v.data_ptr(), sg.nch // 16
        for k in ('c_prev', 'h_out', 'c_out', 'gates_out'):
            t = lstm.get(k)
            self._chk(t)
            if t is not None and tuple(t.shape) != (B, H, W, hd * (4 if k == 'gates_out' else 1)):
                raise L.HipKernelError(f'{plan.name}: {k} shape {tuple(t.shape)}')
        wp, bp = self._packed44[id(plan)]
        a.nsrc, a.B, a.H, a.W, a.Npad, a.hd = len(vsrcs), B, H, W, 4 * hd, hd
        a.wp, a.bias = wp.data_ptr(), bp.data_ptr()
        a.c_prev, a.h_out, a.c_out, a.gates_out = _ptr(lstm.get('c_prev')), _ptr(lstm['h_out']), _ptr(lstm['c_out']), _ptr(lstm.get('gates_out'))
        return a

    def wino44_cell(self, plan, vsrcs, B, H, W, lstm):
        """One ConvLSTM cell on its transformed sources ``vsrcs`` (tensors of wino44_transform, in the order of the plan's K segments);
        ``lstm`` as for conv()."""
        a = self._wino44_cell_args(plan, vsrcs, B, H, W, lstm)
        L.check(self.lib.rnh_wino44_cell(C.byref(a), self._stream()), f'rnh_wino44_cell({plan.name})')

    def wino44_cell_pair(self, calls, B, H, W):
        """``calls``: two (plan, vsrcs, lstm) of wino44_cell at one geometry - ONE launch (rnh_wino44_cell_pair), the same results bit for bit."""
        (pa, va, la), (pb, vb, lb) = calls
        a, b = self._wino44_cell_args(pa, va, B, H, W, la), self._wino44_cell_args(pb, vb, B, H, W, lb)
        L.check(self.lib.rnh_wino44_cell_pair(C.byref(a), C.byref(b), self._stream()), f'rnh_wino44_cell_pair({pa.name}, {pb.name})')

    def _pack(self, plan: ConvPlan, w, b=None):
        self._chk(w, b)
        if tuple(w.shape) != (plan.Cout, plan.Cin) + ((3, 3) if plan.ntaps == 9 else (1, 1)):
            raise L.HipKernelError(f'{plan.name}: weight shape {tuple(w.shape)} does not match the plan')
        m = self._plan_maps(plan)
        buf = self._packed.get(id(plan))
        if buf is None:
            if plan.bf16:
                buf = (self.empty(plan.nk * plan.Npad * 16, dtype=torch.bfloat16), self.empty(plan.Npad))
            else:
                buf = (self.empty(plan.wns * 16 * plan.Npad * 4 if plan.wino else plan.nk * plan.Npad * 16), self.empty(plan.Npad))
            self._packed[id(plan)] = buf
        wp, bp = buf
        if plan.bf16:
            # (plan.f16w: IEEE-half weights for the f16 MFMA form of the launch - the upsampler's PixelShuffle convolutions in the forward)
            fn, who = (self.lib.rnh_pack_weights_f16, 'rnh_pack_weights_f16') if getattr(plan, 'f16w', False) else \
                (self.lib.rnh_pack_weights_bf16, 'rnh_pack_weights_bf16')
            L.check(fn(_ptr(w), _ptr(b), _ptr(wp), _ptr(bp), _ptr(m['kbase']), _ptr(m['knv']), _ptr(m['ktap']), _ptr(m['kcoff']), _ptr(m['colmap']),
                       plan.nk, plan.Npad, plan.Cout, plan.Cin, plan.ntaps, plan.kstride, int(plan.transposed), self._stream()), f'{who}({plan.name})')
            return
        if plan.wino:
            L.check(self.lib.rnh_wino_pack_weights(_ptr(w), _ptr(b), _ptr(wp), _ptr(bp), _ptr(m['wkbase']), _ptr(m['wknv']),
                                                   _ptr(m['wkcoff']), _ptr(m['colmap']), plan.wns, plan.Npad, plan.Cout, plan.Cin,
                                                   plan.kstride, int(plan.transposed), self._stream()),
                    f'rnh_wino_pack_weights({plan.name})')
            return
        L.check(self.lib.rnh_pack_weights(_ptr(w), _ptr(b), _ptr(wp), _ptr(bp), _ptr(m['kbase']), _ptr(m['knv']),
                                          _ptr(m['ktap']), _ptr(m['kcoff']), _ptr(m['colmap']), plan.nk, plan.Npad,
                                          plan.Cout, plan.Cin, plan.ntaps, plan.kstride, int(plan.transposed),
                                          self._stream()), f'rnh_pack_weights({plan.name})')

    # ---- descriptors --------------------------------------------------------------------------------
    def _fill_src(self, dst, s: Src):
        t = s.t
        self._chk(t, s.add)
        if s.add is not None and s.add.shape != t.shape:
            raise L.HipKernelError('Src.add must have the geometry of Src.t')
        dst.ptr, dst.ptr2 = t.data_ptr(), (s.add.data_ptr() if s.add is not None else None)
        dst.C, dst.c0 = t.shape[-1], s.c0
        dst.nch = t.shape[-1] - s.c0 if s.nch is None else s.nch
        dst.img_off, dst.scale, dst.sub_y, dst.sub_x = s.img_off, s.scale, s.sub[0], s.sub[1]

    # ---- convolution --------------------------------------------------------------------------------
    # ---- two independent calls of equal geometry in one launch ---------------------------------------------------------------------------
    def pair_cells(self, N, H, W):
        """Should the engine hand the ConvLSTM cells of the two directions (same layer, same wavefront slot) - and their data gradients - to conv_pair?
        Yes: at the reference YAML's training shape (16 crops of 32 x 32: 128 workgroups per cell launch on 256 CUs) it is the difference between
        19.6 and 14.8 ms per bf16 step, at BASELINE config 2 (1024 workgroups per cell) still 0.5-0.8 % (half as many launches and event waits;
        profiles/r05_q_*, r05_w_*).  RNH_PAIR=0 switches it off (A/B runs)."""
        return os.environ.get('RNH_PAIR', '1') != '0'

    def conv_pair(self, calls):
        """``calls``: two argument tuples (plan, srcs, B, H, W, kwargs) of conv().  Same results as the two conv() calls, bit for bit; ONE launch
        (rnh_conv_bf16_pair / rnh_conv_wino_pair) where both are bf16 or both Winograd calls of equal geometry, else the two launches."""
        built = [self._conv_args(pl, srcs, B, H, W, **kw) for pl, srcs, B, H, W, kw in calls]
        (k0, a0, n0), (k1, a1, n1) = built
        if k0 == k1 == 'bf16':
            L.check(self.lib.rnh_conv_bf16_pair(C.byref(a0), C.byref(a1), self._stream()), f'rnh_conv_bf16_pair({n0}, {n1})')
        elif k0 == k1 == 'wino':
            L.check(self.lib.rnh_conv_wino_pair(C.byref(a0), C.byref(a1), self._stream()), f'rnh_conv_wino_pair({n0}, {n1})')
        else:
            for kind, a, name in built:
                self._launch_conv(kind, a, name)

    def _launch_conv(self, kind, a, name):
        fn = {'bf16': self.lib.rnh_conv_bf16, 'wino': self.lib.rnh_conv_wino, 'igemm': self.lib.rnh_conv_igemm}[kind]
        L.check(fn(C.byref(a), self._stream()), f'{fn.__name__}({name})')

    def conv(self, plan: ConvPlan, srcs, B, H, W, dsts=None, ps=None, lstm=None, lstm_bwd=None):
        """One convolution launch (rnh_conv_igemm / rnh_conv_wino / rnh_conv_bf16 by the plan): see _conv_args for the arguments."""
        self._launch_conv(*self._conv_args(plan, srcs, B, H, W, dsts, ps, lstm, lstm_bwd))

    def _conv_args(self, plan: ConvPlan, srcs, B, H, W, dsts=None, ps=None, lstm=None, lstm_bwd=None):
        """The argument structure of one convolution launch -> (kind, args, plan name).  ``dsts``: list[Dst] (STORE), ``ps``: (tensor, r) with the tensor
        (B, rH, rW, cq) (PS), ``lstm``: dict(c_prev, h_out, c_out, gates_out, hd) (LSTM).  ``lstm_bwd`` (only where
        lstm_bwd_fusable(plan, ...)): the data gradient of a ConvLSTM cell with the gate backward of the chain's next frame in its
        epilogue - dict(dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, hd, rec_dtype); dsts = [the input gradient] alone, the hd columns
        behind it (the recurrent state gradient) are consumed in the kernel."""
        if id(plan) not in self._packed:
            raise L.HipKernelError(f'{plan.name}: weights were not packed')
        if len(srcs) != len(plan.ksegs):
            raise L.HipKernelError(f'{plan.name}: {len(srcs)} sources for {len(plan.ksegs)} K segments')
        if plan.bf16:
            return 'bf16', self._conv_bf16(plan, srcs, B, H, W, dsts, ps, lstm, lstm_bwd), plan.name
        if lstm_bwd is not None:
            raise L.HipKernelError(f'{plan.name}: the fused gate backward is an epilogue of rnh_conv_bf16')
        a = L.ConvArgs()
        for i, (s, sg) in enumerate(zip(srcs, plan.ksegs)):
            self._fill_src(a.src[i], s)
            if a.src[i].nch != sg.nch:
                raise L.HipKernelError(f'{plan.name}: source {i} has {a.src[i].nch} channels, plan wants {sg.nch}')
            self._check_src_range(a.src[i], s.t, B, H, W, plan.name)
        wp, bp = self._packed[id(plan)]
        a.nsrc, a.B, a.H, a.W, a.ntaps, a.nk = len(srcs), B, H, W, plan.ntaps, plan.nk
        a.wp, a.bias = wp.data_ptr(), (bp.data_ptr() if plan.bkey is not None else None)
        a.Npad, a.epilogue, a.tile = plan.Npad, plan.epilogue, plan.tile | (L.TILE_DIRECT if (self.direct and (plan.epilogue != L.EPI_PS or self.direct_ps)) else 0)
        if plan.epilogue == L.EPI_STORE:
            a.ndst = len(dsts)
            tot = 0
            for i, d in enumerate(dsts):
                self._chk(d.t)
                if tuple(d.t.shape[1:3]) != (H, W) or d.img_off < 0 or d.img_off + B > d.t.shape[0] or \
                        d.c0 + d.ncols > d.t.shape[-1]:
                    raise L.HipKernelError(f'{plan.name}: destination {i} geometry')
                a.dst[i].ptr, a.dst[i].C, a.dst[i].c0 = d.t.data_ptr(), d.t.shape[-1], d.c0
                a.dst[i].ncols, a.dst[i].accumulate, a.dst[i].img_off = d.ncols, int(d.accumulate), d.img_off
                tot += d.ncols
            if tot > plan.Npad:
                raise L.HipKernelError(f'{plan.name}: destination columns exceed Npad')
        elif plan.epilogue == L.EPI_PS:
            t, r = ps
            self._chk(t)
            cq = t.shape[-1]
            if tuple(t.shape) != (B, H * r, W * r, cq):
                raise L.HipKernelError(f'{plan.name}: pixel-shuffle destination shape {tuple(t.shape)}')
            a.ndst = 1
            a.dst[0].ptr, a.dst[0].C, a.dst[0].c0, a.dst[0].ncols = t.data_ptr(), cq, 0, cq
            a.ps_r, a.ps_cq = r, cq
        else:
            hd = lstm['hd']
            for k in ('c_prev', 'h_out', 'c_out', 'gates_out'):
                t = lstm.get(k)
                self._chk(t)
                if t is not None and tuple(t.shape) != (B, H, W, hd * (4 if k == 'gates_out' else 1)):
                    raise L.HipKernelError(f'{plan.name}: {k} shape {tuple(t.shape)}')
            a.hd = hd
            a.c_prev, a.h_out = _ptr(lstm.get('c_prev')), _ptr(lstm['h_out'])
            a.c_out, a.gates_out = _ptr(lstm['c_out']), _ptr(lstm.get('gates_out'))
        if plan.wino:
            if any(s_.scale != srcs[0].scale or s_.add is not None for s_ in srcs):
                raise L.HipKernelError(f'{plan.name}: the Winograd kernel takes sources of one scale, without a second operand')
            a.nk = plan.wns
            a.tile = plan.wino_cols                             # RNH_WINO_COLS64 / RNH_WINO_COLS128
            return 'wino', a, plan.name
        return 'igemm', a, plan.name

    @staticmethod
    def _check_src_range(d, t, B, H, W, who):
        """Host-side shape check before a hand-written kernel reads through the descriptor."""
        if t.dim() != 4 or t.shape[1] != H * d.scale or t.shape[2] != W * d.scale:
            raise L.HipKernelError(f'{who}: source geometry {tuple(t.shape)} does not match output {H}x{W} x{d.scale}')
        if d.img_off < 0 or d.img_off + B > t.shape[0]:
            raise L.HipKernelError(f'{who}: source image range [{d.img_off}, {d.img_off + B}) outside {t.shape[0]}')

    # ---- bf16-storage path (csrc/conv_bf16.hip, wgrad_bf16.hip, mixed_kernels.hip) ----------------------------------
    def _fill_msrc(self, dst, s: Src, B, H, W, who):
        t = s.t
        self._chk(t, mixed=True)
        if s.add is not None:
            raise L.HipKernelError(f'{who}: the bf16 kernels take no second operand')
        dst.ptr, dst.dtype = t.data_ptr(), L.dt_of(t)
        dst.C, dst.c0 = t.shape[-1], s.c0
        dst.nch = t.shape[-1] - s.c0 if s.nch is None else s.nch
        dst.img_off, dst.scale, dst.sub_y, dst.sub_x = s.img_off, s.scale, s.sub[0], s.sub[1]
        self._check_src_range(dst, t, B, H, W, who)

    @staticmethod
    def lstm_bwd_fusable(plan, cx, hd):
        """Can conv(plan, ..., lstm_bwd=...) serve this cell's data gradient?  (bf16 kernel, all cx + hd columns in ONE column tile;
        RNH_FUSE_GATES_BWD=0 keeps the separate rnh_lstm_gates_bwd_m launches for A/B measurements.)"""
        if os.environ.get('RNH_FUSE_GATES_BWD', '1') == '0':
            return False
        tile = 64 if plan.Npad % 128 else 128
        return bool(plan.bf16 and plan.ntaps == 9 and plan.epilogue == L.EPI_STORE and plan.Npad == tile and cx + hd <= tile and
                    cx % 8 == 0 and hd % 8 == 0)

    def _conv_bf16(self, plan, srcs, B, H, W, dsts, ps, lstm, lstm_bwd=None):
        a = L.ConvBf16Args()
        for i, (s, sg) in enumerate(zip(srcs, plan.ksegs)):
            self._fill_msrc(a.src[i], s, B, H, W, plan.name)
            if a.src[i].nch != sg.nch:
                raise L.HipKernelError(f'{plan.name}: source {i} has {a.src[i].nch} channels, plan wants {sg.nch}')
        wp, bp = self._packed[id(plan)]
        a.nsrc, a.B, a.H, a.W, a.ntaps, a.nchunks = len(srcs), B, H, W, plan.ntaps, plan.nchunks
        a.wp, a.bias = wp.data_ptr(), (bp.data_ptr() if plan.bkey is not None else None)
        a.Npad, a.epilogue = plan.Npad, plan.epilogue
        a.wp_f16 = int(bool(getattr(plan, 'f16w', False)))
        if lstm_bwd is not None:
            hd = lstm_bwd['hd']
            if plan.epilogue != L.EPI_STORE or len(dsts) != 1 or not self.lstm_bwd_fusable(plan, dsts[0].ncols, hd):
                raise L.HipKernelError(f'{plan.name}: not a convolution the gate backward can be fused into')
            for k, mixed, width in (('dh', True, 1), ('dc_next', False, 1), ('gates', True, 4), ('c_prev', False, 1), ('c_next', False, 1),
                                    ('dgates', True, 4), ('dc_prev', False, 1)):
                t = lstm_bwd.get(k)
                if t is None and k in ('dh', 'gates', 'c_next', 'dgates'):
                    raise L.HipKernelError(f'{plan.name}: lstm_bwd needs {k}')
                self._chk(t, mixed=mixed)
                if t is not None and tuple(t.shape) != (B, H, W, hd * width):
                    raise L.HipKernelError(f'{plan.name}: lstm_bwd {k} shape {tuple(t.shape)}')
            a.epilogue, a.hd = L.EPI_LSTM_BWD, hd
            a.bw_dh, a.bw_dc_next, a.bw_gates = _ptr(lstm_bwd['dh']), _ptr(lstm_bwd.get('dc_next')), _ptr(lstm_bwd['gates'])
            a.bw_c_prev, a.bw_c_next = _ptr(lstm_bwd.get('c_prev')), _ptr(lstm_bwd['c_next'])
            a.bw_dgates, a.bw_dc_prev = _ptr(lstm_bwd['dgates']), _ptr(lstm_bwd.get('dc_prev'))
            a.bw_dh_dtype, a.gates_dtype, a.bw_dgates_dtype = L.dt_of(lstm_bwd['dh']), L.dt_of(lstm_bwd['gates']), L.dt_of(lstm_bwd['dgates'])
            a.bw_rec_dtype = L.DT_BF16 if lstm_bwd['rec_dtype'] == torch.bfloat16 else L.DT_F32
        if plan.epilogue == L.EPI_STORE:
            a.ndst = len(dsts)
            tot = 0
            for i, d in enumerate(dsts):
                self._chk(d.t, mixed=True)
                if tuple(d.t.shape[1:3]) != (H, W) or d.img_off < 0 or d.img_off + B > d.t.shape[0] or d.c0 + d.ncols > d.t.shape[-1]:
                    raise L.HipKernelError(f'{plan.name}: destination {i} geometry')
                a.dst[i].ptr, a.dst[i].dtype, a.dst[i].C, a.dst[i].c0 = d.t.data_ptr(), L.dt_of(d.t), d.t.shape[-1], d.c0
                a.dst[i].ncols, a.dst[i].accumulate, a.dst[i].img_off = d.ncols, int(d.accumulate), d.img_off
                tot += d.ncols
            if tot > plan.Npad:
                raise L.HipKernelError(f'{plan.name}: destination columns exceed Npad')
        elif plan.epilogue == L.EPI_PS:
            t, r = ps
            self._chk(t, mixed=True)
            cq = t.shape[-1]
            if tuple(t.shape) != (B, H * r, W * r, cq):
                raise L.HipKernelError(f'{plan.name}: pixel-shuffle destination shape {tuple(t.shape)}')
            a.ndst = 1
            a.dst[0].ptr, a.dst[0].dtype, a.dst[0].C, a.dst[0].c0, a.dst[0].ncols = t.data_ptr(), L.dt_of(t), cq, 0, cq
            a.ps_r, a.ps_cq = r, cq
        else:
            hd = lstm['hd']
            for k in ('c_prev', 'h_out', 'c_out', 'gates_out'):
                t = lstm.get(k)
                self._chk(t, mixed=k in ('h_out', 'gates_out'))
                if t is not None and tuple(t.shape) != (B, H, W, hd * (4 if k == 'gates_out' else 1)):
                    raise L.HipKernelError(f'{plan.name}: {k} shape {tuple(t.shape)}')
            a.hd = hd
            a.c_prev, a.h_out = _ptr(lstm.get('c_prev')), _ptr(lstm['h_out'])
            a.c_out, a.gates_out = _ptr(lstm['c_out']), _ptr(lstm.get('gates_out'))
            a.h_dtype = L.dt_of(lstm['h_out'])
            a.gates_dtype = L.dt_of(lstm['gates_out']) if lstm.get('gates_out') is not None else L.DT_F32
        return a

    def _wgrad_bf16(self, plan, xsrcs, ysrcs, B, H, W, dw, db, accumulate):
        m = self._plan_maps(plan)
        a = L.WgradBf16Args()
        for i, (s, sg) in enumerate(zip(xsrcs, plan.xsegs)):
            self._fill_msrc(a.xs[i], s, B, H, W, plan.name)
            if a.xs[i].nch != sg.nch:
                raise L.HipKernelError(f'{plan.name}: x source {i} channels')
        for i, (s, sg) in enumerate(zip(ysrcs, plan.ysegs)):
            self._fill_msrc(a.ys[i], s, B, H, W, plan.name)
            if a.ys[i].nch != sg.nch:
                raise L.HipKernelError(f'{plan.name}: dy source {i} channels')
        rpi = min(H, 32)
        nitems = B * (-(-W // 32)) * (-(-H // rpi))
        nsplit = plan.nsplit_bf16(nitems)
        a.nxs, a.nys, a.xrows_pad, a.ycols_pad = len(xsrcs), len(ysrcs), plan.xrows_pad64, plan.ycols_pad64
        a.B, a.H, a.W, a.ntaps, a.nsplit = B, H, W, plan.ntaps, nsplit
        slab = self._workspace('wgrad_slab', nsplit * plan.ntaps * plan.xrows_pad64 * plan.ycols_pad64)
        bslab = self._workspace('wgrad_bslab', nsplit * plan.ycols_pad64) if db is not None else None
        a.slab, a.bslab = slab.data_ptr(), (bslab.data_ptr() if bslab is not None else None)
        st = self._stream()
        L.check(self.lib.rnh_wgrad_bf16(C.byref(a), st), f'rnh_wgrad_bf16({plan.name})')
        L.check(self.lib.rnh_wgrad_reduce(_ptr(slab), _ptr(bslab), nsplit, plan.ntaps, plan.xrows_pad64, plan.ycols_pad64,
                                          _ptr(m['rowmap64']), _ptr(m['colmap64']), plan.Cin, _ptr(dw), _ptr(db),
                                          int(accumulate), st), f'rnh_wgrad_reduce({plan.name})')

    def cast(self, t, dtype):
        """fp32 <-> bf16 copy (rnh_cast); the same tensor if it already has ``dtype``."""
        if t.dtype == dtype:
            return t
        self._chk(t, mixed=True)
        out = self.empty(*t.shape, dtype=dtype)
        if t.numel() % 8:
            raise L.HipKernelError('cast: element count must be a multiple of 8')
        L.check(self.lib.rnh_cast(_ptr(t), L.dt_of(t), _ptr(out), L.dt_of(out), t.numel(), self._stream()), 'rnh_cast')
        return out

    def _wgrad44(self, plan: WgradPlan, xsrcs, ysrcs, B, H, W, dw, db, accumulate):
        """The weight gradient in Winograd form F(4x4, 3x3) over 4x4 tiles (csrc/wgrad_wino44.hip) - or False where this call is not one it takes:
        fp32, 3x3, whole 4x4 tiles, the x sources in pairs of 64 channels (a pair = one block of 128 weight input channels: h_fwd | h_bwd of a
        window slot, each at its own image offset), every source tensor transformed ONCE over the frames its slots use, one dy source of 128 k channels."""
        if os.environ.get('RNH_WINO44_WGRAD', '0') != '1' or not getattr(plan, 'wino44w', False) or plan.bf16 or (H & 3) or (W & 3):
            return False
        tpi = (H // 4) * (W // 4)                                           # tiles per image
        if len(ysrcs) != 1 or len(xsrcs) % 2 or len(xsrcs) // 2 > 8 or (B * tpi) % 8 or B * H * W >= 2 ** 27:
            return False
        ys = ysrcs[0]
        CO = ys.t.shape[-1] - ys.c0 if ys.nch is None else ys.nch
        if CO % 128 or ys.scale != 1 or ys.add is not None or len(plan.ysegs) != 1 or plan.ysegs[0].co_base or plan.ysegs[0].stride != 1:
            return False
        probs, tensors = [], {}
        for j in range(len(xsrcs) // 2):
            pair = xsrcs[2 * j:2 * j + 2]
            sg = plan.xsegs[2 * j:2 * j + 2]
            if any(s.scale != 1 or s.add is not None or (s.t.shape[-1] - s.c0 if s.nch is None else s.nch) != 64 for s in pair) or \
                    any(g.nch != 64 or g.nvalid != 64 for g in sg) or sg[1].ci_base != sg[0].ci_base + 64:
                return False
            for s in pair:
                key = (s.t.data_ptr(), s.c0)
                lo, hi, t = tensors.get(key, (s.img_off, s.img_off, s.t))
                tensors[key] = (min(lo, s.img_off), max(hi, s.img_off), t)
            probs.append((pair, sg[0].ci_base))
        if any(((hi - lo) * tpi) % 8 or ((hi - lo + B) * tpi) % 8 for lo, hi, _ in tensors.values()):
            return False
        T8 = B * tpi // 8
        ncb, nprob = CO // 128, len(probs)
        want = max(1, 1280 // (nprob * 9 * ncb))                           # ~5 rounds of the chip
        splits = [s_ for s_ in range(1, 65) if T8 % (2 * s_) == 0]             # (at most 64 K splits: the bound memory_plan budgets the partial sums with)
        if not splits:                                                      # (an odd number of k8 rows: the kernel's ring is two deep)
            return False
        S = min(splits, key=lambda s_: abs(s_ - want))
        st = self._stream()
        vt = {}
        for key, (lo, hi, t) in tensors.items():                           # one transform per source tensor over the frames its slots use
            nimg = hi - lo + B
            buf = self._workspace(f'w44_vt{len(vt)}', int(self.lib.rnh_wino44_tmajor_floats(nimg, H, W, 64)))
            L.check(self.lib.rnh_wino44_tmajor(t.data_ptr() + lo * H * W * t.shape[-1] * 4, t.shape[-1], key[1], 64, nimg, H, W, 0, _ptr(buf), st), 'rnh_wino44_tmajor(x)')
            vt[key] = (buf, nimg * tpi // 8, lo)
        zt = self._workspace('w44_zt', int(self.lib.rnh_wino44_tmajor_floats(B, H, W, CO)))
        L.check(self.lib.rnh_wino44_tmajor(ys.t.data_ptr() + ys.img_off * H * W * ys.t.shape[-1] * 4, ys.t.shape[-1], ys.c0, CO, B, H, W, 1, _ptr(zt), st), 'rnh_wino44_tmajor(dy)')
        a = L.Wino44WgradArgs()
        for i, (pair, _) in enumerate(probs):
            for h_, s in enumerate(pair):
                buf, k8, lo = vt[(s.t.data_ptr(), s.c0)]
                a.a[i][h_], a.a_k8[i][h_], a.a_k80[i][h_] = buf.data_ptr(), k8, (s.img_off - lo) * tpi // 8
        part = self._workspace('w44_part', S * nprob * 36 * 128 * CO)
        a.z, a.z_k8, a.part, a.nprob, a.CO, a.T8, a.S = zt.data_ptr(), T8, part.data_ptr(), nprob, CO, T8, S
        L.check(self.lib.rnh_wino44_wgrad_gemm(C.byref(a), st), f'rnh_wino44_wgrad_gemm({plan.name})')
        rb = self._maps.setdefault(('w44_rowbase', id(plan)), self._i32([kb for _, kb in probs]))
        ncol = sum(g.nvalid for g in plan.ysegs)
        zpart = self._workspace('w44_zpart', 64 * CO) if db is not None else None
        L.check(self.lib.rnh_wino44_wgrad_finish(C.byref(a), _ptr(rb), ncol, plan.Cin, _ptr(dw), _ptr(db), _ptr(zpart), 64, int(accumulate), st),
                f'rnh_wino44_wgrad_finish({plan.name})')
        return True

    @staticmethod
    def wino44f_wgrad_on(plan):
        """Is the F(4x4)-tile fused weight gradient (rnh_wino44f_wgrad) wanted for this plan?  RNH_WINO44F_WGRAD: '1' (default) = the plans it was measured
        faster for (hipvsr/plans.py: plan.wino44f), 'all' = wherever the kernel takes the call, '0' = nowhere (A/B runs)."""
        mode = os.environ.get('RNH_WINO44F_WGRAD', '1')
        return mode == 'all' or (mode != '0' and bool(getattr(plan, 'wino44f', False)))

    def wgrad(self, plan: WgradPlan, xsrcs, ysrcs, B, H, W, dw, db=None, accumulate=False, vsrcs=None, vN=None):
        """dw (+)= the weight gradient of the plan's convolution over B images.  ``vsrcs`` (optional, one entry per x source): (V, first, step[, C, c_first]) - the
        (frames, floats) tensor of transformed images (wino44_v / wino44_transform of the source's whole tensor - or of its channels [c_first, c_first + C) -, ``vN`` images per frame), the row of the
        frame that holds the launch's first vN images, and the row step per frame (+1 / -1): where the fused F(4x4)-tile kernel takes the call it then copies
        the x operand from those images instead of transforming the raw tensor again (rnh_wino44f_wgrad_v); otherwise they are ignored."""
        m = self._plan_maps(plan)
        self._chk(dw, db)
        if self._wgrad44(plan, xsrcs, ysrcs, B, H, W, dw, db, accumulate):
            return
        if plan.bf16:
            if len(xsrcs) != len(plan.xsegs) or len(ysrcs) != len(plan.ysegs):
                raise L.HipKernelError(f'{plan.name}: source count')
            return self._wgrad_bf16(plan, xsrcs, ysrcs, B, H, W, dw, db, accumulate)
        if len(xsrcs) != len(plan.xsegs) or len(ysrcs) != len(plan.ysegs):
            raise L.HipKernelError(f'{plan.name}: source count')
        a = L.WgradArgs()
        for i, (s, sg) in enumerate(zip(xsrcs, plan.xsegs)):
            self._fill_src(a.xs[i], s)
            if a.xs[i].nch != sg.nch:
                raise L.HipKernelError(f'{plan.name}: x source {i} channels')
            self._check_src_range(a.xs[i], s.t, B, H, W, plan.name)
        for i, (s, sg) in enumerate(zip(ysrcs, plan.ysegs)):
            self._fill_src(a.ys[i], s)
            if a.ys[i].nch != sg.nch:
                raise L.HipKernelError(f'{plan.name}: dy source {i} channels')
            self._check_src_range(a.ys[i], s.t, B, H, W, plan.name)
        a.B, a.H, a.W, a.ntaps, a.nxs, a.nys = B, H, W, plan.ntaps, len(xsrcs), len(ysrcs)
        whole = all(sg.nvalid == sg.nch for sg in plan.xsegs) and all(sg.nvalid == sg.nch for sg in plan.ysegs)
        if self.wino_wgrad and whole and self.wino44f_wgrad_on(plan) and self.lib.rnh_wino44f_wgrad_supported(C.byref(a)):
            # Winograd form F(3x3, 4x4) over 4x4 tiles, both transforms fused (csrc/wgrad_wino44f.hip): 36 GEMMs over the tiles, fixed-order reduction
            sz = (C.c_int64 * 3)()
            L.check(self.lib.rnh_wino44f_wgrad_ws_floats(C.byref(a), sz), 'rnh_wino44f_wgrad_ws_floats')
            xp = self._workspace('wino_xp', sz[0])
            part = self._workspace('w44f_part', sz[1])
            bpart = self._workspace('w44f_bpart', sz[2])
            if vsrcs is not None and self._wgrad44f_v(plan, a, xsrcs, vsrcs, vN, B, H, W, part, bpart, m, dw, db, accumulate):
                return
            L.check(self.lib.rnh_wino44f_wgrad(C.byref(a), _ptr(xp), _ptr(part), _ptr(bpart), _ptr(m['rowmap']), _ptr(m['colmap']), plan.Cin, _ptr(dw), _ptr(db),
                                               int(accumulate), self._stream()), f'rnh_wino44f_wgrad({plan.name})')
            return
        if self.wino_wgrad and whole and self.lib.rnh_wino_wgrad_supported(C.byref(a)):
            # Winograd form F(3x3, 2x2): zero-padded gathered copy of the inputs, 16 GEMMs over tiles, fixed-order reduction
            sz = (C.c_int64 * 3)()
            L.check(self.lib.rnh_wino_wgrad_ws_floats(C.byref(a), sz), 'rnh_wino_wgrad_ws_floats')
            xp = self._workspace('wino_xp', sz[0])
            slab = self._workspace('wgrad_slab', sz[1])
            bslab = self._workspace('wgrad_bslab', sz[2]) if db is not None else None
            a.slab, a.bslab = slab.data_ptr(), (bslab.data_ptr() if bslab is not None else None)
            L.check(self.lib.rnh_wino_wgrad(C.byref(a), _ptr(xp), _ptr(m['rowmap']), _ptr(m['colmap']), plan.Cin, _ptr(dw), _ptr(db),
                                            int(accumulate), self._stream()), f'rnh_wino_wgrad({plan.name})')
            return
        nsplit = plan.nsplit(B * H * W)
        slab = self._workspace('wgrad_slab', nsplit * plan.ntaps * plan.xcols_pad * plan.ycols_pad)
        bslab = self._workspace('wgrad_bslab', nsplit * plan.ycols_pad) if db is not None else None
        a.nxs, a.xcols_pad, a.nys, a.ycols_pad = len(xsrcs), plan.xcols_pad, len(ysrcs), plan.ycols_pad
        a.xgrp, a.ygrp = m['xgrp'].data_ptr(), m['ygrp'].data_ptr()
        a.B, a.H, a.W, a.ntaps, a.tile, a.nsplit = B, H, W, plan.ntaps, plan.tile, nsplit
        a.slab, a.bslab = slab.data_ptr(), (bslab.data_ptr() if bslab is not None else None)
        a.zero_page = self._zero_page.data_ptr()
        st = self._stream()
        L.check(self.lib.rnh_conv_wgrad(C.byref(a), st), f'rnh_conv_wgrad({plan.name})')
        L.check(self.lib.rnh_wgrad_reduce(_ptr(slab), _ptr(bslab), nsplit, plan.ntaps, plan.xcols_pad, plan.ycols_pad,
                                          _ptr(m['rowmap']), _ptr(m['colmap']), plan.Cin, _ptr(dw), _ptr(db),
                                          int(accumulate), st), f'rnh_wgrad_reduce({plan.name})')

    def _wgrad44f_v(self, plan, a, xsrcs, vsrcs, vN, B, H, W, part, bpart, m, dw, db, accumulate):
        """The fused F(4x4)-tile weight gradient with its x operand copied from transformed images (rnh_wino44f_wgrad_v); False = the call does not qualify
        (the caller launches the form that transforms the raw tensors)."""
        if os.environ.get('RNH_WINO44F_V', '1') == '0' or not vN or len(vsrcs) != len(xsrcs) or B % vN:
            return False
        nfr = B // vN
        vs = (L.Wino44VSrc * len(vsrcs))()
        keep = []
        for i, (vsrc, sx) in enumerate(zip(vsrcs, xsrcs)):
            V, first, step = vsrc[:3]
            Cs, c_first = (vsrc[3], vsrc[4]) if len(vsrc) > 3 else (sx.t.shape[-1], 0)      # the image holds channels [c_first, c_first + Cs) of the tensor
            self._chk(V)
            last = first + step * (nfr - 1)
            if V.dim() != 2 or V.dtype != torch.float32 or Cs % 16 or not (0 <= first < V.shape[0] and 0 <= last < V.shape[0]) or \
                    V.shape[1] != int(self.lib.rnh_wino44_v_floats(vN, H, W, Cs)):
                raise L.HipKernelError(f'{plan.name}: transformed image {i}: a (frames, rnh_wino44_v_floats(vN, H, W, C)) fp32 tensor holding rows {first}..{last}')
            vs[i].v, vs[i].frame_stride, vs[i].nchunks, vs[i].c_first = V.data_ptr() + first * V.shape[1] * 4, step * V.shape[1], Cs // 16, c_first
            keep.append(a.xs[i].img_off)
            a.xs[i].img_off = 0                               # (the frame pointer carries it)
        ok = bool(self.lib.rnh_wino44f_wgrad_v_supported(C.byref(a), vs, vN))
        if ok:
            L.check(self.lib.rnh_wino44f_wgrad_v(C.byref(a), vs, vN, _ptr(part), _ptr(bpart), _ptr(m['rowmap']), _ptr(m['colmap']), plan.Cin, _ptr(dw), _ptr(db),
                                                 int(accumulate), self._stream()), f'rnh_wino44f_wgrad_v({plan.name})')
        for i, o in enumerate(keep):
            a.xs[i].img_off = o
        return ok

    # ---- small kernels ------------------------------------------------------------------------------
    def inconv_fwd(self, x, w, b, slope):
        self._chk(x, w, b, slope)
        B, H, W, Cin = x.shape
        y = self.empty(B, H, W, w.shape[0])
        L.check(self.lib.rnh_inconv_prelu_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(slope), _ptr(y), B, H, W, Cin, w.shape[0],
                                              self._stream()), 'rnh_inconv_prelu_fwd')
        return y

    def inconv_bwd(self, x, w, b, slope, dy, dw, db, dslope, accumulate=False):
        self._chk(x, w, b, slope, dy, dw, db, dslope)
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        if tuple(dy.shape) != (B, H, W, Cout):
            raise L.HipKernelError('inconv_bwd: dy shape')
        ws = self._workspace('inconv', self.lib.rnh_inconv_bwd_ws_floats(Cin, Cout))
        L.check(self.lib.rnh_inconv_prelu_bwd(_ptr(x), _ptr(w), _ptr(b), _ptr(slope), _ptr(dy), _ptr(dw), _ptr(db),
                                              _ptr(dslope), _ptr(ws), B, H, W, Cin, Cout, int(accumulate), self._stream()),
                'rnh_inconv_prelu_bwd')

    def outconv_fwd(self, x, w, b, out=None):
        self._chk(x, w, b, out)
        B, H, W, Cin = x.shape
        y = out if out is not None else self.empty(B, H, W, w.shape[0])
        if tuple(y.shape) != (B, H, W, w.shape[0]):
            raise L.HipKernelError('outconv_fwd: out shape')
        L.check(self.lib.rnh_outconv_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), B, H, W, Cin, w.shape[0], self._stream()),
                'rnh_outconv_fwd')
        return y

    def conv_to_column(self, x, w, col, out, c0, yzero=0):
        """out[..., c0] = the data gradient of a 3x3 convolution with OIHW weights ``w`` (O = x's channels) w.r.t. its input channel ``col``:
        sum_{o, t} x[p - t][o] w[o][col][t]; channels c0+1 .. c0+yzero of ``out`` := 0.  One HBM-bound launch of rnh_outconv_fwd_ld."""
        self._chk(x, w, out)
        B, H, W, Cin = x.shape
        if w.shape[0] != Cin or tuple(w.shape[2:]) != (3, 3) or tuple(out.shape[:3]) != (B, H, W) or c0 + 1 + yzero > out.shape[3]:
            raise L.HipKernelError('conv_to_column: shapes')
        L.check(self.lib.rnh_outconv_fwd_ld(_ptr(x), w.data_ptr() + 4 * 9 * col, 0, w.shape[1] * 9, 1, None, out.data_ptr() + 4 * c0,
                                            out.shape[3], yzero, B, H, W, Cin, 1, self._stream()), 'rnh_outconv_fwd_ld')

    def outconv_dgrad(self, dy, w):
        self._chk(dy, w)
        B, H, W, Cout = dy.shape
        dx = self.empty(B, H, W, w.shape[1])
        L.check(self.lib.rnh_outconv_dgrad(_ptr(dy), _ptr(w), _ptr(dx), B, H, W, w.shape[1], Cout, self._stream()),
                'rnh_outconv_dgrad')
        return dx

    def outconv_wgrad(self, x, dy, dw, db, accumulate=False):
        self._chk(x, dy, dw, db)
        B, H, W, Cin = x.shape
        Cout = dy.shape[-1]
        ws = self._workspace('outconv', self.lib.rnh_outconv_wgrad_ws_floats(Cin, Cout))
        L.check(self.lib.rnh_outconv_wgrad(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), B, H, W, Cin, Cout,
                                           int(accumulate), self._stream()), 'rnh_outconv_wgrad')

    # ---- side path of refine conv1's odd output channel 2*cl (csrc/small_kernels.hip, xcol_*) -----------------
    _XCOL_SEGS = staticmethod(lambda cl: ((0, cl, cl), (cl, cl, cl), (2 * cl, 4, 1)))       # (c0, nch, nvalid) of h_fwd, h_bwd, phase

    def refine_xcol_fwd(self, srcs, w1, b1, R1, N, J, cl):
        """R1[window i][..., 2*cl] = conv1 channel 2*cl over the J frame slots of srcs = (h_fwd, h_bwd, phase plane)."""
        self._chk(w1, b1, R1, *srcs)
        co, cs, Cin = 2 * cl, 2 * cl + 1, w1.shape[1]
        nwin, H, W, C = R1.shape[0] // N, R1.shape[1], R1.shape[2], R1.shape[3]
        zero, zs = self.zeros(J), []
        for s, (c0, nch, nv) in zip(srcs, self._XCOL_SEGS(cl)):
            if s.shape[0] != (nwin + J - 1) * N or s.shape[3] != nch:
                raise L.HipKernelError('refine_xcol_fwd: source shape')
            wx = self.empty(J, nch, 3, 3)
            L.check(self.lib.rnh_xcol_pack(_ptr(w1), _ptr(wx), Cin, co, J, cs, c0, nch, nv, self._stream()), 'rnh_xcol_pack')
            zs.append(self.outconv_fwd(s, wx, zero))
        L.check(self.lib.rnh_xcol_combine(_ptr(zs[0]), _ptr(zs[1]), _ptr(zs[2]), b1.data_ptr() + 4 * co, _ptr(R1), H * W, N, nwin, J, C, co,
                                          self._stream()), 'rnh_xcol_combine')

    def refine_phase_wgrad(self, dy, P4, dw1, N, J, cl, ncols, accumulate):
        """dw1[:ncols, j*(2*cl + 1) + 2*cl] (+)= weight gradient of conv1 w.r.t. its J phase-plane input channels, from border-class sums of
        dy (nwin*N, H, W, C) - the transpose of refine_phase_bias."""
        self._chk(dy, P4, dw1)
        nwin, H, W, C = dy.shape[0] // N, dy.shape[1], dy.shape[2], dy.shape[3]
        if P4.shape[0] != (nwin + J - 1) * N or tuple(P4.shape[1:]) != (H, W, 4):
            raise L.HipKernelError('refine_phase_wgrad: plane shape')
        ws = self._workspace('phase_wgrad', self.lib.rnh_phase_wgrad_ws_floats(H, N, nwin, ncols))
        L.check(self.lib.rnh_phase_wgrad(_ptr(dy), _ptr(P4), _ptr(dw1), _ptr(ws), H, W, N, nwin, J, dw1.shape[1], 2 * cl + 1, 2 * cl, C, ncols,
                                         int(accumulate), self._stream()), 'rnh_phase_wgrad')

    def refine_xcol_dgrad(self, g, w1, dHf, dHb, N, J, cl):
        """dHf / dHb += the data gradient of conv1's output channel 2*cl (g = gradient planes with the window halo): a 45-tap stencil."""
        self._chk(g, w1, dHf, dHb)
        T, H, W, C = dHf.shape[0] // N, dHf.shape[1], dHf.shape[2], g.shape[3]
        if g.shape[0] != (T + J - 1) * N or tuple(dHf.shape) != tuple(dHb.shape) or dHf.shape[3] != cl or tuple(g.shape[1:3]) != (H, W):
            raise L.HipKernelError('refine_xcol_dgrad: shapes')
        ws = self._workspace('xcol_dgrad', J * 9 * 2 * cl)
        L.check(self.lib.rnh_xcol_dgrad(_ptr(g), _ptr(w1), _ptr(dHf), _ptr(dHb), _ptr(ws), H, W, N, T, J, w1.shape[1], 2 * cl + 1, 2 * cl, C, cl,
                                        self._stream()), 'rnh_xcol_dgrad')

    def xcol_combine_m(self, z, b1, R1, N, J, c0):
        """bf16-storage path: R1[window i][..., c0] = b1[c0] + sum_j z[frame i + j][..., j] (channels c0 + 1 .. c0 + 7 := 0); z holds
        the J slot convolutions of conv1's channel c0 per source frame (one small rnh_conv_bf16 over the frames)."""
        self._chk(z, b1)
        self._chk(R1, mixed=True)
        nwin, H, W, C = R1.shape[0] // N, R1.shape[1], R1.shape[2], R1.shape[3]
        if tuple(z.shape) != ((nwin + J - 1) * N, H, W, 8):
            raise L.HipKernelError('xcol_combine_m: shapes')
        L.check(self.lib.rnh_xcol_combine_m(_ptr(z), _ptr(b1), _ptr(R1), L.dt_of(R1), H * W, N, nwin, J, C, c0, self._stream()), 'rnh_xcol_combine_m')

    def xcol_gather_m(self, dy, N, J, c, dtype):
        """E ((nwin + J - 1)*N, H, W, 8)[frame f][..., j] = dy[window f - j][..., c]: the gradient operand of the per-frame convolution."""
        self._chk(dy, mixed=True)
        nwin, H, W, C = dy.shape[0] // N, dy.shape[1], dy.shape[2], dy.shape[3]
        E = self.empty((nwin + J - 1) * N, H, W, 8, dtype=dtype)
        L.check(self.lib.rnh_xcol_gather_m(_ptr(dy), L.dt_of(dy), _ptr(E), L.dt_of(E), H * W, N, nwin, J, C, c, self._stream()), 'rnh_xcol_gather_m')
        return E

    @staticmethod
    def put_scalar(dst, src, accumulate):
        """dst (a 1-element view of a gradient) = or += src (a 1-element view): plumbing, one tiny ATen launch."""
        if accumulate:
            dst.add_(src)
        else:
            dst.copy_(src)

    def refine_phase_bias(self, R1, P4, w1, N, J, cl, ncols):
        """R1[..., :ncols] += conv1 over the J phase planes (input channel 2*cl of every frame slot), as a bias field."""
        self._chk(R1, P4, w1)
        nwin, H, W, C = R1.shape[0] // N, R1.shape[1], R1.shape[2], R1.shape[3]
        if P4.shape[0] != (nwin + J - 1) * N or tuple(P4.shape[1:]) != (H, W, 4):
            raise L.HipKernelError('refine_phase_bias: plane shape')
        ws = self._workspace('phase_bias', 16 * J * ncols)
        L.check(self.lib.rnh_phase_bias_add(_ptr(R1), _ptr(P4), _ptr(w1), _ptr(ws), H, W, N, nwin, J, w1.shape[1], 2 * cl + 1, 2 * cl, C,
                                            ncols, self._stream()), 'rnh_phase_bias_add')

    def refine_xcol_wgrad(self, srcs, dy, dw1, db1, N, J, cl, accumulate):
        """Weight / bias gradient of conv1's channel 2*cl: srcs = the (nwin + J - 1)*N source frames of (h_fwd, h_bwd,
        phase plane), dy = (nwin*N, H, W, C1p) gradient of conv1's output."""
        self._chk(dy, dw1, db1, *srcs)
        co, cs, Cin = 2 * cl, 2 * cl + 1, dw1.shape[1]
        nwin, H, W, C = dy.shape[0] // N, dy.shape[1], dy.shape[2], dy.shape[3]
        E = self.empty((nwin + J - 1) * N, H, W, J)
        L.check(self.lib.rnh_xcol_gather(_ptr(dy), _ptr(E), H * W, N, nwin, J, C, co, self._stream()), 'rnh_xcol_gather')
        for k, (s, (c0, nch, nv)) in enumerate(zip(srcs, self._XCOL_SEGS(cl))):
            if s.shape[0] != (nwin + J - 1) * N or s.shape[3] != nch:
                raise L.HipKernelError('refine_xcol_wgrad: source shape')
            dwx, dbx = self.empty(J, nch, 3, 3), self.empty(J)
            self.outconv_wgrad(s, E, dwx, dbx)
            L.check(self.lib.rnh_xcol_unpack(_ptr(dwx), _ptr(dbx) if k == 0 else None, _ptr(dw1), _ptr(db1), Cin, co, J, cs, c0, nch, nv,
                                             int(accumulate), self._stream()), 'rnh_xcol_unpack')

    # ---- collapsed backward of the upsampler tail (csrc/uptail.hip) -----------------------------------------
    def uptail_fwd(self, y1, w2, b2, w3, b3, r, out):
        self._chk(w2, b2, w3, b3, out)
        self._chk(y1, mixed=True)
        B, Hm, Wm, C1 = y1.shape
        Co, Cq = w3.shape[0], w3.shape[1]
        if tuple(out.shape) != (B, Hm * r, Wm * r, Co) or tuple(w2.shape) != (Cq * r * r, C1, 3, 3):
            raise L.HipKernelError('uptail_fwd: shapes')
        if y1.dtype == torch.bfloat16:                         # bf16-storage path: the tail on bf16 MFMA (csrc/uptail_bf16.hip)
            ws = self._workspace('uptail_fwd_bf16', self.lib.rnh_uptail_fwd_bf16_ws_floats(C1, Cq, r, Co))
            L.check(self.lib.rnh_uptail_fwd_bf16(_ptr(y1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), _ptr(out), _ptr(ws), B, Hm, Wm, C1, Cq,
                                                 r, Co, self._stream()), 'rnh_uptail_fwd_bf16')
            return out
        ws = self._workspace('uptail_fwd', self.lib.rnh_uptail_fwd_ws_floats(C1, Cq, r, Co))
        L.check(self.lib.rnh_uptail_fwd(_ptr(y1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), _ptr(out), _ptr(ws), B, Hm, Wm, C1, Cq, r, Co,
                                        self._stream()), 'rnh_uptail_fwd')
        return out

    def uptail_bf16_supported(self, C1, r, Co):
        """The tail's input / input gradient may live in bf16 (rnh_uptail_*_bf16): r == 2, C1 == 64, out_channels == 1."""
        return bool(self.lib.rnh_uptail_bf16_supported(C1, r, Co))

    @staticmethod
    def uptail_fwd_supported(r, Co):
        return r in (2, 3) and Co == 1

    def uptail_compose(self, w2, w3, r):
        self._chk(w2, w3)
        Co, Cq = w3.shape[0], w3.shape[1]
        C1 = w2.shape[1]
        G = self.empty(self.lib.rnh_uptail_g_floats(C1, r, Co))
        L.check(self.lib.rnh_uptail_compose(_ptr(w2), _ptr(w3), _ptr(G), C1, Cq, r, Co, self._stream()), 'rnh_uptail_compose')
        return G

    def uptail_dgrad(self, d_o, G, C1, r, dtype=torch.float32):
        self._chk(d_o, G)
        B, Hh, Wh, Co = d_o.shape
        if Hh % r or Wh % r:
            raise L.HipKernelError('uptail_dgrad: output size not a multiple of r')
        dy1 = self.empty(B, Hh // r, Wh // r, C1, dtype=dtype)
        if dtype == torch.bfloat16:
            ws = self._workspace('uptail_dgrad_bf16', self.lib.rnh_uptail_dgrad_bf16_ws_floats())
            L.check(self.lib.rnh_uptail_dgrad_bf16(_ptr(d_o), _ptr(G), _ptr(dy1), _ptr(ws), B, Hh // r, Wh // r, C1, Co, r, self._stream()),
                    'rnh_uptail_dgrad_bf16')
            return dy1
        L.check(self.lib.rnh_uptail_dgrad(_ptr(d_o), _ptr(G), _ptr(dy1), B, Hh // r, Wh // r, C1, Co, r, self._stream()),
                'rnh_uptail_dgrad')
        return dy1

    def uptail_xcorr_supported(self, C1, r, Co):
        return bool(self.lib.rnh_uptail_xcorr_supported(C1, r, Co))

    def uptail_xcorr(self, y1, d_o, r, out=None):
        """M (ND*ND, C1, 3, 3) and S (ND*ND) of the collapsed tail straight from the conv input and d_o (Co == 1); out = (M, S) allocated
        by the caller (a caller that launches this on the helper stream allocates on its own stream)."""
        self._chk(d_o)
        self._chk(y1, mixed=True)
        B, Hm, Wm, C1 = y1.shape
        if tuple(d_o.shape) != (B, Hm * r, Wm * r, 1):
            raise L.HipKernelError('uptail_xcorr: shapes')
        nd2 = (r + 2) * (r + 2)
        M, S = out if out is not None else (self.empty(nd2, C1, 3, 3), self.empty(nd2))
        if tuple(M.shape) != (nd2, C1, 3, 3) or tuple(S.shape) != (nd2,):
            raise L.HipKernelError('uptail_xcorr: out shapes')
        self._chk(M, S)
        ws = self._workspace('uptail_xcorr', self.lib.rnh_uptail_xcorr_ws_floats(B, Hm, Wm, C1, r))
        fn, name = (self.lib.rnh_uptail_xcorr_bf16, 'rnh_uptail_xcorr_bf16') if y1.dtype == torch.bfloat16 else \
            (self.lib.rnh_uptail_xcorr, 'rnh_uptail_xcorr')
        L.check(fn(_ptr(y1), _ptr(d_o), _ptr(M), _ptr(S), _ptr(ws), B, Hm, Wm, C1, r, self._stream()), name)
        return M, S

    def uptail_expand(self, d_o, r, Dc=None):
        self._chk(d_o)
        B, Hh, Wh, Co = d_o.shape
        Dc = (Co * (r + 2) * (r + 2) + 3) // 4 * 4 if Dc is None else Dc
        D = self.empty(B, Hh // r, Wh // r, Dc)
        L.check(self.lib.rnh_uptail_expand(_ptr(d_o), _ptr(D), B, Hh // r, Wh // r, Co, r, Dc, self._stream()), 'rnh_uptail_expand')
        return D

    def uptail_wcontract(self, M, S, w2, b2, w3, dw2, db2, dw3, db3, r, acc2, acc3):
        self._chk(M, S, w2, b2, w3, dw2, db2, dw3, db3)
        Co, Cq = w3.shape[0], w3.shape[1]
        L.check(self.lib.rnh_uptail_wcontract(_ptr(M), _ptr(S), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(dw2), _ptr(db2), _ptr(dw3),
                                              _ptr(db3), w2.shape[1], Cq, r, Co, int(acc2), int(acc3), self._stream()),
                'rnh_uptail_wcontract')

    def lstm_gates_bwd(self, dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, dh2=None):
        # the 8-elements-per-thread kernel of mixed_kernels.hip also serves all-fp32 operands (same arithmetic, same
        # results bit for bit; 5 TB/s against the 1.8 TB/s of the 4-element kernel): taken whenever hd % 8 == 0
        if dh.shape[-1] % 8 == 0 or any(t is not None and t.dtype != torch.float32 for t in (dh, dh2, gates, dgates)):
            self._chk(dh, dh2, gates, dgates, mixed=True)
            self._chk(dc_next, c_prev, c_next, dc_prev)
            hd = dh.shape[-1]
            if gates.numel() != 4 * dh.numel() or dgates.numel() != 4 * dh.numel() or any(
                    t is not None and t.shape != dh.shape for t in (dc_next, c_prev, c_next, dc_prev, dh2)):
                raise L.HipKernelError('lstm_gates_bwd: shapes')
            L.check(self.lib.rnh_lstm_gates_bwd_m(_ptr(dh), L.dt_of(dh), _ptr(dh2), L.dt_of(dh2) if dh2 is not None else L.DT_F32,
                                                  _ptr(dc_next), _ptr(gates), L.dt_of(gates), _ptr(c_prev), _ptr(c_next), _ptr(dgates),
                                                  L.dt_of(dgates), _ptr(dc_prev), dh.numel() // hd, hd, self._stream()), 'rnh_lstm_gates_bwd_m')
            return
        self._chk(dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, dh2)
        hd = dh.shape[-1]
        npix = dh.numel() // hd
        for t in (dc_next, c_prev, c_next, dc_prev, dh2):
            if t is not None and t.shape != dh.shape:
                raise L.HipKernelError('lstm_gates_bwd: state shapes')
        if gates.numel() != 4 * dh.numel() or dgates.numel() != 4 * dh.numel():
            raise L.HipKernelError('lstm_gates_bwd: gate shapes')
        L.check(self.lib.rnh_lstm_gates_bwd(_ptr(dh), _ptr(dh2), _ptr(dc_next), _ptr(gates), _ptr(c_prev), _ptr(c_next), _ptr(dgates),
                                            _ptr(dc_prev), npix, hd, self._stream()), 'rnh_lstm_gates_bwd')

    def wino44_gates_bwd_supported(self, H, W, hd):
        """Does rnh_wino44_gates_bwd (the gate backward + the transform of its gate gradients in one launch) serve cells of this shape?"""
        return bool(self.lib.rnh_wino44_gates_bwd_supported(H, W, hd))

    def wino44_gates_bwd(self, dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, dh2, v):
        """lstm_gates_bwd(...) and wino44_transform(Src(dgates), ..., v) in ONE launch (fp32; rnh_wino44_gates_bwd)."""
        self._chk(dh, dc_next, gates, c_prev, c_next, dgates, dc_prev, dh2, v)
        B, H, W, hd = dh.shape
        for t in (dc_next, c_prev, c_next, dc_prev, dh2):
            if t is not None and t.shape != dh.shape:
                raise L.HipKernelError('wino44_gates_bwd: state shapes')
        if tuple(gates.shape) != (B, H, W, 4 * hd) or tuple(dgates.shape) != (B, H, W, 4 * hd):
            raise L.HipKernelError('wino44_gates_bwd: gate shapes')
        if v.numel() != int(self.lib.rnh_wino44_v_floats(B, H, W, 4 * hd)):
            raise L.HipKernelError('wino44_gates_bwd: size of the transformed image')
        L.check(self.lib.rnh_wino44_gates_bwd(_ptr(dh), _ptr(dh2), _ptr(dc_next), _ptr(gates), _ptr(c_prev), _ptr(c_next), _ptr(dgates), _ptr(dc_prev),
                                              _ptr(v), B, H, W, hd, self._stream()), 'rnh_wino44_gates_bwd')

    def add(self, out, a, b=None, c=None, accumulate=False):
        n = out.numel()
        for t in (a, b, c):
            if t is not None and t.numel() != n:
                raise L.HipKernelError('add: size mismatch')
        if any(t is not None and t.dtype != torch.float32 for t in (out, a, b, c)):
            self._chk(out, a, b, c, mixed=True)
            dt = lambda t: L.dt_of(t) if t is not None else L.DT_F32          # noqa: E731
            L.check(self.lib.rnh_ew_add_m(_ptr(out), dt(out), _ptr(a), dt(a), _ptr(b), dt(b), _ptr(c), dt(c), n, int(accumulate),
                                          self._stream()), 'rnh_ew_add_m')
            return out
        self._chk(out, a, b, c)
        L.check(self.lib.rnh_ew_add(_ptr(out), _ptr(a), _ptr(b), _ptr(c), n, int(accumulate), self._stream()), 'rnh_ew_add')
        return out

    def phase_plane(self, pos, N, F, H, W, dtype=torch.float32, channels=4):
        pos = pos.reshape(N, F).contiguous()
        self._chk(pos)
        if channels == 8:                                     # the bf16-storage path's 8-channel plane (p, 0, ..., 0)
            out = self.empty(F * N, H, W, 8, dtype=dtype)
            L.check(self.lib.rnh_phase_plane_m(_ptr(pos), _ptr(out), L.dt_of(out), N, F, H, W, self._stream()), 'rnh_phase_plane_m')
            return out
        out = self.empty(F * N, H, W, 4)
        L.check(self.lib.rnh_phase_plane(_ptr(pos), _ptr(out), N, F, H, W, self._stream()), 'rnh_phase_plane')
        return out

    def loss_total(self, x, w, G, T, backward=False):
        """sum_g w[g] * mean_i x[g*T + i] (one element), or with backward its gradient [G*T] from the upstream scalar x."""
        self._chk(x, w)
        if w.numel() != G or x.numel() != (1 if backward else G * T):
            raise L.HipKernelError('loss_total: shapes')
        out = self.empty(G * T if backward else 1)
        L.check(self.lib.rnh_loss_total(_ptr(x), _ptr(w), _ptr(out), G, T, 1 if backward else 0, self._stream()), 'rnh_loss_total')
        return out

    def loss(self, o, y, G, T, kind, eps, gscale=None, want_grad=False):
        """o: (G*T, ...) outputs, y: (T, ...) targets.  Returns (loss[G*T], dO or None)."""
        self._chk(o, y, gscale)
        per = y.numel() // T
        if o.numel() != G * T * per:
            raise L.HipKernelError('loss: shapes')
        loss = self.empty(G * T)
        d_o = torch.empty_like(o) if want_grad else None
        ws = self._workspace('loss', G * T * L.LOSS_BLOCKS)
        L.check(self.lib.rnh_loss_fwd_bwd(_ptr(o), _ptr(y), _ptr(loss), _ptr(d_o), _ptr(gscale), _ptr(ws), G, T, per, kind,
                                          float(eps), self._stream()), 'rnh_loss_fwd_bwd')
        return loss, d_o
