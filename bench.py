#!/usr/bin/env python3
"""Headline benchmark: cine-frames/sec of the RefineNet x4 training step (forward + deep-supervision L1 loss +
backward + gradient all-reduce + Adam step) on synthetic Gaussian cine stacks, BASELINE.json config 2:
N = 8 samples per GPU, T = 7 supervised frames (F = 19 input frames), 128x128 -> 512x512, fp32.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (one rank per GPU, weak scaling: 8 samples per rank)
  python bench.py --gpus N ...                            (no WORLD_SIZE in the environment: starts exactly that torchrun command
                                                           itself, before anything touches the GPU, relays rank 0's line and exits
                                                           with the children's code; WORLD_SIZE != --gpus is an error)
  python bench.py --config {2,4,5}                        (BASELINE.json configs[1] (default), [3]: x2, N=16, T=5, 256x256 and [4]: x4 with
                                                           phase codes, N=8 per GPU, T=11, 96x96; the bf16 step of the same shape as `secondary`)

Rank 0 prints ONE JSON line.  `value` = supervised frames (N_global * T) per second over the timed steps (max over
ranks); `ms_per_step_median` is the median of the per-step HIP-event times of the same steps.  `roofline` prices the
ConvLSTM cell forward (318 launches per step) from HIP-event timing of that launch on this run, in the form the engine runs it at
the benchmark's shape: Winograd F(4x4,3x3) on transformed inputs (rnh_wino44_cell: wherever the images are whole
4x4 tiles - every BASELINE config) or F(2x2,3x3) (rnh_conv_wino).  `achieved` / `frac` = the FLOPs the matrix cores EXECUTE (36 GEMMs over the 4x4 tiles =
1/4 of the direct form; 16 GEMMs over the 2x2 tiles = 4/9) over the launch time and the fp32 MFMA peak - a fraction of a ceiling,
<= 1; the same launch priced in the reference's direct 3x3 formulation (589 824 FLOP per pixel) is reported
beside it as `algorithmic_equiv_tflops` / `algorithmic_equiv_frac` (may exceed 1: Winograd does not do that work).  The F(4x4) form
needs one launch of the input transform per cell (rnh_wino44_transform, HBM-bound): `input_transform` times it against 8 TB/s and
`cell_plus_transform_ms` / `algorithmic_equiv_tflops_with_transform` price the pair.  `secondary` carries the SAME step in the bf16-storage path (BASELINE config 3 per GPU: same N, T,
size, steps and warm-up, timed the same way in the same process right after the headline) with its own `ms_per_step`, `value`,
`roofline` (conv_bf16d_kernel<LSTM> against the dense bf16 MFMA peak; executed = algorithmic there) and `dtype: "bf16"`;
`--no-secondary` skips it, `--dtype bf16` makes it the only line as before.  `traffic` = HBM bytes per launch from rocprofv3 PMC passes
(profiles/lstm44_kernel_hbm_bytes.json or lstm_kernel_hbm_bytes.json; profiles/lstm_bf16_kernel_hbm_bytes.json for the secondary), reported only if that
record was measured on THIS kernel source (sha256 of csrc/conv_wino44.hip / conv_wino.hip / conv_bf16.hip), else null.  `cpu_baseline` times
the CPU oracle (= the reference's computation, bit-exact) on this host's cores at BASELINE config 1 (rank 0, 1 GPU runs
only).  `config` carries the step's FLOPs in the reference's formulation and as executed here.
Diagnosis (environment, no effect on the numbers' definition): BENCH_EACH_STEP=1 prints every timed step's HIP-event time and host enqueue time to stderr;
BENCH_DUMMY_STREAMS=<n> / BENCH_DUMMY_ALLOC_GB=<n> give a case the stream-pool position / allocator history an earlier case of the same process would have left
(how the 5 % penalty of the second engine of a process was traced to its side streams, DESIGN.md 4d c).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import subprocess                 # noqa: E402

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at 256 CUs x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA (the 5 PF headline figure includes 2:1 sparsity)
PEAK_HBM_TBS = 8.0


def step_flops_per_lr_pixel(T, U=6, S=3, L=3, scale=4, executed=False, cell44=False, refine44=False, up44=False, refine_dgrad44=False, cell_dgrad44=False,
                            cell_wgrad44=False, refine1_wgrad44=False, refine2_wgrad44=False, up_wgrad44=False):
    """Conv FLOPs (2*MAC) per LR pixel per sample.  executed=False: the reference's layer-by-layer formulation
     - what its PyTorch step computes and what `step_tflop` reports.  executed=True (x4 only):
    what this implementation issues - the last PixelShuffle conv + final conv (1 198 080 FLOP/LR pixel forward, twice
    that backward) are replaced by the composed 5x5 forward (51 200) and the merged-offset backward (2 x 32 768)."""
    F = T + 2 * U
    out_s = {4: 1492992, 2: 299520, 3: 673920}[scale]
    out_f, out_b = out_s, 2 * out_s
    w = 4.0 / 9.0 if executed else 1.0        # convolutions that run in Winograd form execute 4/9 of their direct FLOPs
    if executed and scale == 4:
        # first PixelShuffle conv: fwd, dgrad, wgrad Winograd (up44: the forward in F(4x4, 3x3) form); tail collapsed
        # (up_wgrad44, round 6: its weight gradient in the fused F(4x4)-tile form)
        out_f, out_b = (0.25 if up44 else w) * 294912 + 51200, (w + (0.25 if up_wgrad44 else w)) * 294912 + 65536
    elif executed and scale == 2:
        out_f, out_b = 12800, 16384                                           # the whole upsampler IS the collapsed tail (one PixelShuffle stage)
    # fwd, dgrad, wgrad in Winograd F(2x2, 3x3) form; cell44: the forward cell in F(4x4, 3x3) form (rnh_wino44_cell: 36 products per 16 outputs,
    # 1/4 of the direct FLOPs) where the engine selects it (HipOps.wino44_ok)
    # cell_dgrad44 / cell_wgrad44 (round 6): the cell's data gradient (rnh_wino44_conv on the transformed gate gradients) and its weight gradient
    # (rnh_wino44f_wgrad: F(4x4)-tile form, both transforms fused) at 1/4 of the direct FLOPs as well
    lstm_f = (0.25 if executed and cell44 else w) * 589824
    lstm_b = ((0.25 if executed and cell_dgrad44 else w) + (0.25 if executed and cell_wgrad44 else w)) * 589824
    r1, r2 = 645 * 129 * 18, 129 * 64 * 18                                    # refine conv1 / conv2, 2*MAC per pixel
    r2h, r2x = 128 * 64 * 18, 1 * 64 * 18                                     # conv2: the 128 hidden-state channels / channel 128
    # conv1 in Winograd form (fwd, dgrad, wgrad); conv2 forward and data gradient in Winograd form over its 128 hidden-state
    # channels (channel 128 through the implicit GEMM), its weight gradient as pixel-contraction GEMM (direct)
    # refine44: conv1's forward in F(4x4, 3x3) form (rnh_wino44_conv on the transformed top-layer h')
    # refine_dgrad44: its data gradient likewise (the weight gradient stays in F(2x2)-tile form)
    # refine1_wgrad44 / refine2_wgrad44 (round 6): their weight gradients over the hidden-state rows in the fused F(4x4)-tile form
    ref_f = (0.25 if executed and refine44 else w) * r1 + w * r2h + r2x
    ref_b = ((0.25 if executed and refine_dgrad44 else w) + (0.25 if executed and refine1_wgrad44 else w)) * r1 + (w * r2h + r2x) + \
        ((0.25 * r2h + r2x) if executed and refine2_wgrad44 else r2)
    nfr, nwin = S * F, S * (F - 4)            # ConvLSTM frames per direction and refine windows, all stages
    if executed:                              # the last stage stops at the last refine window / computes the T supervised windows only
        nfr, nwin = (S - 1) * F + (U + T + 2), (S - 1) * (F - 4) + T
    fwd = F * 1152 + 2 * nfr * L * lstm_f + nwin * ref_f + 3 * S * T * out_f
    bwd = T * 1152 + S * 2 * T * L * lstm_b + S * T * ref_b + 3 * S * T * out_b
    return fwd + bwd


def step_flops_bf16(T, U=6, S=3, L=3, scale=4):
    """Executed conv FLOPs per LR pixel per sample of the bf16-storage path (x4): every convolution in direct form, refine
    conv1 on 192 padded columns and its phase planes as 8-channel sources are NOT counted (only the reference's 129 x 645),
    the upsampler tail collapsed as in the fp32 path, the dead last-stage work skipped."""
    F = T + 2 * U
    lstm = 589824
    r1, r2 = 645 * 129 * 18, 129 * 64 * 18
    out_f, out_b = (294912 + 51200, 2 * 294912 + 65536) if scale == 4 else (12800, 16384)
    nfr, nwin = (S - 1) * F + (U + T + 2), (S - 1) * (F - 4) + T
    fwd = F * 1152 + 2 * nfr * L * lstm + nwin * (r1 + r2) + 3 * S * T * out_f
    bwd = T * 1152 + S * 2 * T * L * 2 * lstm + S * T * 2 * (r1 + r2) + 3 * S * T * out_b
    return fwd + bwd


def make_net(dev, seed=0, scale=4):
    from src.model.nets import RefineNet
    torch.manual_seed(seed)
    net = RefineNet(in_channels=1, out_channels=1, num_features=[64, 64, 64], upscale_factor=scale, num_stages=3,
                    update_memory=True, num_updated_frames=6, refine_window_size=5, positional_encoding=True)
    return net.to(dev).train()


def synthetic_batch(dev, n, t, h, w, seed, u=6, s=4):
    g = torch.Generator(device=dev).manual_seed(seed)
    F = t + 2 * u
    # frame lists as views of one packed (F, N, 1, h, w) / (T, N, 1, sh, sw) buffer each - what the product's loader
    # (hipvsr.cine_cache.GpuCineLoader: one gather launch per batch) hands to the trainer, so the engine and the fused
    # loss take them without a copy; same values as F + T separate randn calls with this generator
    inputs = list(torch.stack([torch.randn(n, 1, h, w, generator=g, device=dev) for _ in range(F)]).unbind(0))
    targets = list(torch.stack([torch.randn(n, 1, s * h, s * w, generator=g, device=dev) for _ in range(t)]).unbind(0))
    phi = torch.randint(0, 30, (n, 1), generator=g, device=dev).float()
    k = torch.arange(F, device=dev).float().unsqueeze(0)
    pos = torch.cos(2 * torch.pi * (k + phi) / 30.0).unsqueeze(-1)
    return inputs, targets, pos


def lstm_kernel_roofline(net, dev, n, h, w, reps=200, warm=50):
    """Average duration of ONE ConvLSTM-cell launch (rnh_conv_wino with the LSTM epilogue in the fp32 path, rnh_conv_bf16 in the
    bf16-storage path) at the benchmark shape, by HIP events on the stream the kernels run on (torch's current stream)."""
    from hipvsr.plans import Src
    eng = net._engine()
    ops, pl = eng.ops, eng.plans.lstm[('forward', 1)]
    hd, cx = pl['hd'], pl['cx']
    params = {k: p.detach() for k, p in net.named_parameters()}
    ops.pack(pl['full'], params[pl['full'].wkey], params[pl['full'].bkey])
    act = eng.act
    x, hp, cp = torch.randn(n, h, w, cx, device=dev).to(act), torch.randn(n, h, w, hd, device=dev).to(act), torch.randn(n, h, w, hd, device=dev)
    ho, co = ops.empty(n, h, w, hd, dtype=act), ops.empty(n, h, w, hd)
    go = ops.empty(n, h, w, 4 * hd, dtype=act)

    lstm = dict(hd=hd, c_prev=cp, h_out=ho, c_out=co, gates_out=go)
    # the form the engine runs this launch in: Winograd F(4x4, 3x3) on transformed inputs (rnh_wino44_cell) or F(2x2, 3x3) (rnh_conv_wino)
    cell44 = (not eng.bf16) and ops.wino44_ok(pl['full'], n, h, w)
    if cell44:
        vx, vh = ops.wino44_v(n, h, w, cx)[0], ops.wino44_v(n, h, w, hd)[0]
        ops.wino44_transform(Src(x), n, h, w, vx)
        ops.wino44_transform(Src(hp), n, h, w, vh)

    def launch():
        if cell44:
            ops.wino44_cell(pl['full'], [vx, vh], n, h, w, lstm)
        else:
            ops.conv(pl['full'], [Src(x), Src(hp)], n, h, w, lstm=lstm)
    for _ in range(warm):                                  # (50 launches = 16 ms: the clocks have settled; with 3 the 20 timed launches that followed came
        launch()                                           #  out anywhere between 0.325 and 0.346 ms on the same tree - the profiler's 403-launch average is 0.325)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    wino = bool(getattr(pl['full'], 'wino', False))
    if eng.bf16:
        # bf16-storage path: the direct 3x3 form on bf16 MFMA (rnh_conv_bf16); executed == algorithmic FLOPs.  The launch is
        # priced against both ceilings: the dense bf16 MFMA peak and HBM (bf16 x, h in; fp32 c in / out; bf16 h', gates out)
        flops = 2.0 * n * h * w * (4 * hd) * (9 * (cx + hd))
        byts = n * h * w * (2 * cx + 2 * hd + 4 * hd + 4 * hd + 2 * hd + 2 * 4 * hd)
        tf, tbs = flops / (ms * 1e-3) / 1e12, byts / (ms * 1e-3) / 1e12
        traffic, traffic_src = quoted_traffic('lstm_bf16_kernel_hbm_bytes.json', 'conv_bf16.hip', (n, h, w))
        return {'bound': 'mfma', 'kernel': 'conv_bf16d_kernel<LSTM,128,9,KC 32> (ConvLSTM cell 128->256, direct 3x3 on v_mfma_f32_32x32x16_bf16, fused gates)',
                'achieved': round(tf, 2), 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / PEAK_BF16_MFMA_TFLOPS, 4),
                'traffic': traffic, 'traffic_source': traffic_src, 'avg_launch_ms': round(ms, 4), 'executed_mfma_flop_per_launch': flops,
                'algorithmic_flop_per_launch': flops,
                'algorithmic_bytes_per_launch': byts, 'hbm_algorithmic_tbs': round(tbs, 3), 'hbm_frac_of_8tbs': round(tbs / PEAK_HBM_TBS, 4)}
    # ALGORITHMIC work of the launch: 589 824 FLOP per pixel at cx = hd = 64, the direct 3x3 form
    flops = 2.0 * n * h * w * (4 * hd) * (9 * (cx + hd))
    if cell44:
        # the F(4x4, 3x3) cell: 36 GEMMs over n*h*w/16 tiles = 1/4 of the direct FLOPs on the matrix cores.  Its inputs arrive in transform-domain
        # form, written by rnh_wino44_transform - one launch per cell (the cell's own h', read by both cells that consume it), timed here too
        for _ in range(warm):
            ops.wino44_transform(Src(ho), n, h, w, vh)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            ops.wino44_transform(Src(ho), n, h, w, vh)
        e1.record()
        torch.cuda.synchronize()
        t_ms = e0.elapsed_time(e1) / reps
        flops_exec = flops * 0.25
        executed, algorithmic = flops_exec / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12
        traffic, traffic_src = quoted_traffic('lstm44_kernel_hbm_bytes.json', 'conv_wino44.hip', (n, h, w))
        return {'bound': 'mfma', 'kernel': 'wino44_kernel<LSTM> (ConvLSTM cell 128->256 in Winograd F(4x4,3x3) form on transformed inputs, fused gates)',
                'achieved': round(executed, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(executed / PEAK_F32_MFMA_TFLOPS, 4),
                'traffic': traffic, 'traffic_source': traffic_src, 'avg_launch_ms': round(ms, 4), 'executed_mfma_flop_per_launch': flops_exec,
                'algorithmic_flop_per_launch': flops, 'algorithmic_equiv_tflops': round(algorithmic, 2),
                'algorithmic_equiv_frac': round(algorithmic / PEAK_F32_MFMA_TFLOPS, 4),
                # x, h, c in; h', c', gates out - in the reference's formulation; this kernel reads x and h in transform-domain form, 2.25 x their bytes
                'algorithmic_bytes_per_launch': 4 * n * h * w * (cx + 2 * hd + 2 * hd + 4 * hd),
                'formulation_bytes_per_launch': 4 * n * h * w * (2.25 * (cx + hd) + hd + 2 * hd + 4 * hd),
                'input_transform': {'kernel': 'wino44_transform_kernel (B^T d B of the cell output, 64 channels: one launch per cell)', 'avg_launch_ms': round(t_ms, 4),
                                    'bound': 'hbm', 'bytes_per_launch': int(4 * n * h * w * hd * 3.25), 'achieved_tbs': round(4 * n * h * w * hd * 3.25 / (t_ms * 1e-3) / 1e12, 3),
                                    'frac_of_8tbs': round(4 * n * h * w * hd * 3.25 / (t_ms * 1e-3) / 1e12 / PEAK_HBM_TBS, 4)},
                'cell_plus_transform_ms': round(ms + t_ms, 4),
                # the honest figure for "one ConvLSTM cell": the matrix kernel AND the transform launch it cannot run without, against the same peak
                'frac_with_transform': round(flops_exec / ((ms + t_ms) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                'algorithmic_equiv_tflops_with_transform': round(flops / ((ms + t_ms) * 1e-3) / 1e12, 2)}
    # what the matrix cores execute: the Winograd F(2x2,3x3) kernel runs 16 GEMMs over n*h*w/4 tiles = 4/9 of it
    flops_exec = flops * 4.0 / 9.0 if wino else flops
    algorithmic = flops / (ms * 1e-3) / 1e12
    executed = flops_exec / (ms * 1e-3) / 1e12
    traffic, traffic_src = quoted_traffic('lstm_kernel_hbm_bytes.json', 'conv_wino.hip', (n, h, w))
    name = f'{wino_kernel_name()}<LSTM> (ConvLSTM cell 128->256 in Winograd F(2x2,3x3) form, fused gates)' if wino else \
        'conv_igemm_kernel<4,1,1,4,LSTM> (ConvLSTM cell 128->256, fused gates)'
    out = {'bound': 'mfma', 'kernel': name, 'achieved': round(executed, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
           'frac': round(executed / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
           'avg_launch_ms': round(ms, 4), 'executed_mfma_flop_per_launch': flops_exec,
           'algorithmic_flop_per_launch': flops, 'algorithmic_equiv_tflops': round(algorithmic, 2),
           'algorithmic_equiv_frac': round(algorithmic / PEAK_F32_MFMA_TFLOPS, 4),
           'algorithmic_bytes_per_launch': 4 * n * h * w * (cx + 2 * hd + 2 * hd + 4 * hd)}    # x, h, c in; h', c', gates out
    return out


def wino_kernel_name():
    """The Winograd convolution kernel of csrc/conv_wino.hip (rnh_conv_wino)."""
    return 'conv_winoh_kernel'


def kernel_source_sha256(source='conv_wino.hip'):
    """sha256 of the source file of a kernel: the PMC record under profiles/ is only quoted while it matches."""
    import hashlib
    with open(os.path.join(PKG, 'csrc', source), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


def quoted_traffic(record, source, shape=(8, 128, 128)):
    """(HBM bytes per launch, where they come from) out of profiles/<record> (rocprofv3 PMC passes, FETCH_SIZE / WRITE_SIZE corrected as
    MI355X_MICROARCH.md prescribes) - or (None, None) when the record was measured on another revision of csrc/<source>, at another
    launch shape than (N, H, W) = ``shape`` (the records are config 2's launch), or when the library in use is not the product build
    (tools/bench_with_lib.py: a diagnostic build of the same source)."""
    from hipvsr import lib as L
    prof = os.path.join(ROOT, 'profiles', record)
    try:
        rec = json.load(open(prof))
        default_lib = os.path.join(PKG, 'hipvsr', 'librefinenet_hip.so')
        if tuple(shape) != tuple(rec.get('launch_shape', (8, 128, 128))) or os.path.realpath(getattr(L, 'LIB_PATH', default_lib)) != os.path.realpath(default_lib):
            return None, None
        if rec.get('kernel_source_sha256') == kernel_source_sha256(source):
            return rec.get('hbm_bytes_per_launch'), {k: rec.get(k) for k in ('commit', 'date', 'command')}
    except Exception:
        pass
    return None, None


def cpu_model():
    """The host CPU's model string (/proc/cpuinfo) and its socket count."""
    try:
        names, sockets = [], set()
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                names.append(ln.split(':', 1)[1].strip())
            elif ln.startswith('physical id'):
                sockets.add(ln.split(':', 1)[1].strip())
        return f'{max(len(sockets), 1)} x {names[0]}' if names else None
    except OSError:
        return None


def cpu_baseline():
    """The oracle (bit-exact restatement of the reference) at BASELINE config 1 on the host cores: 1 warm-up + 10 timed
    steps of forward + discounted L1 loss + backward (about 10-20 s of CPU work on the GPU box's host)."""
    from oracle import refinenet_oracle as orc
    # 16 threads is the fastest setting on the GPU box's host (2 x EPYC 9575F, 256 hardware threads): 0.88 s/step at
    # 16 threads against 1.37 (8), 1.42 (32), 3.2 (64) and 10.6 (128) - the convolutions of one 64x64 sample are small
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    cfg = orc.exp1_x4_config()
    sd = orc.init_state_dict(cfg, seed=20200526)
    inputs, targets, pos = orc.synthetic_batch(cfg, n=1, t=3, h=64, w=64, seed=20200527)
    orc.step(sd, cfg, [x.clone() for x in inputs], targets, pos)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.step(sd, cfg, [x.clone() for x in inputs], targets, pos)
    dt = (time.perf_counter() - t0) / reps
    return {'value': round(3.0 / dt, 4), 'unit': 'frames/s', 'cores': threads, 'host_cores': os.cpu_count(), 'cpu_model': cpu_model(), 'kind': 'port',
            'sample': f'BASELINE config 1 (x4, N=1, T=3, F=15, 64x64->256x256, fp32), {reps} timed steps after 1 warm-up, '
                      f'{dt:.2f} s/step, PyTorch CPU oracle == reference bit for bit',
            'tflops': round(step_flops_per_lr_pixel(3) * 64 * 64 / dt / 1e12, 3)}


# BASELINE.json configs by their 1-based number: per-GPU batch, supervised frames T, LR size, scale
# 'yaml' = the reference's own training shape (configs/train/refine_net/exp1_x4.yaml:20-33: 16 crops of 32 x 32, num_frames 7): the launch-bound regime
CONFIGS = {2: dict(batch=8, frames=7, size=128, scale=4), 4: dict(batch=16, frames=5, size=256, scale=2), 5: dict(batch=8, frames=11, size=96, scale=4),
           'yaml': dict(batch=16, frames=7, size=32, scale=4)}


def config_label(args, bf):
    c = CONFIGS[args.config]
    if (args.batch, args.frames, args.size, args.scale) != (c['batch'], c['frames'], c['size'], c['scale']):
        return 'a shape of its own: --batch / --frames / --size given'
    if args.config == 'yaml':
        return "the reference YAML's training shape, exp1_x4.yaml:20-33" + (', bf16 storage' if bf else '')
    if args.config == 2:
        return f'BASELINE config {3 if bf else 2}'
    return f'BASELINE config {args.config}' + (', bf16 storage' if bf else '') + (' per GPU' if args.config == 5 else '')


def gates_label(net, args):
    n = net._engine().recompute_stages(args.batch, args.size, args.size, args.frames + 12)
    return 'stored' if n == 0 else f'recomputed in the backward of {n} of {net.num_stages} stages (activation-memory plan: the stored-gates step does not fit)'


def run_case(args, dtype, dev, world, rank):
    """W warm-up steps, then exactly K timed steps of the training step in `dtype`, bracketed by barrier + synchronize on both
    sides; the step time is the max over ranks.  Returns the fields of the JSON line that belong to this case."""
    from hipvsr import dp
    from hipvsr.step_tail import FlatAdam
    from src.runner.trainers import AcdcVSRRefineNetTrainer
    if os.environ.get('BENCH_DUMMY_STREAMS'):        # diagnosis: advance torch's stream pool as an earlier case in the same process would have
        _dummy = [torch.cuda.Stream(dev) for _ in range(int(os.environ['BENCH_DUMMY_STREAMS']))]
    if os.environ.get('BENCH_DUMMY_ALLOC_GB'):       # diagnosis: the memory history an earlier case in the same process would have left
        _blocks = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(int(os.environ['BENCH_DUMMY_ALLOC_GB']))]
        for _b in _blocks:
            _b.fill_(1)
        torch.cuda.synchronize()
        del _blocks, _b
        torch.cuda.empty_cache()
    net = make_net(dev, seed=0, scale=args.scale)
    net.set_compute_dtype(dtype)
    dp.broadcast_parameters(net)
    opt = FlatAdam(net.parameters(), lr=1e-4, weight_decay=0)      # exp1_x4.yaml:56-60; one launch per run of parameters
    tr = object.__new__(AcdcVSRRefineNetTrainer)
    tr.net, tr.loss_fns, tr.metric_fns, tr.optimizer = net, [torch.nn.L1Loss()], [], opt
    tr.loss_weights = torch.tensor([1.0], device=dev)
    tr.graph, tr._graphed = args.graph == 'on', None
    inputs, targets, pos = synthetic_batch(dev, args.batch, args.frames, args.size, args.size, seed=20200526 + (9 if args.config == 'yaml' else args.config) + rank, s=args.scale)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.reset_peak_memory_stats(dev)
    loss = None
    for _ in range(args.warmup):
        _, loss, _ = tr.train_step(inputs, targets, pos)
    barrier()
    dp.allreduce_ms()                                # (forget the warm-up steps' collectives)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step times (no host sync inside the region)
    t0 = time.perf_counter()
    marks[0].record()
    host = []
    for i in range(args.steps):
        h0 = time.perf_counter()
        _, loss, _ = tr.train_step(inputs, targets, pos)
        marks[i + 1].record()
        host.append(time.perf_counter() - h0)
    barrier()
    dt = time.perf_counter() - t0
    if os.environ.get('BENCH_EACH_STEP'):            # diagnosis: the steps in order, to stderr
        print(dtype, 'per-step ms:', [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(args.steps)], 'wall', round((time.perf_counter() - t0) * 1e3, 1), 'host enqueue ms per step:', [round(h * 1e3, 1) for h in host], file=sys.stderr, flush=True)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    # the ranks' own times (median step by HIP events, wall time of the timed region) and the gradient all-reduce's own HIP-event time, so that a
    # scaling run explains itself: a slow rank, or a slow collective, shows in the line
    ar_ms = dp.allreduce_ms()
    mine = torch.tensor([median_ms, dt * 1e3 / args.steps, ar_ms if ar_ms is not None else -1.0], device=dev, dtype=torch.float64)
    per_rank = [mine.clone() for _ in range(world)]
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_gather(per_rank, mine)
    per_rank = [[round(float(v), 3) for v in r.tolist()] for r in per_rank]
    dt = float(tt)
    n_global = args.batch * world
    bf = dtype == 'bf16'
    flop_step = step_flops_per_lr_pixel(args.frames, scale=args.scale) * args.size * args.size * n_global
    # the forms the timed steps ran in: the engine's own record (hipvsr/forms.py) at this shape and mode - incl. the F(2x2) fallback of a captured step at
    # a shape whose F(4x4) cells need the transformed-h' ring
    fm = net._engine().resolve_forms(args.batch, args.size, args.size, args.frames + 12, need_grad=True, capturing=args.graph == 'on')
    cell44, refine44, up44, rd44 = fm.cells44, fm.refine_fwd44, bool(fm.up44 and fm.up44[0]), fm.refine_dgrad44
    flop_exec = step_flops_per_lr_pixel(args.frames, scale=args.scale, executed=True, cell44=cell44, refine44=refine44, up44=up44, refine_dgrad44=rd44,
                                        cell_dgrad44=fm.cell_dgrad44, cell_wgrad44=fm.cell_wgrad44f, refine1_wgrad44=fm.refine1_wgrad44f,
                                        refine2_wgrad44=fm.refine2_wgrad44f, up_wgrad44=fm.up_wgrad44f) * args.size * args.size * n_global
    if bf:      # direct-form convolutions on bf16 MFMA; only the collapsed tail and the skipped dead cells reduce the work
        flop_exec = step_flops_bf16(args.frames, scale=args.scale) * args.size * args.size * n_global
    # gate recomputation: one more cell launch per cell and supervised frame of the recomputing stages
    n_rc = fm.recompute
    flop_exec += n_rc * 2 * args.frames * 3 * 589824 * (1.0 if bf else (0.25 if cell44 else 4.0 / 9.0)) * args.size * args.size * n_global
    peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_F32_MFMA_TFLOPS
    prec = ('bf16 storage + bf16 MFMA, fp32 accumulate (BASELINE config 3 per GPU)' if bf else 'fp32')
    out = None
    if rank == 0:
        roof = lstm_kernel_roofline(net, dev, args.batch, args.size, args.size)
        out = {
            'metric': f'cine-frames/sec fwd+bwd, x{args.scale} SR {args.size}->{args.scale * args.size} T={args.frames}', 'value': round(n_global * args.frames * args.steps / dt, 3), 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 2),
            'ms_per_step_median': round(median_ms, 2), 'ms_per_step_min_max': [round(per_step[0], 2), round(per_step[-1], 2)],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': f'RefineNet x{args.scale} training step (fwd + deep-supervision L1 + bwd + grad all-reduce + Adam), '
                                   f'N={args.batch}/GPU, T={args.frames} (F={args.frames + 12}), {args.size}x{args.size}->'
                                   f'{args.scale * args.size}x{args.scale * args.size}, {prec}, exp1_x4 net'
                                   f'{" with upscale_factor=2" if args.scale == 2 else ""} ({config_label(args, bf)})',
                       'global_batch': n_global, 'frames_per_sample': args.frames, 'parallelism': f'dp{world}',
                       'rccl_world': dist.get_world_size() if dist.is_initialized() else 1,
                       'rank_ms_per_step_median_min_max': [min(r[0] for r in per_rank), max(r[0] for r in per_rank)],
                       'rank_ms_per_step_wall_min_max': [min(r[1] for r in per_rank), max(r[1] for r in per_rank)],
                       'grad_allreduce_ms_min_max': ([min(r[2] for r in per_rank), max(r[2] for r in per_rank)] if per_rank[0][2] >= 0 else None),
                       'cpu_affinity': sorted(os.sched_getaffinity(0))[:4] + ['...', len(os.sched_getaffinity(0))] if hasattr(os, 'sched_getaffinity') else None,
                       'hip_graph_step': tr._graphed is not None,
                       'input_frames_per_s': round(n_global * (args.frames + 12) * args.steps / dt, 2),
                       'step_tflop_reference_formulation': round(flop_step / 1e12, 2),
                       'step_tflop_executed': round(flop_exec / 1e12, 2),
                       'executed_tflops_per_gpu': round(flop_exec / world / (dt / args.steps) / 1e12, 2),
                       ('executed_frac_of_bf16_mfma_peak' if bf else 'executed_frac_of_f32_mfma_peak'):
                           round(flop_exec / world / (dt / args.steps) / 1e12 / peak, 4),
                       'final_loss': round(float(loss.detach()), 6),
                       'gates': gates_label(net, args),
                       'forms': fm.describe(),
                       'peak_hbm_gb': round(torch.cuda.max_memory_allocated(dev) / 2**30, 1)},
            'roofline': roof,
        }
        if bf:      # the whole step against the dense bf16 MFMA peak in the reference's (= this path's, up to the collapsed tail) formulation
            out['config']['algorithmic_frac_of_bf16_mfma_peak'] = round(flop_step / world / (dt / args.steps) / 1e12 / peak, 4)
    del tr, opt, net, inputs, targets, pos
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=lambda v: v if v == 'yaml' else int(v), choices=list(CONFIGS), default=2,
                    help='BASELINE.json config (1-based): 2 = x4, N=8, T=7, 128x128 (the headline; 3 is its bf16 `secondary`), 4 = x2, N=16, T=5, '
                         "256x256, 5 = x4 with phase codes, N=8 per GPU, T=11, 96x96; yaml = the reference YAML's own training shape "
                         '(x4, batch 16 of 32x32 crops, T=7)')
    ap.add_argument('--batch', type=int, default=None, help='samples per GPU (default: the config\'s)')
    ap.add_argument('--frames', type=int, default=None, help='supervised frames T (default: the config\'s)')
    ap.add_argument('--size', type=int, default=None, help='LR height = width (default: the config\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the bf16-storage run that rides along as `secondary`')
    ap.add_argument('--graph', choices=['on', 'off'], default='off',
                    help='replay forward + loss + backward of the step from a HIP graph (hipvsr.graph.GraphedTrainStep, opt-in: no '
                         'measured gain at any benchmarked shape, profiles/ARCHIVE/r02_l_train_shape.txt)')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32',
                    help="f32: the headline (BASELINE config 2, the reference's precision) with the bf16-storage step of BASELINE "
                         "config 3 as `secondary` in the same line; bf16: that bf16 step alone as the line")
    ap.add_argument('--allow-overrides', action='store_true',
                    help='print the line although RNH_* A/B switches are set in the environment (hipvsr/forms.py: SWITCHES); they are listed in config.forms.env_overrides '
                         'either way.  Without this flag a run with overrides is refused: a stray variable on a box must not silently change the headline')
    ap.add_argument('--dry-run', action='store_true',
                    help='no GPU: the launch / rendezvous / barrier / max-over-ranks protocol over gloo with a stub step (what tests/ checks on CPU)')
    ap.add_argument('--dry-run-fail-rank', type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument('--dry-run-slow-rank', type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    c = CONFIGS[args.config]
    args.scale = c['scale']
    for k in ('batch', 'frames', 'size'):
        if getattr(args, k) is None:
            setattr(args, k, c[k])
    return args


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def launcher_command(gpus, argv, port):
    """What `python bench.py --gpus N ...` runs when it is not itself a rank: the driver's own multi-GPU command line."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={gpus}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes(gpus, smi_text=None):
    """NUMA node of GPUs 0 .. gpus-1 as `rocm-smi --showtoponuma` reports it (a child process: this one never touches the GPU); {} if unknown."""
    import re
    if smi_text is None:
        try:
            smi_text = subprocess.run(['rocm-smi', '--showtoponuma'], capture_output=True, text=True, timeout=60).stdout
        except (OSError, subprocess.SubprocessError):
            return {}
    nodes = {}
    for m in re.finditer(r'GPU\[(\d+)\]\s*:\s*\(Topology\) Numa Node:\s*(-?\d+)', smi_text):
        if int(m.group(1)) < gpus and int(m.group(2)) >= 0:
            nodes[int(m.group(1))] = int(m.group(2))
    return nodes


def rank_cpu_map(gpus, nodes=None, node_cpus=None, allowed=None):
    """local rank -> the CPUs it may run on: the CPUs of its GPU's NUMA node (those this process may use), split evenly among the ranks of that
    node, so that eight Python processes that each enqueue ~700-900 launches per step neither migrate between sockets nor share cores.
    {} (no pinning) unless every rank's node is known and has CPUs."""
    nodes = gpu_numa_nodes(gpus) if nodes is None else nodes
    if len(nodes) < gpus:
        return {}
    allowed = set(os.sched_getaffinity(0)) if allowed is None else set(allowed)
    out = {}
    for node in sorted(set(nodes.values())):
        if node_cpus is not None:
            cpus = node_cpus.get(node, [])
        else:
            try:
                cpus = parse_cpulist(open(f'/sys/devices/system/node/node{node}/cpulist').read())
            except OSError:
                cpus = []
        cpus = [c for c in cpus if c in allowed]
        ranks = [r for r in range(gpus) if nodes[r] == node]
        share = len(cpus) // max(len(ranks), 1)
        if share < 1:
            return {}
        for i, r in enumerate(ranks):
            out[r] = cpus[i * share:(i + 1) * share]
    return out


def apply_rank_affinity(local):
    """In a rank, BEFORE anything touches the GPU: take the CPUs the launcher assigned (BENCH_RANK_CPUS, JSON {local rank: [cpu, ...]})."""
    spec = os.environ.get('BENCH_RANK_CPUS')
    if not spec or not hasattr(os, 'sched_setaffinity'):
        return None
    cpus = json.loads(spec).get(str(local))
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)
            torch.set_num_threads(max(1, min(len(cpus), int(os.environ.get('OMP_NUM_THREADS', '8')))))
        except OSError:
            return None
    return cpus


def launch_ranks(args, argv):
    """--gpus N > 1 without WORLD_SIZE: start the N ranks as a child torchrun (this process never touches the GPU), relay their output -
    rank 0's JSON line - and return the children's exit code.  Each rank is pinned to a share of the CPUs of its GPU's NUMA node."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cpus = {} if args.dry_run else rank_cpu_map(args.gpus)
    if cpus:
        env['BENCH_RANK_CPUS'] = json.dumps({str(k): v for k, v in cpus.items()})
    env.setdefault('OMP_NUM_THREADS', str(max(1, min(8, min((len(v) for v in cpus.values()), default=8)))))
    proc = subprocess.run(launcher_command(args.gpus, argv, free_port()), env=env)
    return proc.returncode


def dry_run(args, world, rank):
    """The multi-rank protocol of run_case without the GPU: process group (gloo), W + K stub steps, barrier on both sides, max over ranks."""
    if 'RANK' in os.environ:
        dist.init_process_group('gloo')
    if rank == args.dry_run_fail_rank:
        raise SystemExit(3)
    bar = (lambda: dist.barrier()) if dist.is_initialized() else (lambda: None)
    for _ in range(args.warmup):
        time.sleep(0.001)
    bar()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank) + (0.05 if rank == args.dry_run_slow_rank else 0.0))
    own = time.perf_counter() - t0                # this rank's own steps, without the wait for the others
    bar()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    per_rank = [torch.tensor([own * 1e3 / max(args.steps, 1)], dtype=torch.float64) for _ in range(world)]
    if dist.is_initialized():
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_gather(per_rank, per_rank[rank].clone())
    if rank == 0:
        print(json.dumps({'metric': 'dry run (no GPU work)', 'value': round(args.batch * world * args.frames * args.steps / float(tt), 3), 'unit': 'frames/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(float(tt) / args.steps * 1e3, 3),
                          'dry_run': True, 'config': {'global_batch': args.batch * world, 'parallelism': f'dp{world}',
                                                      'rank_ms_per_step_wall_min_max': [round(min(float(r) for r in per_rank), 3), round(max(float(r) for r in per_rank), 3)],
                                                      'rccl_world': dist.get_world_size() if dist.is_initialized() else 1,
                                                      'local_rank_env': os.environ.get('LOCAL_RANK'), 'master_addr': os.environ.get('MASTER_ADDR')}}), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ:
        if args.gpus > 1:
            # not a rank: become the launcher.  Nothing above has initialised the GPU (importing torch does not), and the ranks are
            # child processes - never an exec of a process that has touched the device
            raise SystemExit(launch_ranks(args, argv))
        world, rank, local = 1, 0, 0
    else:
        world = int(os.environ['WORLD_SIZE'])
        rank = int(os.environ.get('RANK', '0'))
        local = int(os.environ.get('LOCAL_RANK', '0'))
        if world != args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree; '
                             f'run `python bench.py --gpus {args.gpus}` (it starts the ranks itself) or torchrun with --nproc-per-node {args.gpus}')
    if args.dry_run:
        return dry_run(args, world, rank)
    from hipvsr.forms import env_overrides
    if env_overrides() and not args.allow_overrides:
        raise SystemExit(f'bench.py: A/B switches are set in the environment ({", ".join(env_overrides())}): this would not be the product\'s own step. '
                         f'Unset them, or pass --allow-overrides (they are then recorded in config.forms.env_overrides).')
    apply_rank_affinity(local)                    # (before the first GPU call of this process)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    dev = torch.device(f'cuda:{local}')
    torch.cuda.set_device(dev)
    if 'RANK' in os.environ:                      # launched by torchrun (also with one rank): RCCL process group
        # librccl prints a five-line banner (versions, host, library path) to STDOUT when its first communicator is made; rank 0's stdout is
        # the ONE JSON line of the contract, so the communicator is made here, with fd 1 pointed at stderr meanwhile
        if os.environ.get('BENCH_TOUCH_FIRST', '1') != '0':
            # the engine's side streams are used once, in the order a step first uses them, BEFORE RCCL makes its stream: which streams share a
            # hardware queue is decided by first use (DESIGN.md 4d d), and this keeps the pairing of the single-process run
            from hipvsr.hip_ops import touch_side_streams
            touch_side_streams(dev)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group('nccl', device_id=dev)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            os.dup2(saved, 1)
            os.close(saved)

    out = run_case(args, args.dtype, dev, world, rank)
    sec = None
    if args.dtype == 'f32' and not args.no_secondary:
        sec = run_case(args, 'bf16', dev, world, rank)
    if rank == 0:
        if sec is not None:
            out['secondary'] = {k: sec[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'ms_per_step_median', 'ms_per_step_min_max',
                                                    'dtype', 'data', 'config', 'roofline')}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
