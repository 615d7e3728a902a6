"""Numerical ablation of the bf16-storage path's PSNR deviation at trained weights (VERDICT r05 "next" item 1).

TEST INFRASTRUCTURE (it imports oracle/ as the checker; nothing here is product code).  Two sub-commands, both on the GPU box:

  train   three fp32 training runs of 600 steps at the reference YAML's training shape (exp1_x4.yaml:21-33; the `trained` fixture of
          tests/test_parity_r04.py): A = seed 61 in the tree's default forms (the weights the round-5 tests end at), B = seed 61 with every
          Winograd launch in F(2x2) form (the round-4 trajectory's forms), C = seed 161.  State dicts -> <out>/weights_{A,B,C}.pt
  ablate  for each weight set and BASELINE config 1 / config 2's geometry on the structured cine: PSNR of the fused group of the last stage
          against the TRUE HR frames through the fp32 oracle (== reference), the HIP fp32 path, the HIP bf16-storage path, and the bf16 path
          with one class of stored tensors at a time kept in fp32 (hipvsr.engine.RefineNetEngine.STORAGE_CLASSES) - on the real kernels
          (backend hip) and on the rounding torch double of the kernel interface (tests/torch_ops.py, backend double; there also with
          unrounded weights / unrounded MFMA operands) - a table of per-class dPSNR.
"""
import argparse
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import refinenet_oracle as orc          # noqa: E402
from oracle import step_tail_oracle as sto          # noqa: E402

TRAIN = dict(steps=600, batch=16, crop=32, t=7, pool=64, lr=1e-4)
SETS = {'A': dict(seed=61, env={}), 'B': dict(seed=61, env={'RNH_WINO44': '0'}), 'C': dict(seed=161, env={})}
CASES = [('config 1', 1, 3, 64), ('config 2 geometry', 2, 7, 128)]


def _net(cfg, sd, dtype, storage=None):
    from src.model.nets import RefineNet
    net = RefineNet(**dict(cfg))
    net.load_state_dict(sd)
    return net.to('cuda:0').set_compute_dtype(dtype).set_storage(storage)


def train(out):
    import functools
    from hipvsr.step_tail import FlatAdam
    from src.model.metrics import PSNR
    from src.runner.trainers import AcdcVSRRefineNetTrainer
    from src.utils import denormalize
    os.makedirs(out, exist_ok=True)
    cfg, c, dev = orc.exp1_x4_config(), TRAIN, torch.device('cuda:0')
    for name, spec in SETS.items():
        for k, v in spec['env'].items():
            os.environ[k] = v
        sd0 = orc.init_state_dict(cfg, seed=spec['seed'])
        pool = orc.structured_cine(cfg, c['pool'], c['t'], c['crop'], c['crop'], seed=spec['seed'] + 1)
        pin, ptg, ppos = [x.to(dev) for x in pool[0]], [y.to(dev) for y in pool[1]], pool[2].to(dev)
        net = _net(cfg, sd0, 'f32').train()
        tr = object.__new__(AcdcVSRRefineNetTrainer)
        tr.net, tr.loss_fns, tr.metric_fns = net, [torch.nn.L1Loss()], [PSNR().to(dev)]
        tr._denormalize = functools.partial(denormalize, dataset='acdc')
        tr.optimizer = FlatAdam(net.parameters(), lr=c['lr'], weight_decay=0)
        tr.loss_weights = torch.tensor([1.0], device=dev)
        tr.graph, tr._graphed = False, None
        nb, losses = c['pool'] // c['batch'], []
        t0 = time.time()
        for i in range(c['steps']):
            sl = slice((i % nb) * c['batch'], (i % nb + 1) * c['batch'])
            _, loss, _ = tr.train_step([x[sl] for x in pin], [y[sl] for y in ptg], ppos[sl])
            losses.append(loss.detach())
        torch.cuda.synchronize()
        losses = [float(x) for x in losses]
        sd = {k: p.detach().cpu().clone() for k, p in net.state_dict().items()}
        torch.save(dict(sd=sd, seed=spec['seed'], env=spec['env'], losses=losses), os.path.join(out, f'weights_{name}.pt'))
        print(f'weights {name}: seed {spec["seed"]} env {spec["env"]}: loss {losses[0]:.4f} -> {sum(losses[-25:]) / 25:.4f} in {time.time() - t0:.0f} s', flush=True)
        for k in spec['env']:
            os.environ.pop(k)
        del net, tr
        torch.cuda.empty_cache()


def psnr_frames(last, targets):
    """PSNR per supervised frame as the reference's trainer computes it (oracle/step_tail_oracle.py), on the CPU, in one place for all variants."""
    return [float(sto.trainer_metrics([o.detach().float().cpu()], [y])[0]) for o, y in zip(last, targets)]


def run_hip(cfg, sd, dtype, storage, inputs, pos):
    net = _net(cfg, sd, dtype, storage).eval()
    with torch.no_grad():
        outs = net([x.to('cuda:0') for x in inputs], pos.to('cuda:0'))
    last = [o.clone() for o in outs[-1]]
    del net
    return last


def run_double(cfg, sd, storage, inputs, pos, mode=''):
    """The bf16-storage engine over the rounding torch double (on the GPU through ATen - a checker, not the product); mode: 'w32' = weights
    not rounded, 'x32' = the convolutions' input operands not rounded (both: the rounding of stores alone)."""
    from hipvsr.engine import RefineNetEngine
    from hipvsr.spec import NetConfig
    from torch_ops import TorchOps, effective_weight, gather_src

    only = [m[5:] for m in mode.split() if m.startswith('only:')]          # weights rounded in these plan groups alone

    def diffuse(weff):
        """bf16 rounding with the rounding error carried from tap to tap of each (column, channel) filter: the filter's tap sum (its DC gain) is
        kept to half an ulp of ONE element instead of accumulating nine independent errors."""
        v = weff.reshape(*weff.shape[:2], -1)
        out, carry = torch.empty_like(v), torch.zeros_like(v[..., 0])
        for t in range(v.shape[-1]):
            x = v[..., t] + carry
            r = x.bfloat16().float()
            out[..., t], carry = r, x - r
        return out.reshape(weff.shape)

    class Ops(TorchOps):
        def pack(self, plan, w, b=None, **forms):
            super().pack(plan, w, b)
            weff, bp = self._w[id(plan)]
            exact = effective_weight(plan, w.detach())
            if 'w32' in mode or (only and not any(plan.name.startswith(o) or (o == 'lstm' and plan.name[:3] in ('for', 'bac')) for o in only)):
                self._w[id(plan)] = (exact, bp)
            elif 'wdiff' in mode:
                self._w[id(plan)] = (diffuse(exact), bp)

        @staticmethod
        def _r(t, on):
            return t.float() if 'x32' in mode else TorchOps._r(t, on)

    ops = Ops('cuda:0')
    eng = RefineNetEngine(NetConfig(**dict(cfg)), ops, dtype='bf16', storage=storage)
    params = {k: v.to('cuda:0') for k, v in sd.items()}
    with torch.no_grad():
        O_all, _ = eng.forward(params, inputs, pos.to('cuda:0'), need_grad=False)
    N = inputs[0].shape[0]
    T = O_all.shape[2] // N
    return [O_all[-1, 2, i * N:(i + 1) * N].permute(0, 3, 1, 2).clone() for i in range(T)]


def ablate(wdir, out, sets, backends):
    cfg = orc.exp1_x4_config()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    # full-precision matrix products in the double (ATen on the GPU): no TF32-like shortcuts
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    classes = ['feat', 'h', 'r1', 'r', 'sb', 'ys']
    for name in sets:
        w = torch.load(os.path.join(wdir, f'weights_{name}.pt'), weights_only=False)
        sd = w['sd']
        for cname, n, t, size in CASES:
            inputs, targets, pos = orc.structured_cine(cfg, n, t, size, size, seed=71)
            t0 = time.time()
            with torch.no_grad():
                ref = orc.forward(orc.as_leaf_params(sd), cfg, [x.clone() for x in inputs], pos)
            want = psnr_frames(ref[-1], targets)
            say(f'== weights {name} (seed {w["seed"]}, env {w["env"]}), {cname}: oracle PSNR vs true HR {sum(want) / len(want):.4f} dB ({time.time() - t0:.0f} s)')

            def row(label, last):
                got = psnr_frames(last, targets)
                d = [a - b for a, b in zip(got, want)]
                say(f'   {label:<44s} dPSNR mean {sum(d) / len(d):+.4f}  worst |d| {max(abs(x) for x in d):.4f}   per frame ' + ' '.join(f'{x:+.4f}' for x in d))

            if 'hip' in backends:
                row('hip f32', run_hip(cfg, sd, 'f32', None, inputs, pos))
                row('hip bf16', run_hip(cfg, sd, 'bf16', None, inputs, pos))
                for c in classes:
                    try:
                        row(f'hip bf16, {c} stored in fp32', run_hip(cfg, sd, 'bf16', {c: 'f32'}, inputs, pos))
                    except Exception as e:                                      # a kernel that does not take that element type: say so, go on
                        torch.cuda.synchronize()
                        say(f'   hip bf16, {c} stored in fp32: not runnable ({type(e).__name__}: {str(e)[:120]})')
                try:
                    row('hip bf16, all classes in fp32', run_hip(cfg, sd, 'bf16', {c: 'f32' for c in classes}, inputs, pos))
                except Exception as e:
                    torch.cuda.synchronize()
                    say(f'   hip bf16, all classes in fp32: not runnable ({type(e).__name__}: {str(e)[:120]})')
            if 'double' in backends:
                row('double bf16', run_double(cfg, sd, None, inputs, pos))
                for c in classes:
                    row(f'double bf16, {c} stored in fp32', run_double(cfg, sd, {c: 'f32'}, inputs, pos))
                allf = {c: 'f32' for c in classes}
                row('double bf16, all classes in fp32', run_double(cfg, sd, allf, inputs, pos))
                row('double bf16, weights unrounded', run_double(cfg, sd, None, inputs, pos, 'w32'))
                row('double bf16, conv inputs unrounded', run_double(cfg, sd, None, inputs, pos, 'x32'))
                row('double: only the weights rounded', run_double(cfg, sd, allf, inputs, pos, 'x32'))
                row('double: nothing rounded (sanity)', run_double(cfg, sd, allf, inputs, pos, 'w32 x32'))
            if 'hipf16' in backends:
                # the product (the upsampler's forward with IEEE-half weights on the f16 MFMA forms) against the round-5 form of the conv (RNH_UP_F16=0;
                # the collapsed tail has no switch: 'ys stored in fp32' sends it to the fp32 kernel)
                row('hip bf16', run_hip(cfg, sd, 'bf16', None, inputs, pos))
                row('hip bf16, ys stored in fp32 (fp32 tail)', run_hip(cfg, sd, 'bf16', {'ys': 'f32'}, inputs, pos))
                os.environ['RNH_UP_F16'] = '0'
                row('hip bf16, up1 with bf16 weights', run_hip(cfg, sd, 'bf16', None, inputs, pos))
                os.environ.pop('RNH_UP_F16')
            if 'weights' in backends:
                # which layers' weight rounding carries the shift (everything else exact), and what error-diffusion rounding makes of it
                allf = {c: 'f32' for c in classes}
                row('double: only the weights rounded', run_double(cfg, sd, allf, inputs, pos, 'x32'))
                for g in ('lstm', 'refine1', 'refine2', 'up1'):
                    row(f'double: only the weights of {g} rounded', run_double(cfg, sd, allf, inputs, pos, f'x32 only:{g}'))
                row('double: only the weights rounded, diffused', run_double(cfg, sd, allf, inputs, pos, 'x32 wdiff'))
                for g in ('lstm', 'refine1', 'refine2', 'up1'):
                    row(f'double: only the weights of {g}, diffused', run_double(cfg, sd, allf, inputs, pos, f'x32 wdiff only:{g}'))
                row('double bf16 (up1: fp16 weights)', run_double(cfg, sd, None, inputs, pos))
                row('double bf16, weights diffused', run_double(cfg, sd, None, inputs, pos, 'wdiff'))
    if out:
        with open(out, 'w') as f:
            f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('cmd', choices=['train', 'ablate'])
    ap.add_argument('--weights', default='gpurun_out/r06_weights')
    ap.add_argument('--out', default='')
    ap.add_argument('--sets', default='A,B,C')
    ap.add_argument('--backends', default='hip,double')
    a = ap.parse_args()
    if a.cmd == 'train':
        train(a.weights)
    else:
        ablate(a.weights, a.out, a.sets.split(','), a.backends.split(','))
