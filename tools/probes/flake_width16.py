#!/usr/bin/env python3
"""Is one fp32 training step of the width-16 net (generic refine path, pixel-contraction weight gradients) bitwise repeatable?  Runs the
first step of tests/test_parity_r04.py::test_helper_stream_changes_no_bit_even_when_it_runs_late REPS times from the same state and
lists which gradients ever differ from the first run.  Environment switches (RNH_ASIDE, RNH_DEFER_WGRAD, RNH_LSTM_STREAMS, ...) bisect.
GPU box only.   python tools/probes/flake_width16.py [reps] [width] [dtype]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch                                                  # noqa: E402
from hipvsr import lib as _L                                  # noqa: E402
if os.environ.get('RNH_LIB'):
    _L.LIB_PATH = os.environ['RNH_LIB']                        # a diagnostic build
import test_parity_r04 as T                                   # noqa: E402
from oracle import refinenet_oracle as orc                    # noqa: E402  (inputs and the initial state only)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
width = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dtype = sys.argv[3] if len(sys.argv) > 3 else 'f32'
cfg = orc.Config(in_channels=1, out_channels=1, num_features=[width, width], num_stages=3, refine_window_size=5, upscale_factor=4,
                 update_memory=True, num_updated_frames=2, positional_encoding=True)
sd = orc.init_state_dict(cfg, seed=8)
dev = torch.device('cuda:0')
size = int(os.environ.get('PROBE_SIZE', '32'))
inputs, targets, pos = orc.synthetic_batch(cfg, 2, 3, size, size, seed=90)
cold = os.environ.get('PROBE_COLD', '1') == '1'               # a fresh net (engine, streams, scratch buffers) for every repetition, as the test has
net = tr = None
first, bad, nbad = None, {}, 0
for r in range(reps):
    if cold or net is None:
        del net, tr
        net = T._net(cfg, sd, dtype).train()
        tr = T._train_trainer(net, 1e-3)
    _, loss, _ = tr.train_step([x.to(dev) for x in inputs], [t.to(dev) for t in targets], pos.to(dev))
    torch.cuda.synchronize()
    cur = {'loss': loss.detach().clone()}
    eng = getattr(net, '_eng', None) or getattr(net, 'engine', None)
    for obj in [net] + list(net.__dict__.values()):
        if hasattr(obj, 'dbgsum'):
            cur.update({'~' + k: v.clone() for k, v in obj.dbgsum})
            obj.dbgsum.clear()
    ops_ = net._eng.ops
    if os.environ.get('RNH_WS_GUARD') == '1':
        g = ops_.check_ws_guards()
        if g:
            print('rep', r, 'scratch guard bands written:', g, flush=True)
    keep = dict(getattr(net._eng, 'dbgkeep', []))
    if keep:
        cur.update({'~~keep ' + k: v for k, v in keep.items()})          # (references: whatever the buffers hold at the end of the step)
        net._eng.dbgkeep.clear()
    ent = ops_._ws.get(('halo', 'dR1p'))
    if ent is not None:
        cur['~~dR1p at the end'] = ent[1].clone()
    for pid, pk in ops_._packed.items():
        nm = ops_._maps[pid]['_plan'].name if pid in ops_._maps else str(pid)
        for j, t in enumerate(pk):
            if t is not None:
                cur['~~packed %s [%d]' % (nm, j)] = t.clone()
    for pid, mp in ops_._maps.items():
        for kk, t in mp.items():
            if kk != '_plan':
                cur['~~map %s %s' % (mp['_plan'].name, kk)] = t.clone()
    cur.update({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
    if first is None:
        first = cur
        continue
    diff = [k for k in cur if not torch.equal(cur[k], first[k])]
    if diff:
        nbad += 1
        for k in diff:
            bad[k] = bad.get(k, 0) + 1
        if nbad <= 6 or ('~bwd 1 gsrc' not in diff and bad.get('__analysed', 0) < 5):
            if '~bwd 1 gsrc' not in diff:
                bad['__analysed'] = bad.get('__analysed', 0) + 1
            print('   ', [(k, float(first[k]), float(cur[k])) for k in diff if k.startswith('~') and not k.startswith('~~')][:3])
            if '~~keep bwd 0 dS' in cur and '~bwd 1 gsrc' not in diff:
                import torch.nn.functional as Fn
                w2 = sd['refine_block.body.conv2.weight'].double()
                TN_ = cur['~~keep bwd 0 dS'].shape[0] // 3
                for tag, src in (('this run', cur), ('first run', first)):
                    dR_ = src['~~keep bwd 0 dS'][2 * TN_:].cpu().double()
                    want = Fn.conv_transpose2d(dR_.permute(0, 3, 1, 2), w2, padding=1).permute(0, 2, 3, 1)
                    got = src['~~dR1p at the end'][4:4 + TN_, ..., :w2.shape[1]].cpu().double()
                    dfn = src['~~keep bwd 1 dfeat'].cpu().double()
                    want0 = Fn.conv_transpose2d((dR_ - dfn).permute(0, 3, 1, 2), w2, padding=1).permute(0, 2, 3, 1)
                    print('    ', tag, ': |dR1p - convT(dR at end)| max', float((got - want).abs().max()), ' |dR1p - convT(dR - dfeat_next)| max', float((got - want0).abs().max()),
                          'per image', [float((got[i] - want[i]).abs().max()) for i in range(TN_)])
                    print('         vs convT(dR - dfeat_next) per image', [float((got[i] - want0[i]).abs().max()) for i in range(TN_)], ' |want| per image', [float(want[i].abs().max()) for i in range(TN_)])
                    bad_i = max(range(TN_), key=lambda i: float((got[i] - want[i]).abs().max()))
                    e = (got[bad_i] - want[bad_i]).abs()
                    nz = (e > 1e-9).nonzero()
                    if nz.shape[0]:
                        m_ = e > 1e-9
                        print('         on the wrong elements of image', bad_i, ': max |got - convT(dR before += dfeat_next)| =', float((got[bad_i] - want0[bad_i]).abs()[m_].max()),
                              ' max |got - convT(final dR)| =', float(e[m_].max()), ' max |got| there', float(got[bad_i].abs()[m_].max()))
                    print('         worst image', bad_i, ': wrong elements', nz.shape[0], 'of', e.numel(), 'rows', sorted(set(nz[:, 0].tolist())), 'cols', sorted(set(nz[:, 1].tolist())), 'chans', sorted(set(nz[:, 2].tolist())))
            if '~~dR1p at the end' in diff:
                a, b = first['~~dR1p at the end'], cur['~~dR1p at the end']
                idx = (a != b).nonzero()
                print('    dR1p differs in', idx.shape[0], 'elements of', a.numel(), 'shape', tuple(a.shape), '; frames', sorted(set(idx[:, 0].tolist())), 'rows', sorted(set(idx[:, 1].tolist()))[:40],
                      'cols', sorted(set(idx[:, 2].tolist()))[:40], 'channels', sorted(set(idx[:, 3].tolist())), 'max abs diff', float((a - b).abs().max()), 'max |a|', float(a.abs().max()))
            print('rep', r, 'differs in', len(diff), 'tensors; intermediate sums that differ:', [k for k in diff if k.startswith('~') and ('packed' in k or 'map' in k or 'gsrc' in k)], 'of', sum(k.startswith('~') for k in cur), flush=True)
nan = [k for k, v in first.items() if not bool(torch.isfinite(v).all())]
print('non-finite tensors of the first run:', nan[:20])
print('env', {k: v for k, v in os.environ.items() if k.startswith('RNH_')}, 'reps', reps, 'width', width, dtype, 'runs that differ:', nbad)
for k, v in sorted(bad.items(), key=lambda kv: -kv[1])[:40]:
    print('   %4d  %s' % (v, k))
