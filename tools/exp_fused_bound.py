"""Timing experiment (results are garbage, only the clock counts): an UPPER BOUND on what fusing the fp32 gate backward into the Winograd
data gradient's epilogue could save.  The fused form (built for the bf16 kernel, csrc/conv_bf16.hip RNH_EPI_LSTM_BWD)
removes exactly one thing from the two-launch form: the round trip of dh_rec, the recurrent part of the previous frame's dh, through HBM
(stored by the data-gradient launch, read back by rnh_lstm_gates_bwd) - every other load and store of the gate backward moves into the
epilogue unchanged.  Here both launches stay, but after the warm-up the data gradient no longer stores dh_rec and the gate backward no
longer reads it: the step then runs with the memory traffic of the fused form (plus the launch that fusion would also remove: +126 x ~2 us).

  python tools/exp_fused_bound.py -- --steps 10 --warmup 3 --no-cpu-baseline --no-secondary
"""
import sys

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import bench                                                    # noqa: E402  (puts the package on sys.path)
from hipvsr import hip_ops                                      # noqa: E402

rest = sys.argv[sys.argv.index('--') + 1:]
steps_real = 3                                                  # the warm-up steps run the real thing
conv, gates = hip_ops.HipOps.conv, hip_ops.HipOps.lstm_gates_bwd
state = {'bwd': 0, 'cut_conv': 0, 'cut_gates': 0}


def conv_(self, plan, srcs, B, H, W, dsts=None, **k):
    if plan.name.endswith('.dgrad') and plan.name[:-6].rstrip('0123456789') in ('forward', 'backward') and dsts is not None and len(dsts) == 2 \
            and state['bwd'] > steps_real * 126:
        dsts = dsts[:1]
        state['cut_conv'] += 1
    return conv(self, plan, srcs, B, H, W, dsts=dsts, **k)


def gates_(self, dh, dc_next, g, c_prev, c_next, dgates, dc_prev, dh2=None):
    state['bwd'] += 1
    if dh2 is not None and state['bwd'] > steps_real * 126:
        dh2 = None
        state['cut_gates'] += 1
    return gates(self, dh, dc_next, g, c_prev, c_next, dgates, dc_prev, dh2=dh2)


hip_ops.HipOps.conv, hip_ops.HipOps.lstm_gates_bwd = conv_, gates_
sys.argv = ['bench.py'] + rest
bench.main()
print('gate-backward launches', state['bwd'], '- dh_rec stores removed', state['cut_conv'], ', dh_rec loads removed', state['cut_gates'], file=sys.stderr)
