import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd', '/root/repo/tests']
import torch
from oracle import refinenet_oracle as orc
import test_parity_r05 as t5
for name, idx in (('x8', 0), ('no_phase_code', 1), ('no_memory', 2)):
    over, n, t, h, w = t5._VARIANTS[name]
    cfg = orc.exp1_x4_config(**over)
    sd = orc.init_state_dict(cfg, seed=500 + idx)
    inputs, targets, pos = orc.synthetic_batch(cfg, n, t, h, w, seed=600 + idx)
    torch.set_num_threads(32)
    ref_out, ref_loss, ref_grads = orc.step(sd, cfg, [x.clone() for x in inputs], targets, pos)
    for f16 in ('0', '1'):
        os.environ['RNH_UP_F16'] = f16
        net, tr, outs, loss = t5._module_step(dict(cfg), sd, inputs, targets, pos, 'bf16')
        rels = {k: float((p.grad.cpu() - ref_grads[k]).norm()) / float(ref_grads[k].norm()) for k, p in net.named_parameters() if ref_grads[k] is not None}
        worst = sorted(rels.items(), key=lambda kv: -kv[1])[:3]
        print(name, 'RNH_UP_F16=' + f16, 'loss rel', abs(float(loss) - float(ref_loss)) / float(ref_loss), 'worst grads', [(k, round(v, 4)) for k, v in worst], flush=True)
