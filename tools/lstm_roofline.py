"""The roofline launch of bench.py on its own: 400 back-to-back launches of the ConvLSTM cell kernel at BASELINE config 2's
shape, timed with HIP events (bench.lstm_kernel_roofline).  Run under `rocprofv3 --kernel-trace --stats` to obtain the
profiler's average duration of the same launches (tools/prof_lstm_isolated.sh)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench import lstm_kernel_roofline, make_net  # noqa: E402

dev = torch.device('cuda:0')
net = make_net(dev)
print(json.dumps(lstm_kernel_roofline(net, dev, 8, 128, 128, reps=400)))
