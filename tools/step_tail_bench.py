"""Times the two kernels of rows f3 / f4 at BASELINE config 2's sizes with HIP events on the launch stream:
rnh_metrics_psnr_ssim over 7 x 8 image pairs of 512 x 512 (algorithmic bytes: 8 per pixel pair) and rnh_adam_step over
RefineNet's 2 890 993 parameters (28 bytes per parameter), torch.optim.Adam (the reference's optimizer) beside it.
  python tools/step_tail_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from hipvsr.step_tail import FlatAdam, psnr_ssim  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device('cuda:0')
    P, H = 56, 512
    o, y = torch.randn(P, H, H, device=dev), torch.randn(P, H, H, device=dev)
    for ssim in (True, False):
        ms = timed(lambda: psnr_ssim(o, y, P, 1, H, H, (54.089, 48.084), want_ssim=ssim))
        print(f'metrics (want_ssim={ssim}): {ms * 1e3:.1f} us per step, {8 * P * H * H / ms / 1e6:.0f} GB/s algorithmic')
    from bench import make_net
    for cls in (FlatAdam, torch.optim.Adam):
        net = make_net(dev)
        opt = cls(net.parameters(), lr=1e-4, weight_decay=0)
        n = sum(p.numel() for p in net.parameters())
        flat = torch.randn(n, device=dev)
        off = 0
        for k, p in net.named_parameters():
            if k != 'refine_block.prelu.weight':
                p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        ms = timed(opt.step)
        print(f'{cls.__name__}: {ms * 1e3:.1f} us per step ({getattr(opt, "launches", "-")} launches), {28 * n / ms / 1e6:.0f} GB/s algorithmic')


if __name__ == '__main__':
    main()
