#!/usr/bin/env python3
"""Per-layer timing of the BraTS-shaped forward (HIP events between kernels): python tools/layer_report.py [forwards] [samples per launch, default 160]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    n = int(sys.argv[2]) if len(sys.argv) > 2 else bench.SLICES
    dev = torch.device('cuda')
    model = bench.make_model(20, dev)
    x = bench.make_volume(20)[0]
    x = x.repeat((n + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:n].to(dev)          # more samples than slices: the volume again (a pass group's batch)
    from rcu_amd import steps
    steps.set_dropout_mode(model, True)
    for _ in range(2):
        model(x)
    model.profile_begin(bench.HEIGHT, bench.WIDTH, n, reps)
    for _ in range(reps):
        model(x)
    torch.cuda.synchronize()
    cnt, ms = model.profile_collect(bench.HEIGHT, bench.WIDTH, n)
    layers = model.layer_table(bench.HEIGHT, bench.WIDTH, n)
    print('{:<52} {:>4}->{:<4} {:>3}x{:<3} {:<34} {:>8} {:>7} {:>6}'.format('layer', 'cin', 'cout', 'H', 'W', 'kernel', 'ms', 'TF/s', '%peak') + '  pipe%')
    tot_ms = tot_fl = tot_is = 0.0
    for L, t in zip(layers, ms[1:1 + len(layers)]):
        t /= cnt
        fl = L['flops_per_slice'] * n
        tot_ms += t
        tot_fl += fl
        issued = L['mfma_flops_per_slice'] * n
        tot_is += issued
        print('{:<52} {:>4}->{:<4} {:>3}x{:<3} {:<34} {:>8.3f} {:>7.1f} {:>6.1f} {:>6.1f}'.format(
            L['name'][:52], L['cin'], L['cout'], L['height'], L['width'], L['kernel'], t, fl / t / 1e9, fl / t / 1e9 / 1.573,
            issued / t / 1e9 / 1.573))
    print('input re-layout {:.3f} ms, head {:.3f} ms'.format(ms[0] / cnt, ms[-1] / cnt))
    print('conv total {:.3f} ms  {:.1f} TF/s algorithmic ({:.1f}% of 157.3); MFMA pipe issue {:.1f}%'.format(
        tot_ms, tot_fl / tot_ms / 1e9, tot_fl / tot_ms / 1e9 / 1.573, tot_is / tot_ms / 1e9 / 1.573))


if __name__ == '__main__':
    main()
