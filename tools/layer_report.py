#!/usr/bin/env python3
"""Per-layer timing of the BraTS-shaped forward (HIP events between kernels):
    python tools/layer_report.py [forwards] [samples per launch, default 160] [stats] [option=value ...]
`stats`: the product's launch -- the samples are pass groups of the 160-slice volume adding into exact MC statistics (MI on), as the MC step
launches them -- instead of a logits forward; option=value: plan options (rcu_unet_options, e.g. head_winograd4=0, pad_levels=0).
shape=HxW [cin=C]: another slice size than the benchmark's 192x128 (the reference's real data: shape=240x240 with 155 samples per launch is a
native BraTS volume, shape=192x256 cin=3 an ISIC batch); the input is then plain N(0,1) noise."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    n = int(sys.argv[2]) if len(sys.argv) > 2 else bench.SLICES
    rest = sys.argv[3:]
    as_stats = 'stats' in rest
    dev = torch.device('cuda')
    kv = dict(a.split('=') for a in rest if '=' in a)
    shape = kv.pop('shape', None)
    cin = int(kv.pop('cin', bench.CHANNELS))
    model = bench.make_model(20, dev, params=dict(bench.MODEL_PARAMS, in_channels=cin))
    model.plan_options = {k: int(v) for k, v in kv.items()}
    if shape:
        bench.HEIGHT, bench.WIDTH = (int(v) for v in shape.split('x'))
        x = torch.randn(min(n, 160), cin, bench.HEIGHT, bench.WIDTH, generator=torch.Generator().manual_seed(20))
    else:
        x = bench.make_volume(20)[0]
    from rcu_amd import steps
    steps.set_dropout_mode(model, True)
    if as_stats:
        passes = max(n // x.shape[0], 1)
        x = x.to(dev)
        n = passes * x.shape[0]
        stats = steps.McStatistics(x.shape[0], 2, bench.HEIGHT, bench.WIDTH, dev, do_mi=True, do_var=False, exact=True)
        run = lambda: model.forward_accumulate(x, stats, passes=passes)  # noqa: E731
    else:
        x = x.repeat((n + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:n].to(dev)          # more samples than slices: the volume again (a pass group's batch)
        run = lambda: model(x)  # noqa: E731
    for _ in range(2):
        run()
    model.profile_begin(bench.HEIGHT, bench.WIDTH, n, reps)
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    cnt, ms = model.profile_collect(bench.HEIGHT, bench.WIDTH, n)
    layers = model.layer_table(bench.HEIGHT, bench.WIDTH, n)
    print('{} samples of {} x {} x {} per launch'.format(n, cin, bench.HEIGHT, bench.WIDTH))
    print('{:<52} {:>4}->{:<4} {:>3}x{:<3} {:>7} {:<34} {:>8} {:>7} {:>6}'.format('layer', 'cin', 'cout', 'H', 'W', 'grid', 'kernel', 'ms', 'TF/s', '%peak') + '  pipe%')
    tot_ms = tot_fl = tot_is = 0.0
    for L, t in zip(layers, ms[1:1 + len(layers)]):
        t /= cnt
        fl = L['flops_per_slice'] * n
        tot_ms += t
        tot_fl += fl
        issued = L['mfma_flops_per_slice'] * n
        tot_is += issued
        print('{:<52} {:>4}->{:<4} {:>3}x{:<3} {:>7} {:<34} {:>8.3f} {:>7.1f} {:>6.1f} {:>6.1f}'.format(
            L['name'][:52], L['cin'], L['cout'], L['height'], L['width'], '{}x{}'.format(L['grid_height'], L['grid_width']), L['kernel'] + ('+head' if L['head_fusable'] and model.fuse_head else ''), t, fl / t / 1e9, fl / t / 1e9 / 1.573,
            issued / t / 1e9 / 1.573))
    print('input re-layout {:.3f} ms, head {:.3f} ms'.format(ms[0] / cnt, ms[-1] / cnt))
    print('conv total {:.3f} ms  {:.1f} TF/s algorithmic ({:.1f}% of 157.3); MFMA pipe issue {:.1f}%'.format(
        tot_ms, tot_fl / tot_ms / 1e9, tot_fl / tot_ms / 1e9 / 1.573, tot_is / tot_ms / 1e9 / 1.573))


if __name__ == '__main__':
    main()
