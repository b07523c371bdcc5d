#!/usr/bin/env python3
"""The kernels of rcu_conv.hip -- the direct (implicit-GEMM) family that carries what the Winograd kernels cannot: the centre-padded levels of
image sizes not divisible by 2^depth (common/model/unet.py:110-116), units that add to their output (ConvResidualBlock), and every level under
rcu_unet_options.pad_levels = 0 -- timed per layer with HIP events (the family had no evidence since round 1):
    python tools/fallback_profile.py [forwards]
Two forwards: 155 slices of 4 x 240 x 240 with pad_levels = 0 (the plans of rounds 1-5 on the reference's real BraTS shape) and 160 slices of
4 x 100 x 100 (levels 100 / 50 / 25 / 12 / 6: centre pads at two levels).  Under `rocprofv3 --kernel-trace --stats -- python3 tools/fallback_profile.py`
the same launches give the kernel-stats table; with --pmc SQ counters their executed fraction of the matrix peak."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def report(model, n, h, w, reps):
    x = torch.randn(n, 4, h, w, generator=torch.Generator().manual_seed(20)).cuda()
    for _ in range(2):
        model(x)
    model.profile_begin(h, w, n, reps)
    for _ in range(reps):
        model(x)
    torch.cuda.synchronize()
    cnt, ms = model.profile_collect(h, w, n)
    layers = model.layer_table(h, w, n)
    print('{} slices of 4 x {} x {}, plan options {}'.format(n, h, w, model.plan_options or 'default'))
    print('{:<50} {:>4}->{:<4} {:>7} {:>7} {:<40} {:>8} {:>7} {:>7}'.format('layer', 'cin', 'cout', 'HxW', 'grid', 'kernel', 'ms', 'TF/s', 'exec %'))
    per = {}
    for L, t in zip(layers, ms[1:1 + len(layers)]):
        t /= cnt
        e = per.setdefault(L['kernel'], [0.0, 0.0, 0.0])
        e[0] += t
        e[1] += L['flops_per_slice'] * n
        e[2] += L['mfma_flops_per_slice'] * n
        print('{:<50} {:>4}->{:<4} {:>7} {:>7} {:<40} {:>8.3f} {:>7.1f} {:>7.1f}'.format(
            L['name'][:50], L['cin'], L['cout'], '{}x{}'.format(L['height'], L['width']), '{}x{}'.format(L['grid_height'], L['grid_width']),
            L['kernel'], t, L['flops_per_slice'] * n / t / 1e9, L['mfma_flops_per_slice'] * n / t / 1e9 / 1.573))
    print('per kernel: ' + '; '.join('{} {:.3f} ms, executed {:.1f} % of the fp32 matrix peak, canonical {:.1f} TF/s'.format(
        k, v[0], v[2] / v[0] / 1e9 / 1.573, v[1] / v[0] / 1e9) for k, v in sorted(per.items())))
    tot = sum(v[0] for v in per.values())
    print('conv total {:.3f} ms, {:.1f} TF/s canonical; head {:.3f} ms\n'.format(tot, sum(v[1] for v in per.values()) / tot / 1e9, ms[-1] / cnt))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device('cuda')
    direct = bench.make_model(20, dev)
    direct.plan_options = dict(pad_levels=0)
    report(direct, 155, 240, 240, reps)
    report(bench.make_model(20, dev), 160, 100, 100, reps)


if __name__ == '__main__':
    main()
