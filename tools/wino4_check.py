#!/usr/bin/env python3
"""F(4x4,3x3) kernels (csrc/rcu_wino4.hip) on the GPU: parity against the oracle and against the F(2x2,3x3) build of the same plan
(plan option conv_winograd4=0), then per-layer timing of both on the 160-slice BraTS volume.   python tools/wino4_check.py [slices] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402
from rcu_amd import steps  # noqa: E402
from rcu_amd.model import UNet  # noqa: E402


def build(state, dev, wino4, shape=None):
    m = UNet(**bench.MODEL_PARAMS)
    m.load_state_dict(state)
    m.plan_options = dict(conv_winograd4=1 if wino4 else 0)      # rcu_unet_options (include/rcu.h)
    m = m.to(dev)
    if shape is not None:
        m.layer_table(*shape)
    return m


def main():
    n_big = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device('cuda')
    st = uo.synthetic_state(21, **bench.MODEL_PARAMS)
    g = torch.Generator().manual_seed(6)
    n, h, w = 4, 192, 128
    x = torch.randn(n, 4, h, w, generator=g)
    _, sites = uo.unet_plan(**bench.MODEL_PARAMS)
    masks = uo.sample_masks(sites, n, 0.3, g)
    m4, m2 = build(st, dev, True, (h, w, n)), build(st, dev, False, (h, w, n))
    print('kernels F(4,3) build:', sorted({r['kernel'] for r in m4.layer_table(h, w, n)}))
    for var, mk in [(v, mk) for v in os.environ.get('RCU_W4_PARITY_VARIANTS', '0').split(',') for mk in (None, masks)]:
        os.environ['RCU_W4_VARIANT'] = var
        ref = uo.unet_forward(st, x, mk, **bench.MODEL_PARAMS)
        o4 = m4(x.to(dev), mk).cpu()
        o2 = m2(x.to(dev), mk).cpu()
        print('variant', var, 'masks' if mk else 'eval ', 'max|F43 - oracle| {:.3e}   max|F23 - oracle| {:.3e}   max|F43 - F23| {:.3e}   |logit|max {:.3f}'.format(
            float((o4 - ref).abs().max()), float((o2 - ref).abs().max()), float((o4 - o2).abs().max()), float(ref.abs().max())))
    torch.cuda.synchronize()
    del m4, m2
    xb = bench.make_volume(20)[0][:n_big].to(dev)
    stb = {k: v.detach().cpu() for k, v in bench.make_model(20, 'cpu').state_dict().items()}
    variants = [(True, v) for v in ([int(t) for t in os.environ['RCU_W4_ABLATE'].split(',')] if os.environ.get('RCU_W4_ABLATE') else (0,))] + [(False, 0)]
    for wino4, var in variants:
        os.environ['RCU_W4_VARIANT'] = str(var)
        m = build(stb, dev, wino4, (h, w, n_big))
        steps.set_dropout_mode(m, True)
        for _ in range(2):
            m(xb)
        m.profile_begin(h, w, n_big, reps)
        for _ in range(reps):
            m(xb)
        torch.cuda.synchronize()
        cnt, ms = m.profile_collect(h, w, n_big)
        layers = m.layer_table(h, w, n_big)
        print('--- conv_winograd4={} variant {} (bit 0: no LDS-DMA in the chunks, 1: no epilogue, 2: no input transform, 3: no chunk barrier)'.format(int(wino4), var))
        tot = 0.0
        for L, t in zip(layers, ms[1:1 + len(layers)]):
            t /= cnt
            tot += t
            if var and 'winograd4' not in L['kernel']:
                continue
            ex = L['mfma_flops_per_slice'] * n_big
            print('{:<44} {:>4}->{:<4} {:>3}x{:<3} {:<36} {:>7.3f} ms  pipe {:>5.1f}%  canonical {:>6.1f} TF/s'.format(
                L['name'][:44], L['cin'], L['cout'], L['height'], L['width'], L['kernel'], t, ex / t / 1e9 / 1.573,
                L['flops_per_slice'] * n_big / t / 1e9))
        print('conv total {:.3f} ms'.format(tot))
        del m


if __name__ == '__main__':
    main()
