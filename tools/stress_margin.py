#!/usr/bin/env python3
"""Measured margins of the numerics stress case (golden G18, tests/test_gpu_parity.py::test_unet_stress_golden_wide_activations_and_logits): max |dlogit| of every
kernel-family plan against the oracle, relative to the logit range, and max |dp|.   python tools/stress_margin.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

from conftest import golden_params, load_golden  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402
from rcu_amd.model import UNet  # noqa: E402


def main():
    g = load_golden('g18_unet_stress')
    p = golden_params(g)
    st = uo.stress_state(uo.reference_init_state(int(g['seed']), bn_seed=int(g['seed']) + 1000, **p), float(g['bn_gain']), float(g['head_gain']))
    x = torch.as_tensor(g['x'])
    _, sites = uo.unet_plan(**p)
    dev = torch.device('cuda')
    for tag, opts in (('F(4x4,3x3) where it fits (shipped)', {}), ('F(2x2,3x3) only', dict(conv_winograd4=0)), ('direct kernels', dict(conv_winograd=0))):
        m = UNet(**p)
        m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
        m.plan_options = opts
        m = m.to(dev)
        rows = []
        for t in (None, 0, 1, 2):
            mk = None if t is None else [g['mask{}_{}'.format(t, s)] for s in range(len(sites))]
            ref = uo.unet_forward(st, x, mk, **p)
            out = m(x.to(dev), mk).cpu()
            scale = float(ref.abs().max())
            rows.append((scale, float((out - ref).abs().max()) / scale, float((torch.softmax(out, 1) - torch.softmax(ref, 1)).abs().max())))
        print('{:<36}'.format(tag) + '  '.join('|logit| {:5.1f}: rel {:.2e}, dp {:.2e}'.format(*r) for r in rows))


if __name__ == '__main__':
    main()
