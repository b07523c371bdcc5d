#!/usr/bin/env python3
"""One-off sweep (not collected by pytest; run by hand on a GPU box: `python tests/fuzz_plans.py [cases] [seed]`): random architectures x image
sizes x batch sizes through whatever plan the planner makes -- padded levels, real extents, direct kernels, centre pads -- against the oracle.
Prints every failing case; exit code 1 if any."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import unet_oracle as uo  # noqa: E402
from rcu_amd.model import UNet  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    g = torch.Generator().manual_seed(5)
    dev = torch.device('cuda')
    bad, kernels = [], set()
    for case in range(cases):
        depth = int(rng.choice([2, 3, 4, 4, 4]))
        params = dict(nb_classes=int(rng.choice([2, 2, 2, 3])), in_channels=int(rng.choice([1, 3, 4, 5])), depth=depth,
                      start_filters=int(rng.choice([4, 8, 16, 32])), dropout=float(rng.choice([0.05, 0.3])),
                      residual=bool(rng.rand() < 0.15), sigma_out=bool(rng.rand() < 0.2))
        if rng.rand() < 0.2:
            params['dropout_center'] = depth
        step = 1 if rng.rand() < 0.25 else (1 << depth)            # a quarter of the cases: sizes 2^depth does not divide (centre pads)
        lo = 1 << depth
        h = max(lo, step * int(rng.randint(1, 300 // step + 1)))
        w = max(lo, step * int(rng.randint(1, 300 // step + 1)))
        n = int(rng.randint(1, 10))
        st = uo.synthetic_state(100 + case, **params)
        m = UNet(**params)
        m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
        m = m.to(dev)
        x = torch.randn(n, params['in_channels'], h, w, generator=g)
        _, sites = uo.unet_plan(**params)
        masks = uo.sample_masks(sites, n, 0.3, g)
        rows = m.layer_table(h, w, n)
        kernels.update(r['kernel'] for r in rows)
        for mk in (None, masks):
            ref = uo.unet_forward(st, x, mk, **params)
            out = m(x.to(dev), mk)
            refs = ref if isinstance(ref, tuple) else (ref,)
            outs = out if isinstance(out, tuple) else (out,)
            scale = max(1.0, max(float(r.abs().max()) for r in refs))
            err = max(float((o.cpu() - r).abs().max()) for o, r in zip(outs, refs))
            if not err < 3e-6 * scale:
                bad.append((case, params, n, h, w, mk is not None, err))
                print('FAIL', bad[-1], flush=True)
        del m
    print('{} cases, {} failures, {} distinct kernels seen'.format(cases, len(bad), len(kernels)))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
