#!/usr/bin/env python3
"""BASELINE.json configs[1]: ISIC baseline_mc, 3 x 256 x 256 images, T = 20 MC-dropout passes (+ weight-scaling pass)
through McPredictStep + MultiPredictionSummary on one MI355X.  python tools/isic_bench.py [batch] [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rcu_amd import model as model_mod  # noqa: E402
from rcu_amd import steps  # noqa: E402

PARAMS = dict(nb_classes=2, in_channels=3, depth=4, start_filters=32, dropout=0.05)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device('cuda')
    torch.manual_seed(20)
    model = model_mod.UNet(**PARAMS)        # torch's default init; BatchNorm statistics randomised as in bench.py
    gen = torch.Generator().manual_seed(1020)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
    model.weights_changed()
    model = model.to(dev)
    x = torch.rand(batch, 3, 256, 256, generator=torch.Generator().manual_seed(20)).to(dev)
    ctx = steps.TorchTestContext('cuda', model)
    chain = [steps.McPredictStep(20), steps.MultiPredictionSummary()]

    def one():
        bc = steps.BatchContext({'images': x}, 0)
        for s in chain:
            s(bc, None, ctx)
        return bc.output

    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        out = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_steps
    flops = sum(L['flops_per_slice'] for L in model.layer_table(256, 256, batch)) * batch * 21
    print(json.dumps({'workload': 'ISIC baseline_mc 3x256x256, batch {}, T=20 + ws pass'.format(batch),
                      'ms_per_batch': dt * 1e3, 'mc_sample_images_per_s': 20 * batch / dt,
                      'tflops_canonical': flops / dt / 1e12, 'entropy_mean': float(out['entropy'].mean())}))


if __name__ == '__main__':
    main()
