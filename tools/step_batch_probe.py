#!/usr/bin/env python3
"""GPU time of the MC predict step + summary by batch size (the drop-in scripts' `batch_size`): T = 20 on the BraTS slice size, steps enqueued
back to back, one synchronisation at the end.  What five batches of 32 slices cost against one of 160:
    python tools/step_batch_probe.py [T, default 20] [only=<batch size>] [reps=<volumes>] [option=value ... for McPredictStep: lanes=1, group_pixels=0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from rcu_amd import steps
    T = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
    options = {k: int(v) for k, v in (a.split('=') for a in sys.argv[1:] if '=' in a)}
    only, reps = options.pop('only', None), options.pop('reps', 6)
    dev = torch.device('cuda')
    model = bench.make_model(20, dev)
    x = bench.make_volume(20)[0].to(dev)
    ctx = steps.TorchTestContext('cuda', model)
    summary = steps.MultiPredictionSummary()
    for n in ((only,) if only else (160, 32, 16, 8)):
        step = steps.McPredictStep(T, seed=20, **options)
        batches = [x[i:i + n] for i in range(0, x.shape[0], n)]

        def volume(first_index):
            outs = []
            for b, xb in enumerate(batches):
                bc = steps.BatchContext({'images': xb}, first_index + b)
                step(bc, None, ctx)
                summary(bc, None, ctx)
                outs.append(bc)
            return outs
        volume(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        keep = [volume(1000 + r * len(batches)) for r in range(reps)]
        for outs in keep:
            for bc in outs:
                steps.wait_for_outputs(bc)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print('batch {:>3} slices x {:>2} batches per volume: {:.1f} ms per volume = {:.1f} MC-sample-volumes/s'.format(n, len(batches), dt * 1e3, T / dt))


if __name__ == '__main__':
    main()
