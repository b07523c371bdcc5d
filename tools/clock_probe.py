#!/usr/bin/env python3
"""Is the conv stack clock/power limited?  Runs the 160-slice BraTS forward in a loop for a few seconds with (a) the benchmark's data and
(b) an all-zero volume and all-zero weights (same instructions, no toggling operands) while sampling `rocm-smi` (sclk, power), and
prints the per-forward time of both.    python tools/clock_probe.py [seconds]"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from rcu_amd import steps  # noqa: E402


def sample(stop, rows):
    while not stop.is_set():
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=10).stdout
            keep = [ln.strip() for ln in out.splitlines() if 'sclk' in ln or 'Power' in ln or 'mclk' in ln]
            rows.append(' | '.join(keep))
        except Exception as e:  # noqa: BLE001
            rows.append('rocm-smi failed: {}'.format(e))
        time.sleep(0.5)


def run(label, model, x, seconds):
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    stop, rows = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, rows))
    th.start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            model(x)
        torch.cuda.synchronize()
        n += 20
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    print('{:<28} {:.3f} ms per forward ({} forwards)'.format(label, dt / n * 1e3, n))
    for r in rows[1:6]:
        print('    ', r)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    dev = torch.device('cuda')
    model = bench.make_model(20, 'cpu').to(dev)
    steps.set_dropout_mode(model, True)
    x = bench.make_volume(20)[0].to(dev)
    run('benchmark data', model, x, seconds)
    run('zero volume', model, torch.zeros_like(x), seconds)
    with torch.no_grad():
        for p in model.parameters():
            p.zero_()
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.zero_()
    model.weights_changed()
    model = model.to(dev)
    run('zero volume, zero weights', model, torch.zeros_like(x), seconds)


if __name__ == '__main__':
    main()
