#!/usr/bin/env python3
"""Do MC passes on S HIP streams (one model replica + statistics blob each) overlap their kernel tails?
    python tools/stream_overlap_probe.py [passes per launch: 1 (default) or 2]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from rcu_amd import steps  # noqa: E402


def main():
    dev = torch.device('cuda')
    x = bench.make_volume(20)[0].to(dev)
    T = 20
    group = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for S in (1, 2, 3):
        models = [bench.make_model(20, dev) for _ in range(S)]
        streams = [torch.cuda.Stream() for _ in range(S)]
        for m in models:
            steps.set_dropout_mode(m, True)
        stats = [steps.McStatistics(160, 2, 192, 128, dev) for _ in range(S)]
        torch.cuda.synchronize()
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(T // group):
                s = t % S
                with torch.cuda.stream(streams[s]):
                    models[s].forward_accumulate(x, stats[s], passes=group)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print('streams {}, {} pass(es) per launch: {:.1f} ms for {} passes = {:.2f} ms/pass'.format(S, group, dt * 1e3, T, dt * 1e3 / T))
        del models, stats


if __name__ == '__main__':
    main()
