#!/usr/bin/env python3
"""Diagnostics of the headline step that used to sit inside bench.py's timed loop behind RCU_BENCH_* switches (round 4): the benchmark now has
ONE path through its timed region, and these variants of the same loop live here (one GPU, one process).

    python tools/bench_diagnostics.py [--steps 10] [--warmup 3] [--mode timed | unused | nowait | resident] [--step-times]

  timed     the bench's loop: every volume prefetched from pinned host memory on a copy stream, the step waits for its event
  unused    the copies run, the steps read a resident volume instead (what the copy costs the step when nothing waits for it)
  nowait    the steps read the feeder's buffers without waiting for the copy's event (the event wait's cost; results are garbage)
  resident  no copies at all
--step-times prints when the host had enqueued each step (the host is never the bottleneck: it runs ahead of the GPU)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mode', choices=('timed', 'unused', 'nowait', 'resident'), default='timed')
    ap.add_argument('--step-times', action='store_true')
    ap.add_argument('--lanes', type=int, default=2)
    args = ap.parse_args()
    from rcu_amd import distributed as rdist
    from rcu_amd import steps
    dev = torch.device('cuda', 0)
    model = bench.make_model(20, dev)
    x_cpu = bench.make_volume(20)[0]
    x = x_cpu.to(dev)
    feeder = bench.VolumePrefetcher(x_cpu, dev)
    group = steps.pass_group_size(model, bench.SLICES, bench.HEIGHT, bench.WIDTH, steps.McPredictStep.GROUP_PIXELS)
    runner = rdist.ShardedMcRunner(model, 20, seed=20, pass_group=group, lanes=args.lanes)
    for k in range(args.warmup):
        runner.step(x, k)
    torch.cuda.synchronize()
    first, end = args.warmup, args.warmup + args.steps
    host_times = []
    t0 = time.perf_counter()
    if args.mode != 'resident':
        feeder.issue(first)
    for k in range(first, end):
        if args.mode == 'resident':
            xin = x
        else:
            xin = feeder.bufs[k % len(feeder.bufs)] if args.mode == 'nowait' else feeder.get(k)
            if k + 1 < end:
                feeder.issue(k + 1)
        runner.step(x if args.mode == 'unused' else xin, k)
        if args.mode != 'resident':
            feeder.done(k)
        host_times.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec = dict(mode=args.mode, steps=args.steps, ms_per_step=dt * 1e3 / args.steps, mc_sample_volumes_per_s=20 * args.steps / dt)
    if args.step_times:
        rec['host_enqueued_at_ms'] = [round(v * 1e3, 1) for v in host_times]
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
