#!/usr/bin/env python3
"""Why does the bench step cost 4 % more with a one-rank RCCL process group AND the prefetched host-to-device copy (profiles/r04_bench_force_pg.json:
166.9 against 172.7 MC-sample-volumes/s; the same steps on a resident volume: 173.4)?  Times the bench's step loop (VolumePrefetcher feeding
ShardedMcRunner) in a few configurations of one process:

    python tools/pg_h2d_probe.py [steps, default 12]

  plain          no process group
  pg-idle        process group initialised, steps do not touch it
  pg-exchange    every volume through _exchange (step_async: asynchronous reduce, finalize on a side stream)
  pg-sync        every volume through _exchange, synchronous form (step)
each with the copy inside and with the volume resident."""
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from rcu_amd import distributed as rdist  # noqa: E402


def loop(runner, x, feeder, steps, first, use_async, with_copy):
    def one(k, xin):
        return runner.step_async(xin, k) if use_async else rdist.PendingSummary(runner.step(xin, k))
    for k in range(first - 3, first):
        one(k, x).result()
    runner.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pend = []
    if with_copy:
        feeder.issue(first)
    for k in range(first, first + steps):
        xin = feeder.get(k) if with_copy else x
        if with_copy and k + 1 < first + steps:
            feeder.issue(k + 1)
        pend.append(one(k, xin))
        if with_copy:
            feeder.done(k)
    for p in pend:
        p.result()
    runner.drain()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    model = bench.make_model(20, dev)
    x_cpu = bench.make_volume(20)[0]
    x = x_cpu.to(dev)
    feeder = bench.VolumePrefetcher(x_cpu, dev)
    kw = dict(seed=20, pass_group=4, lanes=2)
    out = {}

    def run(tag, **extra):
        r = rdist.ShardedMcRunner(model, 20, **kw, **{k: v for k, v in extra.items() if k == 'force_exchange'})
        for with_copy in (True, False):
            out['{}/{}'.format(tag, 'copy' if with_copy else 'resident')] = loop(r, x, feeder, steps, 100, extra.get('use_async', True), with_copy)

    run('plain')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    t = torch.ones(4, device=dev)
    dist.all_reduce(t)          # the communicator exists from here on
    torch.cuda.synchronize()
    run('pg-idle')
    run('pg-exchange', force_exchange=True)
    run('pg-sync', force_exchange=True, use_async=False)
    print(json.dumps({k: round(v, 3) for k, v in out.items()}))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
