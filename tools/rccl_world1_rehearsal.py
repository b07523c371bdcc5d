#!/usr/bin/env python3
"""Rehearsal of the N > 1 exchange path on a box with ONE GPU: an RCCL ("nccl") process group of world size 1, every volume routed
through ShardedMcRunner._exchange (rcu_amd/distributed.py) -- the asynchronous sum-reduce on RCCL's stream, the root's finalize on a
side stream that waits for the work handle, record_stream on the reduce buffer, at most `depth` reduces in flight, drain() -- for
both transports of the weight-scaling probabilities ('reduce': in the tail of the one buffer; 'p2p': a send / recv from their owner,
which is a no-op when the root owns them, as it always does at world size 1: asserted).

What a one-rank group can and cannot show.  A sum-reduce over one rank leaves the buffer as it is -- RCCL returns without launching a
kernel for an in-place one-rank collective -- so the outputs must carry the BITS of the plain world-1 step (asserted), and the cost of
the path is host-side: work handles, stream waits, the side stream.  The one device kernel RCCL launches at one rank is
`oneRankReduce<FuncPreMulSum<...>>` (a scaled copy: what ReduceOp.AVG compiles to); the probe below runs it once on the reduce buffer's
size, so that a `rocprofv3 --kernel-trace` of this script shows RCCL device code loading and running on the box.  Bandwidth, link
topology and CU contention with the persistent conv kernels need the 8-GPU node.

    python tools/rccl_world1_rehearsal.py [slices, default 160] [T, default 20] [steps, default 6] [lazy | eager, default lazy]

`lazy` / `eager`: init_process_group without / with `device_id=` (the communicator created at the first collective / at once).  The tool also
times the steps with the bench's prefetched host-to-device copy of every volume (`*_with_copy`): with the eagerly created communicator a step that
overlaps such a copy runs ~4 ms longer on this image (profiles/r04_pg_h2d.txt), which is why bench.py initialises lazily.
"""
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
from rcu_amd import steps as rsteps  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.SLICES
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    eager = len(sys.argv) > 4 and sys.argv[4] == 'eager'
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if 'MASTER_PORT' not in os.environ:
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
    t0 = time.perf_counter()
    dist.init_process_group('nccl', rank=0, world_size=1, **(dict(device_id=dev) if eager else {}))      # "nccl" is RCCL on ROCm
    init_s = time.perf_counter() - t0
    model = bench.make_model(20, dev)
    x_cpu = bench.make_volume(20, n)[0]
    x = x_cpu.to(dev)
    feeder = bench.VolumePrefetcher(x_cpu, dev)
    group = rsteps.pass_group_size(model, n, bench.HEIGHT, bench.WIDTH, rsteps.McPredictStep.GROUP_PIXELS)
    kw = dict(seed=20, pass_group=group, lanes=2)
    plain = rdist.ShardedMcRunner(model, T, **kw)
    record = dict(backend=dist.get_backend(), world=dist.get_world_size(), init='eager (device_id=)' if eager else 'lazy', slices=n, T=T, steps=steps,
                  init_process_group_s=init_s,
                  rccl=torch.cuda.nccl.version() if hasattr(torch.cuda, 'nccl') else None)

    def timed(runner, first, with_copy=False):
        for k in range(first - 2, first):
            runner.step_async(x, k).result()
        runner.drain()
        torch.cuda.synchronize()
        dist.barrier(device_ids=[0])
        torch.cuda.synchronize()
        ts = time.perf_counter()
        pend = []
        if with_copy:
            feeder.issue(first)
        for k in range(first, first + steps):
            xin = feeder.get(k) if with_copy else x
            if with_copy and k + 1 < first + steps:
                feeder.issue(k + 1)
            pend.append(runner.step_async(xin, k))
            if with_copy:
                feeder.done(k)
        outs = [p.result() for p in pend]
        runner.drain()
        torch.cuda.synchronize()
        dist.barrier(device_ids=[0])
        torch.cuda.synchronize()
        return (time.perf_counter() - ts) / steps * 1e3, outs

    ms_plain, outs_plain = timed(plain, 10)
    record['plain_ms_per_volume'] = ms_plain
    record['plain_ms_per_volume_with_copy'] = timed(plain, 10, with_copy=True)[0]
    equal = True
    for transport in ('reduce', 'p2p'):
        r = rdist.ShardedMcRunner(model, T, ws_transport=transport, force_exchange=True, **kw)
        t1 = time.perf_counter()
        first = r.step(x, 9)                      # synchronous form; the first collective creates the communicator
        torch.cuda.synchronize()
        first_s = time.perf_counter() - t1
        ref = plain.step(x, 9)
        same_sync = all(torch.equal(first[k], ref[k]) for k in ref) and set(first) == set(ref)
        ms, outs = timed(r, 10)
        same_async = all(torch.equal(a[k], b[k]) for a, b in zip(outs, outs_plain) for k in b)
        assert r.p2p_messages == 0, 'at world size 1 the root owns every weight-scaling pass: no send / recv'
        assert r.ws_owner(3) == r.root
        equal = equal and same_sync and same_async
        record[transport] = dict(ms_per_volume=ms, over_plain=ms / ms_plain, ms_per_volume_with_copy=timed(r, 10, with_copy=True)[0], first_step_s=first_s, bits_equal_sync=same_sync,
                                 bits_equal_async=same_async, p2p_messages=r.p2p_messages, inflight_after_drain=len(r._inflight))
    # the device-side probe: RCCL's one-rank kernel on a buffer of the reduce's size (statistics + ws tail, float32)
    flat = torch.ones(2 * 2 * n * bench.HEIGHT * bench.WIDTH, device=dev)
    w = dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    record['avg_probe_ok'] = bool(torch.all(flat == 1.0).item())
    tmax = torch.tensor([ms_plain], device=dev, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    record['bits_equal'] = bool(equal)
    print(json.dumps(record))
    dist.destroy_process_group()
    if not (equal and record['avg_probe_ok']):
        raise SystemExit(1)


if __name__ == '__main__':
    main()
