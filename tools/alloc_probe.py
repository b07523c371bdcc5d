#!/usr/bin/env python3
"""Does the caching allocator keep asking the driver for memory in the steady state of the MC runner?  Reserved / allocated bytes and the
allocator's hipMalloc / hipFree counts every ten volumes.    python tools/alloc_probe.py [volumes, default 60] [lanes, default 2]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from rcu_amd import distributed as rdist
    volumes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device('cuda')
    model = bench.make_model(20, dev)
    x = bench.make_volume(20)[0].to(dev)
    runner = rdist.ShardedMcRunner(model, 20, ws_pass=True, seed=20, pass_group=4, lanes=lanes)
    pending = []
    for k in range(volumes):
        pending.append(runner.step_async(x, k))
        if len(pending) > 2:
            pending.pop(0).result()
        if k % 10 == 9:
            torch.cuda.synchronize()
            st = torch.cuda.memory_stats(dev)
            print('volume {:>4}: reserved {:8.1f} MB  allocated {:8.1f} MB  hipMalloc {:>5}  hipFree {:>5}  inactive split {:8.1f} MB'.format(
                k + 1, st['reserved_bytes.all.current'] / 1e6, st['allocated_bytes.all.current'] / 1e6, st.get('num_device_alloc', 0),
                st.get('num_device_free', 0), st.get('inactive_split_bytes.all.current', 0) / 1e6))
    for p in pending:
        p.result()


if __name__ == '__main__':
    main()
