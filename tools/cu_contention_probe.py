#!/usr/bin/env python3
"""What a kernel that takes whole CUs for a while costs the persistent conv kernels (a stand-in, on ONE GPU, for a collective's kernel running beside
them on an 8-GPU node: RCCL's workgroups cannot share a CU with a conv workgroup, which leaves no registers or LDS).  Experiment build only:

    make -C reliability-challenges-uncertainty_amd/csrc BUILD=_build_exp OUT=../librcu_hip_exp.so EXTRA=-DRCU_EXPERIMENTS
    RCU_HIP_LIBRARY=$PWD/reliability-challenges-uncertainty_amd/librcu_hip_exp.so python tools/cu_contention_probe.py [steps, default 12]

Per volume (the bench's step: ws pass + T = 20 passes, two lanes, resident input) a `hog` of W workgroups x U microseconds is launched on a side stream behind
the volume's last launch -- where the asynchronous reduce of the multi-GPU runner sits.  Prints ms per volume for a few (W, U)."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402

import bench  # noqa: E402
from rcu_amd import _lib, steps  # noqa: E402
from rcu_amd import distributed as rdist  # noqa: E402


def main():
    n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device('cuda')
    lib = _lib.load()
    lib.rcu_debug_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    model = bench.make_model(20, dev)
    x = bench.make_volume(20)[0].to(dev)
    group = steps.pass_group_size(model, x.shape[0], bench.HEIGHT, bench.WIDTH, steps.McPredictStep.GROUP_PIXELS)
    runner = rdist.ShardedMcRunner(model, 20, seed=20, pass_group=group, lanes=2)
    side = torch.cuda.Stream(device=dev)
    out = {}
    for wgs, usec in ((0, 0), (8, 1000), (8, 3000), (32, 1000), (32, 3000), (128, 1000), (0, 0)):
        for k in range(3):
            runner.step_async(x, k).result()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = []
        for k in range(10, 10 + n_steps):
            pend.append(runner.step_async(x, k))
            if wgs:
                side.wait_stream(torch.cuda.current_stream())      # behind the volume's launches, as the reduce is
                lib.rcu_debug_hog(wgs, usec, ctypes.c_void_p(side.cuda_stream))
        for p in pend:
            p.result()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n_steps * 1e3
        out['{} workgroups x {} us'.format(wgs, usec) + (' (again)' if (wgs, usec) == (0, 0) and out else '')] = round(ms, 2)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
