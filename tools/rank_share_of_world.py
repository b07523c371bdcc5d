#!/usr/bin/env python3
"""The compute side of an N-GPU run, MEASURED on one GPU: exactly the launches rank r of a world of N makes for a stream of volumes
(rcu_amd.distributed.ShardedMcRunner's job rotation, pass groups, stream lanes, canonical plans -- everything but the collective), timed,
next to the same volumes on a world of one.  T + 1 = 21 jobs per volume leave a rank of 8 two or three forward passes per volume: launches of
320-480 samples where one GPU alone runs 640 -- the deep U-Net levels no longer fill their last round of workgroups.  ``--volumes-per-step v``
hands the runner v consecutive volumes as ONE batch of 160 v slices (what `others.coalesce_pixels` does in the scripts and
`bench.py --volumes-per-step` in the benchmark): the rank's passes then run as groups of 4 / v passes x 160 v slices = 640 samples again.

    python tools/rank_share_of_world.py [--world 8] [--rank 0] [--volumes 16] [--volumes-per-step 1 2 4] [--mc 20] [--out FILE.json]

compute-side efficiency of the world = (ms per forward pass, one GPU alone) / (ms per forward pass of the rank's share)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def run(model, x1, world, rank, T, volumes, v, lanes):
    from rcu_amd import distributed as rdist
    from rcu_amd import steps
    x = x1 if v == 1 else torch.cat([x1] * v)
    n, _, h, w = x.shape
    group = steps.pass_group_size(model, n, h, w, steps.McPredictStep.GROUP_PIXELS)
    runner = rdist.ShardedMcRunner(model, T, ws_pass=True, rank=rank, world=world, seed=20, pass_group=group, lanes=lanes)
    n_steps = volumes // v
    for k in range(2):                                  # warm: plans, workspaces, allocator
        runner._run_jobs(x, k, None)
    torch.cuda.synchronize()
    runner.forwards_run = 0
    t0 = time.perf_counter()
    for k in range(2, 2 + n_steps):
        runner._run_jobs(x, k, None)                    # this rank's jobs of the step; no exchange (one process)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    forwards = runner.forwards_run * v                  # forward passes in units of one 160-slice volume
    launches = sorted({n * g for k in range(2, 2 + n_steps)
                       for g in steps.balanced_groups(sum(1 for j in runner.jobs_of(k, rank) if j != 0), group, lanes)})
    return dict(volumes_per_step=v, slices_per_step=n, pass_group=group, steps=n_steps, volumes=n_steps * v, forward_volumes=forwards,
                elapsed_s=dt, ms_per_forward_volume=dt * 1e3 / max(forwards, 1), mc_launch_samples=launches,
                ms_per_volume_of_the_world=dt * 1e3 / (n_steps * v))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--rank', type=int, default=0)
    ap.add_argument('--volumes', type=int, default=16)
    ap.add_argument('--volumes-per-step', type=int, nargs='+', default=[1, 2, 4])
    ap.add_argument('--mc', type=int, default=20)
    ap.add_argument('--lanes', type=int, default=2)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    model = bench.make_model(20, dev)
    x1 = bench.make_volume(20)[0].to(dev)
    alone = run(model, x1, 1, 0, args.mc, args.volumes, 1, args.lanes)
    record = dict(world=args.world, rank=args.rank, T=args.mc, lanes=args.lanes, one_gpu_alone=alone, shares=[])
    for v in args.volumes_per_step:
        if args.volumes % v:
            continue
        share = run(model, x1, args.world, args.rank, args.mc, args.volumes, v, args.lanes)
        share['compute_side_efficiency'] = alone['ms_per_forward_volume'] / share['ms_per_forward_volume']
        # every rank has the same load over a rotation of `world` steps: the world finishes a volume in this rank's time per volume
        share['world_mc_sample_volumes_per_s_compute_side'] = args.mc * 1e3 / share['ms_per_volume_of_the_world']
        record['shares'].append(share)
    record['one_gpu_mc_sample_volumes_per_s'] = args.mc * 1e3 / alone['ms_per_volume_of_the_world']
    print(json.dumps(record))
    if args.out:
        with open(args.out, 'w') as f:
            f.write(json.dumps(record, indent=1) + '\n')


if __name__ == '__main__':
    main()
