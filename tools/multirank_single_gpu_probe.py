#!/usr/bin/env python3
"""Exercise ShardedMcRunner.step_async with 2 ranks that share the ONE GPU of a gpurun box (no 8-GPU node is
available to development runs).  Tries RCCL first (it may refuse two ranks on one device), then gloo with device
tensors.  Compares the pipelined multi-rank result with a single-rank run on the same injected masks.

    timeout 300 python tools/multirank_single_gpu_probe.py [nccl|gloo]
"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

T = 5


def _model_and_input(dev):
    import bench
    from rcu_amd import steps
    model = bench.make_model(20, dev)
    x = bench.make_volume(20)[0][:32].to(dev)
    g = torch.Generator(device=dev).manual_seed(11)
    steps.set_dropout_mode(model, True)
    mask_sets = [model.sample_masks(32, dev, generator=g) for _ in range(T)]    # same on both ranks (same seed)
    steps.set_dropout_mode(model, False)
    return model, x, mask_sets


def worker(rank, world, port, backend, q):
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)     # a hung collective shows where instead of timing the caller out
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from rcu_amd.distributed import ShardedMcRunner
    model, x, mask_sets = _model_and_input(dev)
    runner = ShardedMcRunner(model, T, ws_pass=True, rank=rank, world=world)
    pend = [runner.step_async(x, k, mask_sets) for k in range(4)]
    outs = [p.result() for p in pend]
    runner.drain()
    torch.cuda.synchronize()
    if rank == 0:
        single = ShardedMcRunner(model, T, ws_pass=True, rank=0, world=1)
        ref = single.step(x, 0, mask_sets)
        worst = 0.0
        for o in outs:
            for k in ('probabilities', 'entropy', 'ws_probabilities'):
                worst = max(worst, float((o[k] - ref[k]).abs().max()))
        q.put(worst)
    dist.barrier()
    dist.destroy_process_group()


def main():
    backend = sys.argv[1] if len(sys.argv) > 1 else 'nccl'
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    mp.spawn(worker, args=(2, port, backend, q), nprocs=2, join=True)
    print('backend {}: max |pipelined 2-rank - single rank| = {:.3e}'.format(backend, q.get()))


if __name__ == '__main__':
    main()
