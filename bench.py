#!/usr/bin/env python3
"""Headline benchmark: MC-sample-volumes/s on a synthetic BraTS-shaped volume (4 x 160 x 192 x 128,
i.e. 160 slices of 4 x 192 x 128), T = 20 MC-dropout passes, on N GPUs of one node.

One step = the reference's hot path for one volume (rechun/dl/customsteps.py:16-71):
    weight-scaling pass + T stochastic U-Net passes (softmax + running statistics fused in) ->
    [N > 1: one RCCL sum-reduce of the per-voxel statistics] -> mean probability + predictive entropy.
Inputs are resident in HBM when the timed region starts.  N > 1 shards the T+1 forward passes of
every step over the ranks (strong scaling); run through
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel (HIP events recorded on the
launch stream inside the timed region, see rcu_unet_profile_begin in include/rcu.h); `cpu_baseline`
is the oracle (a port of the reference's CPU path, pinned against golden vectors) timed on a bounded
sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_HBM_GBS = 8000.0            # same guide: HBM3E peak (about 6.3 TB/s is what a streaming kernel reaches)
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
SLICES, CHANNELS, HEIGHT, WIDTH = 160, 4, 192, 128
MODEL_PARAMS = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05)  # config/train_brats_baseline.yaml:7-12


def make_model(seed, device, sigma_out=False):
    """UNet(2, 4, 4, 32, 0.05) with torch's default init under the seed and randomised BatchNorm statistics
     -- random-init weights of the named architecture; there are no checkpoints offline."""
    from rcu_amd.model import UNet
    torch.manual_seed(seed)
    model = UNet(**MODEL_PARAMS, sigma_out=sigma_out)
    gen = torch.Generator().manual_seed(seed + 1000)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
    model.weights_changed()
    return model.to(device)


def make_volume(seed, n=SLICES):
    """x ~ N(0,1) zeroed outside a centred ellipsoid (= the ECE brain mask); target = smaller ellipsoid."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(n, CHANNELS, HEIGHT, WIDTH, generator=gen)
    zz, yy, xx = torch.meshgrid(torch.linspace(-1, 1, SLICES)[:n] if n <= SLICES else torch.linspace(-1, 1, n),
                                torch.linspace(-1, 1, HEIGHT), torch.linspace(-1, 1, WIDTH), indexing='ij')
    r2 = (zz / 0.9) ** 2 + (yy / 0.85) ** 2 + (xx / 0.8) ** 2
    mask = r2 < 1.0
    target = ((zz / 0.35) ** 2 + ((yy - 0.1) / 0.3) ** 2 + ((xx + 0.1) / 0.3) ** 2 < 1.0).to(torch.uint8)
    x = x * mask[:, None].float()
    return x, mask, target


def cpu_baseline(model, x_cpu, T, seed, budget_s=20.0):
    """The oracle's CPU path (torch-CPU conv stack + torch aggregation) on one batch of 32 slices of the same
    volume (batch_size 32: config/test_brats_baseline_mc.yaml:11).  Bounded: the thread count is the fastest
    of a short probe and the number of stochastic passes is cut so that the sample takes about budget_s."""
    from oracle import summary_oracle as so
    from oracle import unet_oracle as uo
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    n = 32
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    _, sites = uo.unet_plan(**MODEL_PARAMS)
    gen = torch.Generator().manual_seed(seed)
    xs = x_cpu[:n].contiguous()
    fwd = lambda xx, m: uo.unet_forward(state, xx, m, **MODEL_PARAMS)  # noqa: E731
    best = None
    for threads in sorted({min(avail, t) for t in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(threads)
        fwd(xs[:4], None)                       # warm-up (thread pool, primitive caches)
        t0 = time.perf_counter()
        fwd(xs, None)
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (threads, dt)
        if dt > 8.0:
            break
    threads, t_fwd = best
    torch.set_num_threads(threads)
    t_cpu = int(max(1, min(T, budget_s / t_fwd - 1)))
    mask_sets = [uo.sample_masks(sites, n, MODEL_PARAMS['dropout'], gen) for _ in range(t_cpu)]
    t0 = time.perf_counter()
    ws, multi = so.mc_probabilities(fwd, xs, mask_sets)
    out = so.multi_prediction_summary(multi)
    dt = time.perf_counter() - t0
    volumes = t_cpu * n / SLICES            # MC-sample-volume equivalents processed (ws pass timed, not counted)
    return dict(value=volumes / dt, unit='MC-sample-volumes/s', cores=threads, kind='port',
                sample='{} of {} slices x ({} MC passes + ws pass) through oracle/ (torch-CPU, {} of {} host threads) '
                       'in {:.1f} s'.format(n, SLICES, t_cpu, threads, avail, dt)), mask_sets, out


def calibration_kernels(device, volumes=160, reps=5):
    """ECE histogram and uncertainty-error counts over a test-split sized batch (160 BraTS volumes, as
    bin-eval/eval_uncertainty.py processes them), timed with events on the launch stream.  ALGORITHMIC bytes per
    voxel: 4 (confidence / uncertainty f32) + 1 (target) + 1 (mask) for the histogram, + 1 (prediction) for the
    counts."""
    import ctypes
    from rcu_amd import _lib
    lib = _lib.load()
    n = SLICES * HEIGHT * WIDTH
    g = torch.Generator(device=device).manual_seed(5)
    p = torch.rand((volumes, n), device=device, generator=g)
    t = (torch.rand((volumes, n), device=device, generator=g) < p).to(torch.uint8)
    m = (torch.rand((volumes, n), device=device, generator=g) < 0.4).to(torch.uint8)      # brain mask share
    pred = (p > 0.5).to(torch.uint8)
    stream = _lib.current_stream()
    out = {}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    thr = _lib.ece_thresholds(10)
    res = torch.empty(volumes * ctypes.sizeof(_lib.EceResult), device=device, dtype=torch.uint8)
    ws = torch.empty(max(lib.rcu_ece_workspace_bytes(n, volumes), 8), device=device, dtype=torch.uint8)
    ms = timed(lambda: _lib.check(lib.rcu_ece_hist(_lib.ptr(p), _lib.ptr(t), _lib.ptr(m), n, volumes, thr, 10,
                                                   _lib.ptr(res), _lib.ptr(ws), stream)))
    nbytes = volumes * n * 6
    out['ece_hist'] = dict(bound='hbm', ms=ms, achieved=nbytes / ms / 1e6, peak=PEAK_HBM_GBS, unit='GB/s',
                           frac=nbytes / ms / 1e6 / PEAK_HBM_GBS, bytes=nbytes, volumes=volumes)
    ue = (ctypes.c_double * 11)(*[0.05 * k for k in range(1, 11)] + [0.95])
    cnt = torch.empty((volumes, 11, 8), device=device, dtype=torch.int64)
    ws2 = torch.empty(max(lib.rcu_unc_workspace_bytes(n, volumes), 8), device=device, dtype=torch.uint8)
    ms = timed(lambda: _lib.check(lib.rcu_unc_counts(_lib.ptr(p), 0, _lib.ptr(pred), _lib.ptr(t), _lib.ptr(m), n, volumes,
                                                     ue, 11, _lib.ptr(cnt), _lib.ptr(ws2), stream)))
    nbytes = volumes * n * 7
    out['unc_counts'] = dict(bound='hbm', ms=ms, achieved=nbytes / ms / 1e6, peak=PEAK_HBM_GBS, unit='GB/s',
                             frac=nbytes / ms / 1e6 / PEAK_HBM_GBS, bytes=nbytes, volumes=volumes, thresholds=11)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--mc', type=int, default=20, help='T: stochastic passes per volume')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-ws', action='store_true', help='skip the deterministic weight-scaling pass')
    ap.add_argument('--ensemble', type=int, default=0, metavar='K',
                    help='K ensemble members (seeds 20..20+K-1) instead of T MC passes (BASELINE config "BraTS ensemble")')
    ap.add_argument('--aleatoric', action='store_true',
                    help='sigma-head U-Net, per-pass sigma averaged next to the MC statistics (BASELINE config "BraTS aleatoric + MC", use --mc 50)')
    args = ap.parse_args()
    if args.aleatoric and args.ensemble:
        raise SystemExit('--aleatoric and --ensemble exclude each other')

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus {} but WORLD_SIZE={}: launch N>1 through torch.distributed.run'.format(args.gpus, world))
    # Test-only switches for a box with ONE GPU (RCCL refuses two ranks per device): all ranks on device 0 over gloo.
    single_device = os.environ.get('RCU_BENCH_SINGLE_DEVICE') == '1'
    backend = os.environ.get('RCU_BENCH_BACKEND', 'nccl')
    if single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from rcu_amd import distributed as rdist
    from rcu_amd import evaluation as ev
    from rcu_amd import steps

    T = args.mc
    seed = 20                                   # config seed (config/test_brats_baseline_mc.yaml:6)
    model = make_model(seed, device, sigma_out=args.aleatoric)
    x_cpu, mask_cpu, target_cpu = make_volume(seed)
    x = x_cpu.to(device)
    ctx = steps.TorchTestContext(str(device), model)
    if args.ensemble:
        T = args.ensemble
        members = [model] + [make_model(seed + k, device) for k in range(1, T)]
        runner = rdist.ShardedEnsembleRunner(members, rank=rank, world=world)
    elif args.aleatoric:
        members = [model]
        runner = rdist.ShardedAleatoricMcRunner(model, T, ws_pass=not args.no_ws, rank=rank, world=world)
    else:
        members = [model]
        runner = rdist.ShardedMcRunner(model, T, ws_pass=not args.no_ws, rank=rank, world=world)
    torch.manual_seed(seed + rank)              # dropout masks: independent streams per rank

    def one_step(k):
        # N>1: the reduce runs on RCCL's stream and the root finalises on a side stream, so the ranks'
        # compute streams do not meet at every volume (rcu_amd.distributed.ShardedMcRunner.step_async)
        return runner.step_async(x, k)

    for k in range(args.warmup):
        one_step(k).result()
    # per-kernel HIP events for this rank's forwards inside the timed region (per member in ensemble mode)
    my_jobs = [j for k in range(args.warmup, args.warmup + args.steps) for j in runner.jobs_of(k, rank)]
    for i, m in enumerate(members):
        count = sum(1 for j in my_jobs if j - 1 == i) if args.ensemble else len(my_jobs)
        m.profile_begin(HEIGHT, WIDTH, SLICES, max(count, 1))

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = [one_step(k) for k in range(args.warmup, args.warmup + args.steps)]
    out = [p.result() for p in pending][-1]
    runner.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    forwards, slot_ms = 0, None
    for m in members:
        cnt, ms = m.profile_collect(HEIGHT, WIDTH, SLICES)
        forwards += cnt
        slot_ms = ms if slot_ms is None else [a + b for a, b in zip(slot_ms, ms)]
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (this rank's launches in the timed region)
    layers = model.layer_table(HEIGHT, WIDTH, SLICES)
    per_kernel = {}
    for L, ms in zip(layers, slot_ms[1:1 + len(layers)]):
        e = per_kernel.setdefault(L['kernel'], dict(ms=0.0, flops=0.0, issued=0.0, launches=0))
        e['ms'] += ms
        e['flops'] += L['flops_per_slice'] * SLICES * forwards
        e['issued'] += L['mfma_flops_per_slice'] * SLICES * forwards
        e['launches'] += forwards
    dominant = max(per_kernel, key=lambda k_: per_kernel[k_]['ms'])
    d = per_kernel[dominant]
    conv_ms = sum(e['ms'] for e in per_kernel.values())
    conv_flops = sum(e['flops'] for e in per_kernel.values())
    conv_issued = sum(e['issued'] for e in per_kernel.values())
    roofline = dict(bound='mfma', kernel=dominant, achieved=d['flops'] / (d['ms'] * 1e-3) / 1e12,
                    peak=PEAK_FP32_MFMA_TFLOPS, unit='TFLOP/s', frac=d['flops'] / (d['ms'] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                    traffic=None, launches=d['launches'], avg_launch_ms=d['ms'] / max(d['launches'], 1),
                    flops_per_launch=d['flops'] / max(d['launches'], 1),
                    # flops the kernel actually issues to the MFMA pipe (padded channels, whole tiles) / peak:
                    mfma_pipe_frac=d['issued'] / (d['ms'] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                    # Winograd kernels execute 16/36 (conv units) or 9/36 (up-convolutions) of the canonical multiplications,
                    # so the algorithmic rate `achieved` can exceed the MFMA peak; the pipe itself is busy mfma_pipe_frac
                    executed_over_algorithmic=d['issued'] / d['flops'],
                    all_conv_kernels=dict(achieved=conv_flops / (conv_ms * 1e-3) / 1e12,
                                          frac=conv_flops / (conv_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                          mfma_pipe_frac=conv_issued / (conv_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                          ms_per_forward=conv_ms / max(forwards, 1)),
                    other_ms_per_forward=dict(input_relayout=slot_ms[0] / max(forwards, 1),
                                              head_softmax_accumulate=slot_ms[-1] / max(forwards, 1)),
                    per_kernel={k_: dict(ms_per_forward=e['ms'] / max(forwards, 1),
                                         tflops=e['flops'] / (e['ms'] * 1e-3) / 1e12) for k_, e in per_kernel.items()})
    # the fused head (1x1 classifier conv + softmax + entropy + accumulate into the statistics): an HBM scan.
    # It reads the 32-channel feature map instead of logits (the logits never exist in HBM) and
    # read-modify-writes the S=2 float32 statistics planes (94.4 MB per sample-volume if
    # logits were materialised; here 4*V*32 + 2*2*4*V bytes, V = voxels).
    vox = SLICES * HEIGHT * WIDTH
    head_bytes = 4.0 * vox * 32 + 2 * 2 * 4.0 * vox
    # In the timed region the classifier + softmax + statistics update run inside conv_cls.0's epilogue (one pass per
    # sample, two classes: csrc/rcu_wino.hip, wino_epilogue_head) and have no launch of their own; the standalone head kernel
    # -- the path of pass groups and of the sigma / feature outputs, same arithmetic, same bits -- is timed here, outside
    # the timed region, on the same volume (RCU_FUSE_HEAD=0 is read per forward).
    fused_head_ms = slot_ms[-1] / max(forwards, 1)
    os.environ['RCU_FUSE_HEAD'] = '0'
    try:
        probe = steps.McStatistics(SLICES, 2, HEIGHT, WIDTH, device)
        model.forward_accumulate(x, probe)
        model.profile_begin(HEIGHT, WIDTH, SLICES, 3)
        for _ in range(3):
            model.forward_accumulate(x, probe)
        torch.cuda.synchronize()
        cnt_h, ms_h = model.profile_collect(HEIGHT, WIDTH, SLICES)
        head_ms = ms_h[-1] / max(cnt_h, 1)
        del probe
    finally:
        del os.environ['RCU_FUSE_HEAD']
    roofline['other_ms_per_forward']['head_softmax_accumulate'] = fused_head_ms
    roofline['other_ms_per_forward']['head_fused_into'] = 'conv_cls.0 epilogue' if fused_head_ms < 0.5 * head_ms else None
    # the first conv kernel (csrc/rcu_first.hip) reads the NCHW input itself: no re-layout kernel in the timed region then
    roofline['other_ms_per_forward']['input_read_by'] = layers[0]['kernel'] if layers[0]['kernel'].startswith('conv3x3_first') else 'pack_input_kernel'
    roofline['aggregation'] = dict(bound='hbm', kernel='head_kernel', achieved=head_bytes / head_ms / 1e6, peak=PEAK_HBM_GBS,
                                   unit='GB/s', frac=head_bytes / head_ms / 1e6 / PEAK_HBM_GBS, bytes_per_launch=head_bytes,
                                   avg_launch_ms=head_ms, measured='standalone launches outside the timed region')
    pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(pmc_path):
        with open(pmc_path) as f:
            pmc = json.load(f)
        if dominant in pmc:
            roofline['traffic'] = pmc[dominant]

    # ---- parity numbers outside the timed region: ECE on the GPU maps vs the oracle on the same maps, and
    # (with the CPU baseline) the end-to-end difference on the slices the CPU path processed
    pred, p_fg = steps.prediction_and_foreground(out['probabilities'])
    ece_gpu = ev.ece_binary(p_fg, target_cpu, mask=mask_cpu)
    parity = dict(ece=ece_gpu)
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import calib_oracle as co
        p_np = p_fg.cpu().numpy()
        ece_oracle = co.ece_binary(np.stack([1 - p_np, p_np], -1), target_cpu.numpy(), mask=mask_cpu.numpy())
        parity['ece_delta_same_maps'] = abs(ece_gpu - ece_oracle)
        parity['bin_ids_equal'] = bool(np.array_equal(ev.bin_ids(p_np), co.bin_ids(p_np.reshape(-1))))
    if not args.no_cpu_baseline and world == 1 and not args.ensemble and not args.aleatoric:     # the CPU leg: rank 0 at N=1 only
        cpu, mask_sets, ref = cpu_baseline(model, x_cpu, T, seed)
        n = ref['probabilities'].shape[0]
        bc = steps.BatchContext({'images': x[:n].contiguous()}, 0)
        steps.McPredictStep(len(mask_sets), masks=mask_sets)(bc, None, ctx)
        steps.MultiPredictionSummary()(bc, None, ctx)
        parity['max_abs_dprob_vs_cpu'] = float((bc.output['probabilities'].cpu() - ref['probabilities']).abs().max())
        parity['max_abs_dentropy_vs_cpu'] = float((bc.output['entropy'].cpu() - ref['entropy']).abs().max())
        pg = bc.output['probabilities'][:, 1].cpu().numpy()
        pc = ref['probabilities'][:, 1].numpy()
        tg, mk = target_cpu[:n].numpy(), mask_cpu[:n].numpy()
        parity['ece_delta_vs_cpu'] = abs(ev.ece_binary(pg, tg, mask=mk) - co.ece_binary(np.stack([1 - pc, pc], -1), tg, mask=mk))

    result = {
        'metric': 'ensemble-member-volumes/sec (4x160x192x128, K={})'.format(T) if args.ensemble
                  else 'MC-sample-volumes/sec (4x160x192x128, T={}{})'.format(T, ', sigma head' if args.aleatoric else ''),
        'value': T * args.steps / elapsed,
        'unit': 'member-volumes/s' if args.ensemble else 'MC-sample-volumes/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'strong',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': ('BraTS ensemble: {} U-Net(2,4,depth 4,start_filters 32) members over 160 slices of 4x192x128 '
                                '+ mean/entropy aggregation per step'.format(T)) if args.ensemble else
                               ('BraTS {}: 2D U-Net(2,4,depth 4,start_filters 32,dropout 0.05{}) over 160 slices '
                                'of 4x192x128, T={} MC-dropout passes{} + mean/entropy aggregation per step'
                                .format('aleatoric + MC' if args.aleatoric else 'baseline_mc', ', sigma_out' if args.aleatoric else '',
                                        T, '' if args.no_ws else ' + weight-scaling pass')),
                   'T': T, 'ws_pass': not (args.no_ws or args.ensemble), 'slices': SLICES, 'height': HEIGHT, 'width': WIDTH,
                   'sharding': 'passes over ranks, one RCCL sum-reduce of the statistics per step' if world > 1 else 'none',
                   'gflop_per_sample_volume': conv_flops / max(forwards, 1) / 1e9},
        'roofline': roofline,
        'calibration_kernels': calibration_kernels(device) if world == 1 else None,
        'cpu_baseline': cpu,
        'parity': parity,
    }
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
