#!/usr/bin/env python3
"""Headline benchmark: MC-sample-volumes/s on a synthetic BraTS-shaped volume (4 x 160 x 192 x 128,
i.e. 160 slices of 4 x 192 x 128), T = 20 MC-dropout passes, on N GPUs of one node.

One step = the reference's hot path for one volume (rechun/dl/customsteps.py:16-71):
    weight-scaling pass + T stochastic U-Net passes (softmax + running statistics fused in) ->
    [N > 1: one RCCL sum-reduce of the per-voxel statistics] -> mean probability + predictive entropy.
The host-to-device copy of every volume is INSIDE the timed region (SURVEY.md 8d's definition of the metric), prefetched from
pinned memory on a copy stream while the previous volume computes (`value`); the same steps with the volume resident in HBM are
reported next to it (`resident`).  N > 1 shards the T+1 forward passes of every step over the ranks
(strong scaling).  Launch forms:
    python bench.py --gpus N ...                  (starts its N ranks itself, as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
`--workload isic` is BASELINE.json's configs[1] (ISIC baseline_mc: 32 images of 3 x 256 x 256, T = 20) with the same
JSON schema.  `--workload brats-native` / `isic-native` are the shapes the reference's real data has -- 155 slices of 4 x 240 x 240 (the
reference never crops BraTS: scripts/create_brats18_dataset.py:53-72) and 32 images of 3 x 192 x 256 (scripts/prepare_isic_data.py:29-30) --,
whose levels are not whole Winograd tiles and run on padded levels (rcu_unet_options.pad_levels); the default run carries the first as the
sub-record `native_shapes` (a child process of rank 0, behind the headline's timed region; never in the headline's place).

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel (HIP events recorded on the
launch stream between the kernels, see rcu_unet_profile_begin in include/rcu.h -- inside the timed region with
`--lanes 1`; with the default two stream lanes the kernels of the lanes overlap there, so the record comes from a serial
leg right behind the timed region and the overlapped times are kept under `roofline.timed_region`); `cpu_baseline`
is the oracle (a port of the reference's CPU path, pinned against golden vectors) timed on a bounded
sample of the same workload on this box's host cores -- the SAME slices, weights and dropout masks as the
last timed step, so the sample doubles as the parity check of the timed output (`parity`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')   # as rcu_amd/__init__.py does (before torch touches the HIP runtime)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')        # likewise: a hardware queue per stream (lanes, input prefetch, finalize side stream)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_HBM_GBS = 8000.0            # same guide: HBM3E peak (about 6.3 TB/s is what a streaming kernel reaches)
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
SLICES, CHANNELS, HEIGHT, WIDTH = 160, 4, 192, 128
MODEL_PARAMS = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05)  # config/train_brats_baseline.yaml:7-12
# BASELINE.json configs[1]: ISIC baseline_mc (config/test_isic_baseline_mc.yaml: batch_size 32, rescaled to 0..1, 3 x 256 x 256)
ISIC_IMAGES, ISIC_CHANNELS, ISIC_HEIGHT, ISIC_WIDTH = 32, 3, 256, 256
ISIC_PARAMS = dict(nb_classes=2, in_channels=3, depth=4, start_filters=32, dropout=0.05)


def make_model(seed, device, sigma_out=False, params=None):
    """UNet(2, 4, 4, 32, 0.05) with torch's default init under the seed and randomised BatchNorm statistics
     -- random-init weights of the named architecture; there are no checkpoints offline."""
    from rcu_amd.model import UNet
    torch.manual_seed(seed)
    model = UNet(**(params or MODEL_PARAMS), sigma_out=sigma_out)
    gen = torch.Generator().manual_seed(seed + 1000)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
    model.weights_changed()
    return model.to(device)


# the reference's real data (not BASELINE's benchmark shapes): a BraTS volume is 155 slices of 240 x 240 (never cropped:
# scripts/create_brats18_dataset.py:53-72; config/test_brats_baseline_mc.yaml:30-31 slices it), an ISIC image 192 x 256 (scripts/prepare_isic_data.py:29-30)
NATIVE_SLICES, NATIVE_HEIGHT, NATIVE_WIDTH = 155, 240, 240
ISIC_NATIVE_HEIGHT, ISIC_NATIVE_WIDTH = 192, 256


def make_volume(seed, n=SLICES, height=HEIGHT, width=WIDTH, slices=SLICES):
    """x ~ N(0,1) zeroed outside a centred ellipsoid (= the ECE brain mask); target = smaller ellipsoid."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(n, CHANNELS, height, width, generator=gen)
    zz, yy, xx = torch.meshgrid(torch.linspace(-1, 1, slices)[:n] if n <= slices else torch.linspace(-1, 1, n),
                                torch.linspace(-1, 1, height), torch.linspace(-1, 1, width), indexing='ij')
    r2 = (zz / 0.9) ** 2 + (yy / 0.85) ** 2 + (xx / 0.8) ** 2
    mask = r2 < 1.0
    target = ((zz / 0.35) ** 2 + ((yy - 0.1) / 0.3) ** 2 + ((xx + 0.1) / 0.3) ** 2 < 1.0).to(torch.uint8)
    x = x * mask[:, None].float()
    return x, mask, target


def make_isic_batch(seed, n=ISIC_IMAGES, height=ISIC_HEIGHT, width=ISIC_WIDTH):
    """x ~ U(0,1) [n, 3, 256, 256] (images rescaled to 0..1); target = a centred ellipse per image (a lesion),
    no evaluation mask (the ISIC evaluation has none)."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.rand(n, ISIC_CHANNELS, height, width, generator=gen)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, height), torch.linspace(-1, 1, width), indexing='ij')
    radius = 0.3 + 0.4 * torch.rand(n, generator=gen)
    target = (((yy / 0.9) ** 2 + xx ** 2)[None] < (radius ** 2)[:, None, None]).to(torch.uint8)
    return x, torch.ones_like(target, dtype=torch.bool), target


def _host_threads():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def host_description():
    """What makes a CPU figure comparable across boxes: the CPU model, the affinity set of this process, torch's thread settings."""
    model_name, sockets = None, set()
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name') and model_name is None:
                    model_name = line.split(':', 1)[1].strip()
                elif line.startswith('physical id'):
                    sockets.add(line.split(':', 1)[1].strip())
    except OSError:
        pass
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    runs, start = [], None       # compact "0-31,64-95" form of the affinity set
    for i, c in enumerate(cpus):
        if start is None:
            start = c
        if i + 1 == len(cpus) or cpus[i + 1] != c + 1:
            runs.append(str(start) if start == c else '{}-{}'.format(start, c))
            start = None
    info = {k.strip(): v.strip() for k, v in (ln.split(':', 1) for ln in torch.__config__.parallel_info().splitlines() if ':' in ln)}
    return dict(cpu_model=model_name, sockets=len(sockets) or None, logical_cpus=os.cpu_count(), affinity=','.join(runs),
                affinity_count=len(cpus), torch_threads=torch.get_num_threads(), torch_interop_threads=torch.get_num_interop_threads(),
                torch_parallel_backend=info.get('ATen parallel backend'), omp_max_threads=info.get('omp_get_max_threads()'),
                mkl_max_threads=info.get('mkl_get_max_threads()'),
                OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS'), MKL_NUM_THREADS=os.environ.get('MKL_NUM_THREADS'))


def cpu_leg(members, params, x_cpu, sel, mask_sets_sel, with_ws, ensemble, budget_s, thread_counts):
    """The oracle's CPU path (torch-CPU conv stack + torch aggregation, oracle/) on the slices `sel` of the volume with the
    weights and the dropout masks of the last timed step.  BASELINE.md section 4's protocol on a bounded sample: per thread count
    (all host threads, and 32 next to it when the box has more) one warm-up forward, then the MEDIAN of three timed runs of the
    whole sample (weight-scaling pass + T passes + aggregation); `value` is the faster of the two settings.
    -> (cpu_baseline dict, reference summary on those slices)."""
    from oracle import summary_oracle as so
    from oracle import unet_oracle as uo
    states = [{k: v.detach().cpu() for k, v in m.state_dict().items()} for m in members]
    xs = x_cpu[sel].contiguous()
    fwd = [lambda xx, m, st=st: uo.unet_forward(st, xx, m, **params) for st in states]
    n_total = x_cpu.shape[0]
    ref, runs, t_all = None, {}, time.perf_counter()
    for threads in thread_counts:
        torch.set_num_threads(threads)
        fwd[0](xs[:2], None)                        # warm-up (thread pool, primitive caches)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            if ensemble:
                multi = so.ensemble_probabilities(fwd, xs)
                ws = None
            else:
                ws, multi = so.mc_probabilities(fwd[0], xs, mask_sets_sel)
            ref = so.multi_prediction_summary(multi)
            times.append(time.perf_counter() - t0)
        runs[threads] = sorted(times)[1]
    # (outside the timing) the same passes' mutual information and variance: the reference of the all-outputs configuration
    ref.update(so.multi_prediction_summary(multi, True, True))
    if ws is not None and with_ws:
        ref['ws_probabilities'] = ws
    passes = multi.shape[0]
    units = passes * len(sel) / n_total            # sample-volume equivalents (the ws pass is timed, not counted)
    best = min(runs, key=runs.get)
    return dict(value=units / runs[best], cores=best, kind='port',
                by_threads={str(t): units / dt for t, dt in runs.items()}, host_threads=_host_threads(), sample_slices=int(len(sel)),
                host=host_description(),
                protocol='1 warm-up + median of 3 timed runs per thread count (BASELINE.md 4), on a bounded sample',
                sample='{} of {} slices x ({} {}{}) through oracle/ (torch-CPU); median run {:.2f} s at {} threads; the whole leg took {:.1f} s '
                       '(budget {:.0f} s)'.format(len(sel), n_total, passes, 'members' if ensemble else 'MC passes',
                                                  '' if ensemble else ' + ws pass', runs[best], best, time.perf_counter() - t_all,
                                                  budget_s)), ref


def cpu_thread_probe(threads, params, height, width, timeout_s=40.0):
    """Seconds per slice of the oracle's forward at ``threads`` intra-op threads, measured in a child process that is killed after
    ``timeout_s`` (-> None): on a box whose visible host threads are not all ours to use (a 256-thread host shared between pods)
    torch-CPU at the full thread count can take minutes for a forward that 32 threads finish in a second, and a thread pool that has
    gone down that road cannot be stopped from inside the process."""
    import subprocess
    code = ('import sys, time, torch\n'
            'sys.path.insert(0, {root!r})\n'
            'from oracle import unet_oracle as uo\n'
            'torch.set_num_threads({threads})\n'
            'params = {params!r}\n'
            'st = uo.synthetic_state(1, **params)\n'
            'x = torch.randn(4, params["in_channels"], {h}, {w})\n'
            'uo.unet_forward(st, x[:2], None, **params)\n'
            't0 = time.perf_counter()\n'
            'uo.unet_forward(st, x, None, **params)\n'
            'print("PER_SLICE", (time.perf_counter() - t0) / 4)\n').format(root=ROOT, threads=int(threads), params=dict(params),
                                                                           h=height, w=width)
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    try:
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return None
    for line in r.stdout.splitlines():
        if line.startswith('PER_SLICE'):
            return float(line.split()[1])
    return None


def cpu_thread_counts(params, height, width):
    """Thread counts of the CPU leg: every host thread and 32 next to it on a bigger box -- each only if its probe
    (a child process under a timeout) shows it usable: a count whose 4-slice forward does not finish, or runs more than 4x slower
    per slice than the best count, is reported as skipped instead of timed.  -> ([usable counts], {count: probe result})."""
    host = _host_threads()
    counts = [host] + ([32] if host > 32 else [])
    probes = {c: cpu_thread_probe(c, params, height, width) for c in counts}
    finished = {c: v for c, v in probes.items() if v is not None}
    if not finished:
        return [min(host, 32)], probes
    best = min(finished.values())
    return [c for c in counts if c in finished and finished[c] <= 4.0 * best], probes


def cpu_probe_slices(member, params, x_cpu, passes_total, budget_s, thread_counts, n_max=32):
    """How many slices the CPU leg can take within budget_s (three timed runs + a warm-up per thread count).  The time per slice
    depends on the batch (a 32-slice forward takes several times as long per slice as a 4-slice one: the activations fall out
    of the caches), so the estimate is checked on a forward of the chosen size and the size shrunk until the prediction fits."""
    from oracle import unet_oracle as uo
    torch.set_num_threads(thread_counts[0])
    state = {k: v.detach().cpu() for k, v in member.state_dict().items()}
    runs = 3.2 * len(thread_counts)                 # three timed runs + warm-up and set-up per thread count
    uo.unet_forward(state, x_cpu[:2].contiguous(), None, **params)

    def forward_s(n):
        xs = x_cpu[:n].contiguous()
        t0 = time.perf_counter()
        uo.unet_forward(state, xs, None, **params)
        return time.perf_counter() - t0

    n = 4
    per_slice = forward_s(n) / n
    for _ in range(4):
        fit = int(budget_s / max(per_slice * passes_total * runs, 1e-6))
        fit = max(2, min(n_max, x_cpu.shape[0], fit // 2 * 2))
        if fit == n:
            break
        n = fit
        per_slice = forward_s(n) / n                # measured at the size the leg would run
        if per_slice * n * passes_total * runs <= 1.15 * budget_s:
            break
    return n


def calibration_kernels(device, volumes=160, reps=5):
    """ECE histogram and uncertainty-error counts, timed with events on the launch stream, at two batch sizes: `volumes` = a test-split
    sized batch (160 BraTS volumes: the roofline figure) and the batch the PRODUCT issues -- `rcu_amd.evalrun.evaluate_runs` evaluates
    `batch_subjects` = 8 subjects per launch by default (bin-eval/eval_uncertainty.py --batch_subjects).  ALGORITHMIC bytes per voxel
    (SURVEY.md 8d): 4 (probability f32) + 1 (target) + 1 (mask) for the histogram; 4 + 1 (prediction) + 1 (target) for the counts, which
    the product takes straight from the probability map (rcu_unc_counts_from_p, no mask in the bnf_ue action); the map-thresholding
    form rcu_unc_counts (float32 uncertainty + prediction + target + mask, 7 bytes) is timed next to it."""
    import ctypes
    from rcu_amd import _lib
    from rcu_amd import evalrun
    lib = _lib.load()
    n = SLICES * HEIGHT * WIDTH
    g = torch.Generator(device=device).manual_seed(5)
    p = torch.rand((volumes, n), device=device, generator=g)
    t = (torch.rand((volumes, n), device=device, generator=g) < p).to(torch.uint8)
    m = (torch.rand((volumes, n), device=device, generator=g) < 0.4).to(torch.uint8)      # brain mask share
    pred = (p > 0.5).to(torch.uint8)
    stream = _lib.current_stream()
    import inspect
    product_batch = inspect.signature(evalrun.evaluate_runs).parameters['batch_subjects'].default
    out = {'product_batch_subjects': product_batch}

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
    ue = (ctypes.c_double * 11)(*([0.05] + [0.1 * k for k in range(1, 10)] + [0.95]))   # bin-eval/eval_uncertainty.py:239
    ue_table = (ctypes.c_double * 11)(*[lib.rcu_unc_from_p_threshold(i) for i in range(11)])
    for nv, suffix in ((volumes, ''), (product_batch, '_product_batch')):
        res = torch.empty(nv * ctypes.sizeof(_lib.EceResult), device=device, dtype=torch.uint8)
        ws = torch.empty(max(lib.rcu_ece_workspace_bytes(n, nv), 8), device=device, dtype=torch.uint8)
        ms = timed(lambda: _lib.check(lib.rcu_ece_hist(_lib.ptr(p), _lib.ptr(t), _lib.ptr(m), n, nv, thr, 10, _lib.ptr(res), _lib.ptr(ws), stream)))
        nbytes = nv * n * 6
        out['ece_hist' + suffix] = dict(bound='hbm', ms=ms, achieved=nbytes / ms / 1e6, peak=PEAK_HBM_GBS, unit='GB/s',
                                        frac=nbytes / ms / 1e6 / PEAK_HBM_GBS, bytes=nbytes, volumes=nv)
        cnt = torch.empty((nv, 11, 8), device=device, dtype=torch.int64)
        ws3 = torch.empty(lib.rcu_unc_from_p_workspace_bytes(n, nv), device=device, dtype=torch.uint8)
        ms = timed(lambda: _lib.check(lib.rcu_unc_counts_from_p(_lib.ptr(p), _lib.ptr(pred), _lib.ptr(t), None, n, nv, ue_table, 11,
                                                                _lib.ptr(cnt), _lib.ptr(ws3), stream)))
        nbytes = nv * n * 6
        out['unc_counts_from_p' + suffix] = dict(bound='hbm', ms=ms, achieved=nbytes / ms / 1e6, peak=PEAK_HBM_GBS, unit='GB/s',
                                                 frac=nbytes / ms / 1e6 / PEAK_HBM_GBS, bytes=nbytes, volumes=nv, thresholds=11,
                                                 what='the product path: counts from the float32 probability map (table of the reference\'s sets)')
        if not suffix:
            ws2 = torch.empty(max(lib.rcu_unc_workspace_bytes(n, nv), 8), device=device, dtype=torch.uint8)
            ms = timed(lambda: _lib.check(lib.rcu_unc_counts(_lib.ptr(p), 0, _lib.ptr(pred), _lib.ptr(t), _lib.ptr(m), n, nv,
                                                             ue, 11, _lib.ptr(cnt), _lib.ptr(ws2), stream)))
            nbytes = nv * n * 7
            out['unc_counts'] = dict(bound='hbm', ms=ms, achieved=nbytes / ms / 1e6, peak=PEAK_HBM_GBS, unit='GB/s',
                                     frac=nbytes / ms / 1e6 / PEAK_HBM_GBS, bytes=nbytes, volumes=nv, thresholds=11,
                                     what='thresholding an uncertainty map (confidence / sigma runs)')
    return out


class VolumePrefetcher:
    """The host-to-device copy of the volume inside the timed region (the metric counts "H2D of x";
    rechun/dl/customsteps.py:20 is where the reference moves the batch to the device): volume k + 1 goes from pinned host memory into
    one of two device buffers on a copy stream while volume k computes; the compute stream waits for ONE event per volume, and the
    copy stream does not overwrite a buffer before the step that read it has been issued completely (event behind the step)."""

    def __init__(self, x_cpu, device, depth=2):
        self.x_pin = x_cpu.pin_memory()
        self.bufs = [torch.empty(x_cpu.shape, device=device, dtype=x_cpu.dtype) for _ in range(depth)]
        self.stream = torch.cuda.Stream(device=device)
        self.ready = [None] * depth
        self.free = [None] * depth
        self.bytes = x_cpu.numel() * x_cpu.element_size()

    def issue(self, k):
        slot = k % len(self.bufs)
        with torch.cuda.stream(self.stream):
            if self.free[slot] is not None:
                self.stream.wait_event(self.free[slot])
            self.bufs[slot].copy_(self.x_pin, non_blocking=True)
            self.ready[slot] = torch.cuda.Event()
            self.ready[slot].record(self.stream)

    def get(self, k):
        slot = k % len(self.bufs)
        torch.cuda.current_stream().wait_event(self.ready[slot])
        return self.bufs[slot]

    def done(self, k):
        slot = k % len(self.bufs)
        self.free[slot] = torch.cuda.Event()
        self.free[slot].record()


def device_identity(device, rank):
    """What tells two GPUs of a node apart: the HIP device's UUID and PCI address, next to the rank's environment."""
    props = torch.cuda.get_device_properties(device)
    uuid = getattr(props, 'uuid', None)
    return dict(rank=rank, local_rank=int(os.environ.get('LOCAL_RANK', '0')), device_index=device.index, name=props.name,
                uuid=str(uuid) if uuid is not None else None,
                pci=('{:04x}:{:02x}:{:02x}'.format(getattr(props, 'pci_domain_id', 0), getattr(props, 'pci_bus_id', 0), getattr(props, 'pci_device_id', 0))
                     if hasattr(props, 'pci_bus_id') else None),
                hip_visible_devices=os.environ.get('HIP_VISIBLE_DEVICES'), rocr_visible_devices=os.environ.get('ROCR_VISIBLE_DEVICES'))


def volumes_per_step(world, jobs_per_volume, pass_group):
    """Consecutive volumes a rank takes as one batch.  T + 1 jobs per volume over `world` ranks leave a rank jobs / world passes per volume;
    while that is at least a full pass group (4 passes of 160 slices = 640 samples per launch) volumes run one by one, below it a rank's
    passes of v volumes run as groups of pass_group / v passes over 160 v slices -- the same 640-sample launches (measured on one GPU with
    rank 0's exact job list: tools/rank_share_of_world.py, profiles/r05_rank_share_of_8.json)."""
    v = 1
    while v < pass_group and jobs_per_volume * v < world * pass_group:
        v *= 2
    return v


def aggregation_kernels(device, n_slices, height, width, reps=30, all_outputs=False):
    """The standalone aggregation kernels of the step seam (rcu_mc_accumulate: softmax of a logits volume into the statistics;
    rcu_mc_finalize: mean + entropy out of them; rechun/dl/customsteps.py:57-61), timed with events on the launch stream.
    ALGORITHMIC bytes: accumulate V*C*4 logits + 2*S*4*V statistics read-modify-write (S = 2); finalize S*4*V read +
    (C+1)*4*V written."""
    from rcu_amd import steps
    vox = n_slices * height * width
    logits = torch.randn((n_slices, 2, height, width), device=device)
    stats = steps.McStatistics(n_slices, 2, height, width, device, all_outputs, all_outputs)
    out = {}
    # all outputs (mutual information + variance): S = 5 planes [sum p (2)] [sum p^2 (2)] [sum H].  ALGORITHMIC bytes by SURVEY.md 8d's
    # formula (float32 planes): accumulate V*C*4 + 2*S*4*V, finalize S*4*V + (C+3)*4*V; the planes are float64 here (exact variance),
    # so the kernels MOVE 2*S*8*V resp. S*8*V of statistics -- reported next to it as `moved`.
    S = 5 if all_outputs else 2

    def timed(fn):      # (30 launches: the first one waits ~10 us for the host's enqueue, a sixth of these kernels' time if only 5 share it)
        fn()
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    word = 8 if all_outputs else 4
    for name, fn, nbytes, moved in (('mc_accumulate_kernel', lambda: stats.accumulate(logits), vox * (2 * 4 + 2 * S * 4),
                                     vox * (2 * 4 + 2 * S * word)),
                                    ('mc_finalize_kernel', lambda: stats.finalize(all_outputs, all_outputs),
                                     vox * (S * 4 + (5 if all_outputs else 3) * 4), vox * (S * word + (5 if all_outputs else 3) * 4))):
        ms = timed(fn)
        out[name] = dict(bound='hbm', avg_launch_ms=ms, bytes_per_launch=nbytes, achieved=nbytes / ms / 1e6, peak=PEAK_HBM_GBS,
                         unit='GB/s', frac=nbytes / ms / 1e6 / PEAK_HBM_GBS)
        if all_outputs:
            out[name].update(moved_bytes_per_launch=moved, moved_gbs=moved / ms / 1e6, moved_frac=moved / ms / 1e6 / PEAK_HBM_GBS)
    return out


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a child process on a free rendezvous port of 127.0.0.1.  Never an exec, and only from a process that has made no GPU call."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the contract is ONE JSON line on stdout: relay rank 0's line and send whatever else the ranks' libraries print there to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        target = sys.stdout if line.lstrip().startswith('{"metric"') else sys.stderr
        target.write(line)
        target.flush()
    return proc.wait()


def plan_fingerprint(layers, samples_per_launch):
    """Short hash of what determines a kernel's HBM traffic per launch: the layer table (kernel instantiation and shape of every layer)
    and the samples a launch covers."""
    import hashlib
    text = ';'.join('{name}|{kernel}|{cin}|{cout}|{height}x{width}|{grid_height}x{grid_width}'.format(**L) for L in layers) + ';n={}'.format(samples_per_launch)
    return hashlib.sha256(text.encode()).hexdigest()[:12]


def native_sub_record(T, lanes, steps=4, warmup=1):
    """`python bench.py --workload brats-native --brief` as a child process (started fresh: no exec from this GPU-touched process), condensed."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--workload', 'brats-native', '--brief', '--steps', str(steps), '--warmup', str(warmup),
           '--mc', str(T), '--lanes', str(lanes), '--cpu-budget', '8']
    try:
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, env=dict(os.environ, RCU_BENCH_NATIVE='0'))
        line = [ln for ln in proc.stdout.decode().splitlines() if ln.startswith('{"metric"')][-1]
        r = json.loads(line)
    except Exception as exc:  # noqa: BLE001 - the headline must not depend on the sub-record
        return dict(error='{}: {}'.format(type(exc).__name__, exc))
    roof = r['roofline']
    keep = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'achieved_canonical', 'canonical_frac', 'traffic', 'launches', 'avg_launch_ms',
            'flops_per_launch', 'executed_flops_per_launch', 'all_conv_kernels', 'per_kernel', 'plan_fingerprint', 'measured_in')
    return {'brats_155x240x240': dict(metric=r['metric'], value=r['value'], unit=r['unit'], steps=r['steps'], warmup=r['warmup'],
                                      ms_per_step=r['ms_per_step'], config=r['config'], resident=r['resident'],
                                      roofline={k: roof.get(k) for k in keep}, parity=r['parity'],
                                      cpu_baseline={k: r['cpu_baseline'].get(k) for k in ('value', 'unit', 'cores', 'kind', 'sample')} if r.get('cpu_baseline') else None,
                                      command=' '.join(cmd[1:]))}


def sharding_record(runner, world, args, n_slices, height, width, v_step):
    """What crosses xGMI per volume: one sum-reduce of the per-voxel statistics (exact float64 sums: two planes for mean + entropy) and the
    weight-scaling probabilities -- point to point from their owner by default on RCCL (round 6), or in the reduce buffer's tail."""
    vox = n_slices * height * width
    planes = 2 + (3 if args.all_outputs else 0)
    stats_bytes = planes * 8 * vox
    ws_bytes = 0 if (args.no_ws or args.ensemble) else 2 * 8 * vox
    transport = getattr(runner, 'ws_transport', None) or ('p2p' if dist.get_backend() == 'nccl' else 'reduce')
    return dict(what='MC passes / members over ranks (job i of step k on rank (i + k * jobs) mod world), one RCCL sum-reduce of the statistics per step',
                ws_transport=transport if ws_bytes else None,
                reduce_bytes_per_volume=stats_bytes + (ws_bytes if transport == 'reduce' else 0),
                p2p_bytes_per_volume=ws_bytes if transport == 'p2p' else 0,
                p2p_note='only on the volumes whose weight-scaling pass the root does not run itself' if (ws_bytes and transport == 'p2p') else None,
                statistics='exact float64 sums ({} planes of {} voxels)'.format(planes, vox), volumes_per_step=v_step)


def split_masks(model, flat, n, rows):
    """Concatenated device mask tensor [site][n][C_site] -> list of per-site [len(rows), C_site] CPU tensors."""
    out, off = [], 0
    for _, c in model.dropout_sites():
        out.append(flat[off:off + n * c].view(n, c)[rows].cpu())
        off += n * c
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mc', type=int, default=20, help='T: stochastic passes per volume')
    ap.add_argument('--workload', choices=('brats', 'isic', 'brats-native', 'isic-native'), default='brats',
                    help='brats: 160 slices of 4x192x128 (the headline, BASELINE configs[2]); isic: 32 images of 3x256x256 (configs[1]); '
                         'brats-native / isic-native: the shapes of the reference\'s real data, 155 slices of 4x240x240 / 32 images of 3x192x256 '
                         '(padded levels)')
    ap.add_argument('--brief', action='store_true',
                    help='the timed region, roofline and parity only: no all-outputs leg, no standalone aggregation / calibration kernel probes, '
                         'no native-shape sub-record (what the default run\'s `native_shapes` child runs with)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=24.0, help='seconds of CPU work the oracle leg may take (bounded sample)')
    ap.add_argument('--ws-transport', choices=('reduce', 'p2p'), default=None,
                    help='N > 1: the weight-scaling probabilities in the tail of the one reduce buffer (default) or by send / recv from '
                         'their owner (rcu_amd.distributed.ShardedMcRunner)')
    ap.add_argument('--no-ws', action='store_true', help='skip the deterministic weight-scaling pass')
    ap.add_argument('--pass-group', type=int, default=0,
                    help='MC passes of a rank per launch (N * g samples per batch); 0 = McPredictStep\'s rule (steps.pass_group_size): 4 for the '
                         '160-slice volume (2 with the sigma head), 7 for the ISIC batch; the groups of a volume are sized so that the lanes '
                         'carry the same number of passes (steps.balanced_groups)')
    ap.add_argument('--lanes', type=int, default=2,
                    help='HIP streams a rank spreads the launches of a volume over (one workspace + statistics blob each; rcu_amd.distributed)')
    ap.add_argument('--ensemble', type=int, default=0, metavar='K',
                    help='K ensemble members (seeds 20..20+K-1) instead of T MC passes (BASELINE config "BraTS ensemble")')
    ap.add_argument('--aleatoric', action='store_true',
                    help='sigma-head U-Net, per-pass sigma averaged next to the MC statistics (BASELINE config "BraTS aleatoric + MC", use --mc 50)')
    ap.add_argument('--all-outputs', action='store_true',
                    help='every output of MultiPredictionSummary in the timed region: mutual information + variance next to mean + entropy '
                         '(do_mi + do_var, rechun/dl/customsteps.py:63-71: float64 statistics, S = 5 planes).  The default run reports this '
                         'configuration as the sub-record `all_outputs`')
    ap.add_argument('--volumes-per-step', type=int, default=0,
                    help='consecutive volumes a rank takes as ONE batch of 160 v slices, its passes of them in one launch (0 = the rule of '
                         'volumes_per_step(): 1 while a rank\'s share of a volume fills 640-sample launches, 2 at 8 ranks)')
    ap.add_argument('--watchdog', type=float, default=1500.0,
                    help='seconds after which a run that has not finished dumps the stacks of all threads to stderr and exits (0 = off)')
    args = ap.parse_args()
    if args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)
    if args.aleatoric and args.ensemble:
        raise SystemExit('--aleatoric and --ensemble exclude each other')

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as a fresh child (this process has not touched the GPU and
        # never will), relay the child's output -- rank 0 prints the one JSON line -- and exit with its return code
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints its version banner to file descriptor 1 when the
    # communicator is created): from here on descriptor 1 is this process's stderr, and the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit('--gpus {} but WORLD_SIZE={}'.format(args.gpus, world))
    # Test-only switches for a box with ONE GPU (RCCL refuses two ranks per device): all ranks on device 0 over gloo.
    single_device = os.environ.get('RCU_BENCH_SINGLE_DEVICE') == '1'
    backend = os.environ.get('RCU_BENCH_BACKEND', 'nccl')
    if single_device:
        local_rank = 0
        if world * args.lanes > 10:
            # every rank sizes its plans for the canonical 640-sample launch on every lane (24.4 GB each): more than ten of them do not fit the ONE
            # GPU this rehearsal puts all ranks on
            args.lanes = 1
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    # Test-only switch: RCU_BENCH_FORCE_PG=1 at N = 1 initialises the RCCL process group with ONE rank and routes every step through the
    # exchange path (asynchronous reduce on RCCL's stream, root finalize on a side stream, record_stream, drain) -- the rehearsal of the
    # N > 1 code on a box with one GPU (tools/rccl_world1_rehearsal.py; profiles/r04_rccl_world1.txt).
    force_pg = world == 1 and os.environ.get('RCU_BENCH_FORCE_PG') == '1'
    seed = 20                                   # config seed (config/test_brats_baseline_mc.yaml:6)
    isic = args.workload in ('isic', 'isic-native')
    native = args.workload.endswith('-native')
    if isic:
        n_slices, height, width = (ISIC_IMAGES, ISIC_NATIVE_HEIGHT, ISIC_NATIVE_WIDTH) if native else (ISIC_IMAGES, ISIC_HEIGHT, ISIC_WIDTH)
        x_cpu, mask_cpu, target_cpu = make_isic_batch(seed, n_slices, height, width)
    else:
        n_slices, height, width = (NATIVE_SLICES, NATIVE_HEIGHT, NATIVE_WIDTH) if native else (SLICES, HEIGHT, WIDTH)
        x_cpu, mask_cpu, target_cpu = make_volume(seed, n_slices, height, width, n_slices)
    feeder = VolumePrefetcher(x_cpu, device)
    # The RCCL process group is initialised WITHOUT `device_id=` (the communicator is then created at the first collective, in the warm-up
    # steps).  Measured at N = 1 through a one-rank group (round 4, profiles/r04_pg_h2d.txt): with `device_id=device` (eager creation at
    # init_process_group) every step that overlaps a prefetched host-to-device copy runs 4 ms longer -- 166.8-168.2 against 172.5-174.3
    # MC-sample-volumes/s, proportional to the step count, the same steps on a resident volume unaffected, GPU_MAX_HW_QUEUES and the
    # order of the first copy irrelevant -- while the lazily created communicator costs 0.5 % (173.3-173.5; tools/pg_h2d_probe.py is the A/B).
    if force_pg:
        import socket
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            with socket.socket() as sock:
                sock.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
        dist.init_process_group(backend, rank=0, world_size=1)
    collective = world > 1 or force_pg          # a process group exists: barriers and max-over-ranks as the contract prescribes
    barrier_kwargs = dict(device_ids=[local_rank]) if (collective and backend == 'nccl') else {}
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl')   # "nccl" is RCCL on ROCm; the rank's device is the current device (set_device above)
        else:
            dist.init_process_group(backend)

    from rcu_amd import distributed as rdist
    from rcu_amd import evaluation as ev
    from rcu_amd import steps

    T = args.mc
    params = ISIC_PARAMS if isic else MODEL_PARAMS
    unit_name = 'image' if isic else 'volume'
    model = make_model(seed, device, sigma_out=args.aleatoric, params=params)
    if args.ensemble:
        args.pass_group = 1                     # members run one per launch (they differ in their weights)
    elif args.pass_group < 1:                   # McPredictStep's rule: GROUP_PIXELS worth of pixels, no tensor beyond 2 GB
        args.pass_group = steps.pass_group_size(model, n_slices, height, width, steps.McPredictStep.GROUP_PIXELS)
    x = x_cpu.to(device)
    if args.ensemble:
        T = args.ensemble
        members = [model] + [make_model(seed + k, device, params=params) for k in range(1, T)]
        runner = rdist.ShardedEnsembleRunner(members, rank=rank, world=world, lanes=args.lanes)
    elif args.aleatoric:
        members = [model]
        runner = rdist.ShardedAleatoricMcRunner(model, T, ws_pass=not args.no_ws, rank=rank, world=world, seed=seed, lanes=args.lanes,
                                                pass_group=args.pass_group)
        runner.ws_transport = args.ws_transport or runner.ws_transport
    else:
        members = [model]
        runner = rdist.ShardedMcRunner(model, T, ws_pass=not args.no_ws, rank=rank, world=world, seed=seed,
                                       pass_group=args.pass_group, lanes=args.lanes, ws_transport=args.ws_transport,
                                       do_mi=args.all_outputs, do_var=args.all_outputs)
    runner.force_exchange = force_pg
    # dropout masks: drawn per (seed, step, pass) by the runner -- the same T samples whatever the world size
    # Volumes per runner step: a rank of a large world takes v consecutive volumes as one batch of n_slices * v slices and groups
    # pass_group / v passes of it per launch -- the launches keep the size they have on one GPU (volumes_per_step above); `--steps K` stays
    # the number of VOLUMES: K // v steps of v volumes and K % v single ones.
    samples_per_launch = n_slices * args.pass_group
    v_step = args.volumes_per_step if args.volumes_per_step > 0 else volumes_per_step(world, runner.jobs_per_step, args.pass_group)
    if args.ensemble:
        v_step = 1
    runner.group_samples = samples_per_launch if not args.ensemble else None
    x_by_size = {1: x}
    feeders = {1: feeder}
    if v_step > 1:
        x_cpu_v = torch.cat([x_cpu] * v_step)
        x_by_size[v_step] = x_cpu_v.to(device)
        feeders[v_step] = VolumePrefetcher(x_cpu_v, device)

    def chunks(count):
        return [v_step] * (count // v_step) + [1] * (count % v_step)

    step_counter = [0]

    def run_volumes(run, count, fed=True):
        """`count` volumes through runner `run` in chunks; fed: every chunk comes from pinned host memory on the copy stream (prefetched one
        chunk ahead), else the chunks read the resident device copies.  -> (pending summaries, their step indices, their sizes)"""
        sizes = chunks(count)
        idx = list(range(step_counter[0], step_counter[0] + len(sizes)))
        step_counter[0] += len(sizes)
        if fed and sizes:
            feeders[sizes[0]].issue(idx[0])
        pend = []
        for i, (k, size) in enumerate(zip(idx, sizes)):
            xin = feeders[size].get(k) if fed else x_by_size[size]
            if fed and i + 1 < len(sizes):
                feeders[sizes[i + 1]].issue(idx[i + 1])
            # N>1: the reduce runs on RCCL's stream and the root finalises on a side stream, so the ranks' compute streams do not meet at
            # every step (rcu_amd.distributed.ShardedMcRunner.step_async)
            pend.append(run.step_async(xin, k))
            if fed:
                feeders[size].done(k)
        return pend, idx, sizes

    def bracket():
        torch.cuda.synchronize()
        if collective:
            dist.barrier(**barrier_kwargs)
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if not collective:
            return seconds
        tmax = torch.tensor([seconds], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    for p_ in run_volumes(runner, args.warmup, fed=False)[0]:
        p_.result()
    # per-kernel HIP events for this rank's forwards inside the timed region (per member in ensemble mode)
    timed_sizes = chunks(args.steps)
    timed_idx = list(range(step_counter[0], step_counter[0] + len(timed_sizes)))
    my_jobs = [j for k in timed_idx for j in runner.jobs_of(k, rank)]
    # (ensemble members keep to one stream lane: a member of a side lane never runs on its lane-0 handle -- which is the one these
    # calls create and time -- so it is left alone here: 6 GB of workspace per member that nobody would use)
    timed_members = [(i, m) for i, m in enumerate(members) if not (args.ensemble and args.lanes > 1 and i % args.lanes != 0)]
    for i, m in timed_members:
        count = sum(1 for j in my_jobs if j - 1 == i) if args.ensemble else len(my_jobs)
        m.profile_begin(height, width, samples_per_launch, max(count, 1))
    runner.forwards_run = 0

    bracket()
    allocs_before = torch.cuda.memory_stats(device).get('num_device_alloc', 0)     # hipMalloc calls of torch's caching allocator so far
    t0 = time.perf_counter()
    # THE timed region: per volume the host-to-device copy (prefetched: the next chunk travels while this one computes), the
    # weight-scaling pass, the T stochastic passes, [N > 1: the reduce] and the finalize
    pending, timed_idx, timed_sizes = run_volumes(runner, args.steps)
    out = [p_.result() for p_ in pending][-1]
    runner.drain()
    bracket()
    elapsed = time.perf_counter() - t0
    device_allocs_in_timed_region = torch.cuda.memory_stats(device).get('num_device_alloc', 0) - allocs_before
    last_step, last_size = timed_idx[-1], timed_sizes[-1]
    if out is not None and last_size > 1:       # the first volume of the last chunk stands for it in the parity leg
        out = {k_: v_[:n_slices] for k_, v_ in out.items()}
    # forward passes in units of one volume (a job of a v-volume chunk is v of them)
    weight = {k: sz for k, sz in zip(timed_idx, timed_sizes)}
    forwards_this_rank = sum(weight[k] * len(runner.jobs_of(k, rank)) for k in timed_idx)
    passes_run = max(forwards_this_rank, 1)         # this rank's forward passes inside the timed region
    forwards_per_rank = [forwards_this_rank]
    devices_seen = None
    elapsed = max_over_ranks(elapsed)
    if collective:
        counts = torch.zeros(world, device=device, dtype=torch.int64)
        counts[rank] = forwards_this_rank
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        forwards_per_rank = [int(v) for v in counts.tolist()]
        # which GPUs the ranks ran on: all-gathered identities (the driver checks that N ranks mean N distinct devices)
        devices_seen = [None] * dist.get_world_size()
        dist.all_gather_object(devices_seen, device_identity(device, rank))
    else:
        devices_seen = [device_identity(device, rank)]
    n_ranks_seen = dist.get_world_size() if collective else 1

    launches, slot_ms = 0, None
    for _, m in timed_members:
        cnt, ms = m.profile_collect(height, width, samples_per_launch)
        launches += cnt
        slot_ms = ms if slot_ms is None else [a + b for a, b in zip(slot_ms, ms)]
    # ---- the same steps with the volume already resident in HBM (the secondary figure; all ranks take part)
    bracket()
    t1 = time.perf_counter()
    for p_ in run_volumes(runner, args.steps, fed=False)[0]:
        p_.result()
    runner.drain()
    bracket()
    elapsed_resident = max_over_ranks(time.perf_counter() - t1)

    # ---- the all-outputs configuration (north_star: "softmax -> running-mean / variance -> predictive-entropy"; the reference's
    # MultiPredictionSummary(do_mi=True, do_var=True), rechun/dl/customsteps.py:63-71): the same step with mutual information and variance
    # tracked -- exact float64 statistics, S = 5 planes per voxel -- timed like the headline (H2D prefetched inside, all ranks, one reduce per
    # step), over the SAME step indices as the headline's timed region: the runner's masks are a function of (seed, step, pass), so its last
    # volume runs under the masks of the headline's last volume and the CPU leg below is the reference of both.
    all_out, out_ao = None, None
    if not (args.all_outputs or args.ensemble or args.aleatoric or args.brief):
        ao_volumes = sum(timed_sizes[-min(len(timed_sizes), 5):])
        runner_ao = rdist.ShardedMcRunner(model, T, ws_pass=not args.no_ws, rank=rank, world=world, seed=seed, pass_group=args.pass_group,
                                          lanes=args.lanes, ws_transport=args.ws_transport, do_mi=True, do_var=True, force_exchange=force_pg)
        runner_ao.group_samples = samples_per_launch
        runner_ao.step_async(x, last_step + 1000).result()       # warm: the float64 blobs of the lanes come out of torch's allocator
        runner_ao.drain()
        saved = step_counter[0]
        step_counter[0] = timed_idx[-min(len(timed_sizes), 5)]
        bracket()
        ta = time.perf_counter()
        pend_a, idx_a, sizes_a = run_volumes(runner_ao, ao_volumes)
        out_ao = [p_.result() for p_ in pend_a][-1]
        runner_ao.drain()
        bracket()
        elapsed_ao = max_over_ranks(time.perf_counter() - ta)
        assert idx_a[-1] == last_step and sizes_a[-1] == last_size
        step_counter[0] = saved
        if out_ao is not None and last_size > 1:
            out_ao = {k_: v_[:n_slices] for k_, v_ in out_ao.items()}
        del pend_a
        all_out = dict(value=T * (n_slices if isic else 1) * ao_volumes / elapsed_ao, unit='MC-sample-{}s/s'.format(unit_name),
                       ms_per_step=elapsed_ao / ao_volumes * 1e3, steps=ao_volumes, warmup=1,
                       outputs=['probabilities', 'entropy', 'mutual_info', 'variance'] + ([] if args.no_ws else ['ws_probabilities']),
                       statistics='exact float64 sums, S = 5 planes per voxel: sum p_c (2), sum p_c^2 (2), sum H(p_t)',
                       h2d='prefetched, inside the timed steps (as the headline)',
                       reduce_bytes_per_volume=(5 * 8 + (0 if args.no_ws else 2 * 8)) * n_slices * height * width if world > 1 else 0)
    if rank != 0:
        if collective:
            dist.destroy_process_group()
        return

    # ---- stream lanes: in the timed region the kernels of different lanes overlap (that is their point: a lane fills the gaps between
    # the dependent layers of the other), so a kernel's start-to-end time there includes the other lane's work and says nothing about
    # the kernel.  The roofline record below is therefore taken from a SERIAL leg -- the same steps on one lane, HIP events between the
    # kernels -- right behind the timed region; the timed region's own (overlapped) event times are kept under `timed_region`.
    timed_region = None
    if args.lanes > 1:
        dom_ms, dom_layers = {}, {}
        for L, ms in zip(model.layer_table(height, width, n_slices * args.pass_group), slot_ms[1:]):
            name = L['kernel'] + ('+head' if L['head_fusable'] and model.fuse_head else '')
            dom_ms[name] = dom_ms.get(name, 0.0) + ms
            dom_layers[name] = dom_layers.get(name, 0) + 1
        timed_region = dict(lanes=args.lanes, wall_ms_per_forward=elapsed * 1e3 / passes_run, profiled_forward_launches_lane0=launches,
                            kernel_ms_per_launch_lane0={k_: v / max(launches * dom_layers[k_], 1) for k_, v in dom_ms.items()},
                            note='start-to-end times of lane 0\'s kernels while the other lane(s) run: overlapped, not exclusive')
        if args.ensemble:
            serial = rdist.ShardedEnsembleRunner(members, lanes=1)        # rank 0 alone: no collective in this leg
        elif args.aleatoric:
            serial = rdist.ShardedAleatoricMcRunner(model, T, ws_pass=not args.no_ws, seed=seed, lanes=1, pass_group=args.pass_group)
        else:
            serial = rdist.ShardedMcRunner(model, T, ws_pass=not args.no_ws, seed=seed, pass_group=args.pass_group, lanes=1)
        serial_steps = max(1, min(args.steps, 2))
        first = step_counter[0] + 1008
        serial.step(x, first - 1)                     # warm (the lane-0 workspace is the one the timed region used)
        serial_jobs = [j for k in range(first, first + serial_steps) for j in serial.jobs_of(k, 0)]
        for i, m in enumerate(members):
            count = sum(1 for j in serial_jobs if j - 1 == i) if args.ensemble else len(serial_jobs)
            m.profile_begin(height, width, n_slices * args.pass_group, max(count, 1))
        serial.forwards_run = 0
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for k in range(first, first + serial_steps):
            serial.step(x, k)
        torch.cuda.synchronize()
        timed_region['serial_leg'] = dict(steps=serial_steps, ms_per_step=(time.perf_counter() - ts) * 1e3 / serial_steps)
        passes_run = max(serial.forwards_run, 1)
        launches, slot_ms = 0, None
        for m in members:
            cnt, ms = m.profile_collect(height, width, n_slices * args.pass_group)
            launches += cnt
            slot_ms = ms if slot_ms is None else [a + b for a, b in zip(slot_ms, ms)]

    # ---- roofline of the dominant kernel (this rank's launches in the timed region; with stream lanes: in the serial leg above).  A launch
    # covers one pass of the volume, or pass_group passes (n_slices * g samples) where the runner grouped them: FLOPs count the passes,
    # launches the kernel launches.
    g = args.pass_group
    layers = model.layer_table(height, width, n_slices * g)
    per_kernel = {}
    for L, ms in zip(layers, slot_ms[1:1 + len(layers)]):
        # (conv_cls.0 runs with the classifier head in its epilogue: another kernel than the plain form of the same tile, under its own name)
        name = L['kernel'] + ('+head' if L['head_fusable'] and model.fuse_head else '')
        e = per_kernel.setdefault(name, dict(ms=0.0, flops=0.0, issued=0.0, launches=0))
        e['ms'] += ms
        e['flops'] += L['flops_per_slice'] * n_slices * passes_run
        e['issued'] += L['mfma_flops_per_slice'] * n_slices * passes_run
        e['launches'] += launches
    dominant = max(per_kernel, key=lambda k_: per_kernel[k_]['ms'])
    d = per_kernel[dominant]
    conv_ms = sum(e['ms'] for e in per_kernel.values())
    conv_flops = sum(e['flops'] for e in per_kernel.values())
    conv_issued = sum(e['issued'] for e in per_kernel.values())
    tf = lambda fl, ms_: fl / (ms_ * 1e-3) / 1e12  # noqa: E731
    # `achieved` / `frac`: FLOPs the kernel EXECUTES on the matrix pipe (padded channels, whole tiles) per second / peak -- the
    # physical utilisation, <= 1, what SQ_VALU_MFMA_BUSY_CYCLES confirms.  The Winograd kernels execute 16/36 (F(2x2,3x3)),
    # 36/144 (F(4x4,3x3)) or 9/36 (up-convolutions) of the canonical direct-convolution multiplications, so the ALGORITHMIC rate
    # (2*Cin*Cout*9*H*W per layer) is reported apart as achieved_canonical / canonical_frac and can exceed 1.
    roofline = dict(bound='mfma', kernel=dominant, achieved=tf(d['issued'], d['ms']), peak=PEAK_FP32_MFMA_TFLOPS, unit='TFLOP/s',
                    frac=tf(d['issued'], d['ms']) / PEAK_FP32_MFMA_TFLOPS,
                    achieved_canonical=tf(d['flops'], d['ms']), canonical_frac=tf(d['flops'], d['ms']) / PEAK_FP32_MFMA_TFLOPS,
                    executed_over_algorithmic=d['issued'] / d['flops'],
                    traffic=None, launches=d['launches'], avg_launch_ms=d['ms'] / max(d['launches'], 1),
                    flops_per_launch=d['flops'] / max(d['launches'], 1), executed_flops_per_launch=d['issued'] / max(d['launches'], 1),
                    all_conv_kernels=dict(achieved=tf(conv_issued, conv_ms), frac=tf(conv_issued, conv_ms) / PEAK_FP32_MFMA_TFLOPS,
                                          achieved_canonical=tf(conv_flops, conv_ms),
                                          canonical_frac=tf(conv_flops, conv_ms) / PEAK_FP32_MFMA_TFLOPS,
                                          ms_per_forward=conv_ms / passes_run),
                    other_ms_per_forward=dict(input_relayout=slot_ms[0] / passes_run,
                                              head_softmax_accumulate=slot_ms[-1] / passes_run),
                    per_kernel={k_: dict(ms_per_forward=e['ms'] / passes_run, tflops_executed=tf(e['issued'], e['ms']),
                                         frac=tf(e['issued'], e['ms']) / PEAK_FP32_MFMA_TFLOPS,
                                         tflops_canonical=tf(e['flops'], e['ms'])) for k_, e in per_kernel.items()})
    roofline['measured_in'] = 'the timed region' if timed_region is None else 'a serial leg behind the timed region (one lane; see timed_region)'
    if timed_region is not None:
        roofline['timed_region'] = timed_region
    # the fused head (1x1 classifier conv + softmax + entropy + accumulate into the statistics): an HBM scan.
    # It reads the 32-channel feature map instead of logits (the logits never exist in HBM) and
    # read-modify-writes the S=2 float32 statistics planes (94.4 MB per sample-volume if
    # logits were materialised; here 4*V*32 + 2*2*4*V bytes, V = voxels).
    vox = n_slices * height * width
    head_bytes = 4.0 * vox * 32 + 2 * 2 * 4.0 * vox
    # In the timed region the classifier + softmax + statistics update run inside conv_cls.0's epilogue (two classes:
    # csrc/rcu_wino.hip, wino_epilogue_head; pass groups too) and have no launch of their own; the standalone head kernel
    # -- the path of the sigma / feature outputs and of more than two classes, same arithmetic, same bits -- is timed here, outside
    # the timed region, on the same volume (UNet.set_fuse_head = rcu_unet_set_fuse_head, include/rcu.h).
    if not args.brief:
        fused_head_ms = slot_ms[-1] / passes_run
        model.set_fuse_head(False)
        try:
            probe = steps.McStatistics(n_slices, 2, height, width, device)
            model.forward_accumulate(x, probe)
            model.profile_begin(height, width, n_slices, 3)
            for _ in range(3):
                model.forward_accumulate(x, probe)
            torch.cuda.synchronize()
            cnt_h, ms_h = model.profile_collect(height, width, n_slices)
            head_ms = ms_h[-1] / max(cnt_h, 1)
            del probe
            # the same kernel with every output tracked (mutual information + variance: float64 statistics, S = 5 planes)
            probe = steps.McStatistics(n_slices, 2, height, width, device, True, True)
            model.forward_accumulate(x, probe)
            model.profile_begin(height, width, n_slices, 3)
            for _ in range(3):
                model.forward_accumulate(x, probe)
            torch.cuda.synchronize()
            cnt_h, ms_h = model.profile_collect(height, width, n_slices)
            head_all_ms = ms_h[-1] / max(cnt_h, 1)
            del probe
        finally:
            model.set_fuse_head(True)
        roofline['other_ms_per_forward']['head_fused_into'] = 'conv_cls.0 epilogue' if fused_head_ms < 0.5 * head_ms else None
        # the first conv kernel (csrc/rcu_first.hip) reads the NCHW input itself: no re-layout kernel in the timed region then
        roofline['other_ms_per_forward']['input_read_by'] = layers[0]['kernel'] if layers[0]['kernel'].startswith('conv3x3_first') else 'pack_input_kernel'
        roofline['aggregation'] = dict(bound='hbm', kernel='head_stream_kernel', achieved=head_bytes / head_ms / 1e6, peak=PEAK_HBM_GBS,
                                       unit='GB/s', frac=head_bytes / head_ms / 1e6 / PEAK_HBM_GBS, bytes_per_launch=head_bytes,
                                       avg_launch_ms=head_ms, measured='standalone launches outside the timed region')
        roofline['aggregation'].update(aggregation_kernels(device, n_slices, height, width))
        # the aggregation with every output tracked (188.7 MB per sample-volume = V*C*4 of logits + 2*S*4*V of statistics,
        # S = 5).  The head kernel never sees logits -- it reads the 32-channel feature map (4*V*32) -- and its planes are float64 (2*S*8*V):
        # `achieved` prices the ALGORITHMIC bytes against its time, `moved_*` the bytes it really moves.
        ao_alg = vox * (2 * 4 + 2 * 5 * 4.0)
        ao_moved = 4.0 * vox * 32 + 2 * 5 * 8.0 * vox
        agg_all = dict(bound='hbm', kernel='head_kernel (MI + variance statistics)', avg_launch_ms=head_all_ms,
                       bytes_per_launch=ao_alg, achieved=ao_alg / head_all_ms / 1e6, peak=PEAK_HBM_GBS, unit='GB/s',
                       frac=ao_alg / head_all_ms / 1e6 / PEAK_HBM_GBS, moved_bytes_per_launch=ao_moved,
                       moved_gbs=ao_moved / head_all_ms / 1e6, moved_frac=ao_moved / head_all_ms / 1e6 / PEAK_HBM_GBS,
                       measured='standalone launches outside the timed region; in the timed steps the update runs inside conv_cls.0')
        agg_all.update(aggregation_kernels(device, n_slices, height, width, all_outputs=True))
        roofline['aggregation_all_outputs'] = agg_all
    # HBM traffic of the dominant kernel: a builder-run PMC figure (profiles/pmc_traffic.json), reported only while the plan it was
    # measured on is the plan of this run (kernel per layer, shapes, samples per launch) -- a changed tile shape must not inherit it
    plan_hash = plan_fingerprint(layers, n_slices * g)
    roofline['plan_fingerprint'] = plan_hash
    pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json' if args.workload == 'brats' else 'pmc_traffic_{}.json'.format(args.workload))
    if os.path.exists(pmc_path):
        with open(pmc_path) as f:
            pmc = json.load(f)
        meta = pmc.get('_meta', {})
        if dominant in pmc and meta.get('plan_fingerprint') == plan_hash:
            roofline['traffic'] = pmc[dominant]
            roofline['traffic_source'] = (os.path.relpath(pmc_path, ROOT) + ': HBM bytes per launch from a builder-run rocprofv3 --pmc pass on the '
                                          'same plan (fingerprint {}; separate run, not measured in this process)'.format(plan_hash))
        elif dominant in pmc:
            roofline['traffic_source'] = ('none: ' + os.path.relpath(pmc_path, ROOT) + ' was measured on another plan (fingerprint {} there, {} here)'
                                          .format(meta.get('plan_fingerprint'), plan_hash))

    # ---- parity, outside the timed region, of the TIMED output (last step): ECE on the GPU maps vs the oracle on the same maps,
    # and -- the CPU leg -- the oracle's own forward passes on a slice sample of the same volume under the same dropout masks
    pred, p_fg = steps.prediction_and_foreground(out['probabilities'])
    ece_gpu = ev.ece_binary(p_fg, target_cpu, mask=mask_cpu)
    parity = dict(ece=ece_gpu, of='the output of the last timed step')
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import calib_oracle as co
        p_np = p_fg.cpu().numpy()
        ece_oracle = co.ece_binary(np.stack([1 - p_np, p_np], -1), target_cpu.numpy(), mask=mask_cpu.numpy())
        parity['ece_delta_same_maps'] = abs(ece_gpu - ece_oracle)
        parity['bin_ids_equal'] = bool(np.array_equal(ev.bin_ids(p_np), co.bin_ids(p_np.reshape(-1))))
    if not args.no_cpu_baseline:
        # the uncertainty-error counts (bnf_ue action: 11 thresholds, numpyfunctions.py:86-107) of the TIMED output, straight from the float32
        # probability map (rcu_unc_counts_from_p: "uncertain" looked up in the table of the reference's own float32 sets, fixture g20) against
        # the oracle's counts on ITS numpy entropy (ToEntropy, analysis.py:189-203): integer for integer -- no tie census any more
        from oracle import c_oracle
        thr_ue = list(ev.UE_THRESHOLDS)
        got = ev.uncertainty_counts_from_p(pred, target_cpu, p_fg, thresholds=thr_ue, mask=mask_cpu)[0].astype(np.int64)
        pred_np, tg_np, mk_np = pred.cpu().numpy(), target_cpu.numpy(), mask_cpu.numpy().astype(np.uint8)
        u_np = co.normalised_entropy(np.stack([1 - p_np, p_np], -1))
        ref_counts = c_oracle.unc_counts(u_np, pred_np, tg_np, mk_np, thr_ue).astype(np.int64)
        parity['ue_counts_equal'] = bool(np.array_equal(got, ref_counts))
        parity['ue_counts_of'] = 'rcu_unc_counts_from_p on the timed probability map vs the C oracle on the numpy entropy of the same map'
        parity['ue_max_count_delta'] = int(np.abs(got - ref_counts).max())
        parity['ue_voxels'] = int(np.count_nonzero(mk_np))
    if not args.no_cpu_baseline and world == 1 and not args.aleatoric:     # the CPU leg: rank 0 at N=1 only
        passes_total = T + (0 if (args.ensemble or args.no_ws) else 1)
        thread_counts, thread_probes = cpu_thread_counts(params, height, width)
        n_sel = cpu_probe_slices(model, params, x_cpu, passes_total, args.cpu_budget, thread_counts)
        sel = np.unique(np.linspace(0, n_slices - 1, n_sel).round().astype(np.int64))     # first, last and evenly between
        mask_sets_sel = None
        if not args.ensemble:
            mask_sets_sel = [split_masks(model, runner.masks_of(x, last_step, j), n_slices, torch.as_tensor(sel))
                             for j in range(1, T + 1)]
        cpu, ref = cpu_leg(members, params, x_cpu, sel, mask_sets_sel, not args.no_ws, bool(args.ensemble), args.cpu_budget, thread_counts)
        cpu['unit'] = 'member-{}s/s'.format(unit_name) if args.ensemble else 'MC-sample-{}s/s'.format(unit_name)
        cpu['thread_probes_s_per_slice'] = {str(c): ('did not finish a 4-slice forward in 40 s: skipped' if v is None else v)
                                            for c, v in thread_probes.items()}
        if isic:
            cpu['value'] *= n_slices                      # images, not batches
        idx = torch.as_tensor(sel, device=device)
        parity['slices_compared'] = [int(v) for v in sel]
        for key in ('probabilities', 'entropy', 'ws_probabilities'):
            if key in ref and key in out:
                parity['max_abs_d{}_vs_cpu'.format(key)] = float((out[key][idx].cpu() - ref[key]).abs().max())
        pg = out['probabilities'][idx][:, 1].cpu().numpy()
        pc = ref['probabilities'][:, 1].numpy()
        tg, mk = target_cpu[sel].numpy(), mask_cpu[sel].numpy()
        # the metric seam on the CPU: ECE + normalised entropy + the uncertainty-error counts of the bnf_ue action's 11
        # thresholds on the oracle's maps of the sample slices, timed, next to the forward passes
        t_c = time.perf_counter()
        pair = np.stack([1 - pc, pc], -1)
        ece_cpu = co.ece_binary(pair, tg, mask=mk)
        unc = co.normalised_entropy(pair)
        pred_c = (pc > 0.5).astype(np.uint8)
        for thr in [0.05] + [0.1 * k for k in range(1, 10)] + [0.95]:
            co.uncertainty_counts(pred_c, tg, unc > thr, mask=mk)
        cpu['metric_seam_s'] = time.perf_counter() - t_c
        cpu['metric_seam'] = 'ECE + normalised entropy + 11 uncertainty-error count passes (numpy, one thread) on {} of {} slices'.format(
            len(sel), n_slices)
        parity['ece_delta_vs_cpu'] = abs(ev.ece_binary(pg, tg, mask=mk) - ece_cpu)
        # the all-outputs leg ran its last volume under the same masks: its four maps against the same CPU passes
        for res, rec in ((out_ao, all_out), (out if args.all_outputs else None, parity)):
            if res is None:
                continue
            deltas = {key: float((res[key][idx].cpu() - ref[key]).abs().max())
                      for key in ('probabilities', 'entropy', 'mutual_info', 'variance') if key in res and key in ref}
            rec['max_abs_delta_vs_cpu'] = deltas
            rec['tolerance'] = 1e-4
            rec['within_tolerance'] = bool(deltas) and all(v <= 1e-4 for v in deltas.values())
            rec['slices_compared'] = [int(v) for v in sel]

    units = T * (n_slices if isic else 1)
    shape = '{}x{}x{}'.format(ISIC_CHANNELS, height, width) if isic else '4x{}x{}x{}'.format(n_slices, height, width)
    result = {
        'metric': ('ensemble-member-{}s/sec ({}, K={})'.format(unit_name, shape, T) if args.ensemble
                   else 'MC-sample-{}s/sec ({}, T={}{}{})'.format(unit_name, shape, T, ', sigma head' if args.aleatoric else '',
                                                                 ', all outputs' if args.all_outputs else '')),
        'value': units * args.steps / elapsed,
        'unit': 'member-{}s/s'.format(unit_name) if args.ensemble else 'MC-sample-{}s/s'.format(unit_name),
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'strong',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': ('ISIC {}: 2D U-Net(2,3,depth 4,start_filters 32,dropout 0.05) over a batch of {} images of 3x{}x{}, '
                                '{} + mean/entropy aggregation per step'
                                .format('ensemble' if args.ensemble else 'baseline_mc', n_slices, height, width,
                                        '{} members'.format(T) if args.ensemble else
                                        'T={} MC-dropout passes{}'.format(T, '' if args.no_ws else ' + weight-scaling pass'))) if isic else
                               ('BraTS ensemble: {} U-Net(2,4,depth 4,start_filters 32) members over {} slices of 4x{}x{} '
                                '+ mean/entropy aggregation per step'.format(T, n_slices, height, width)) if args.ensemble else
                               ('BraTS {}: 2D U-Net(2,4,depth 4,start_filters 32,dropout 0.05{}) over {} slices '
                                'of 4x{}x{}{}, T={} MC-dropout passes{} + mean/entropy aggregation per step'
                                .format('aleatoric + MC' if args.aleatoric else 'baseline_mc', ', sigma_out' if args.aleatoric else '',
                                        n_slices, height, width, ' (the reference\'s uncropped BraTS slices: padded levels)' if native else '',
                                        T, '' if args.no_ws else ' + weight-scaling pass')),
                   'T': T, 'ws_pass': not (args.no_ws or args.ensemble), 'slices': n_slices, 'height': height, 'width': width,
                   'pass_group': g, 'lanes': args.lanes, 'volumes_per_step': v_step, 'samples_per_launch': samples_per_launch,
                   # real extent -> allocated extent of every level (rcu_unet_options.pad_levels: a level that is not whole Winograd tiles is padded)
                   'level_extents': sorted({'{}x{} -> {}x{}'.format(L['height'], L['width'], L['grid_height'], L['grid_width'])
                                            for L in layers if not L['upsample']}, key=lambda t: -int(t.split('x')[0])),
                   'outputs': 'mean + entropy + mutual information + variance (float64 statistics)' if args.all_outputs else
                              'mean + entropy (MultiPredictionSummary() as every shipped script constructs it)',
                   'h2d': 'prefetched, inside timed region ({} MB per {} from pinned host memory on a copy stream, one event wait per '
                          '{}; rechun/dl/customsteps.py:20)'.format(feeder.bytes // 1000000, unit_name if not isic else 'batch',
                                                                    unit_name if not isic else 'batch'),
                   'sharding': sharding_record(runner, world, args, n_slices, height, width, v_step) if world > 1 else 'none',
                   'gflop_per_sample_{}'.format(unit_name): conv_flops / passes_run / 1e9 / (n_slices if isic else 1)},
        'n_ranks_seen': n_ranks_seen,
        'devices': devices_seen,
        'distinct_devices': len({(d_['uuid'], d_['pci'], d_['device_index']) for d_ in devices_seen}),
        'forwards_per_rank': forwards_per_rank,
        'device_allocs_in_timed_region': device_allocs_in_timed_region,   # hipMalloc calls of torch's allocator inside the timed steps: ~2.6 per step -- the summaries of all K volumes are held until the region ends (dropping them early: 16 instead of 103 calls in 40 steps, the same rate: 176.0-176.5 either way, round 5)
        'resident': dict(value=units * args.steps / elapsed_resident, ms_per_step=elapsed_resident / args.steps * 1e3, steps=args.steps,
                         note='the same steps with the volume already in HBM when the clock starts (no host-to-device copy): the '
                              'secondary figure; `value` has the prefetched copy inside'),
        'roofline': roofline,
        'calibration_kernels': calibration_kernels(device) if (world == 1 and not isic and not args.brief) else None,
        'cpu_baseline': cpu,
        'parity': parity,
        'all_outputs': all_out,
    }
    # ---- the reference's real BraTS shape as a sub-record of the default run (a child process: this process keeps its plans; the child has the
    # GPU to itself while rank 0 waits), with its own roofline and parity -- never in the headline's place
    if (args.workload == 'brats' and world == 1 and not (args.brief or args.ensemble or args.aleatoric or args.all_outputs or args.no_cpu_baseline)
            and os.environ.get('RCU_BENCH_NATIVE', '1') != '0'):
        result['native_shapes'] = native_sub_record(T, args.lanes)
    if force_pg:
        result['rccl_rehearsal'] = dict(backend=dist.get_backend(), world=dist.get_world_size(), ws_transport=runner.ws_transport,
                                        p2p_messages=runner.p2p_messages,
                                        note='RCU_BENCH_FORCE_PG=1: one-rank process group, every step through the exchange path')
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(result) + '\n').encode())
    if collective:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
