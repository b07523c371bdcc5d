"""Step seam: batch steps with the reference's call protocol, computing on librcu_hip.

Mirrors (names, arguments, output keys, error behaviour):
  BatchStep                     common/trainloop/steps.py:14-17
  SegmentationPredictStep       common/trainloop/steps.py:69-89
  McPredictStep                 rechun/dl/customsteps.py:10-39
  MultiPredictionSummary        rechun/dl/customsteps.py:42-71
  EnsemblePredictionStep        bin-dl/brats_test_ensemble.py:72-94
  AleatoricPredictStep          bin-dl/brats_test_aleatoric.py:51-73
  AleatoricMcPredictStep        extension (BASELINE config 'aleatoric + MC'): composition of the two above
  BatchContext / TaskContext    common/trainloop/context.py:334-355
A step is called as ``step(batch_context, task_context, context)``, reads
``batch_context.input['images']`` and writes torch tensors with the channel dim at 1 into
``batch_context.output``.

MI355X-first difference: by default the T (or K) probability volumes are never stacked in HBM.
``McPredictStep`` / ``EnsemblePredictionStep`` put a ``McStatistics`` object (per-voxel running
sums, updated by the fused forward+softmax+accumulate kernel) under ``multi_probabilities`` and
``MultiPredictionSummary`` finalises it.  ``materialize=True`` restores the reference behaviour
(a real ``[T, N, C, H, W]`` tensor), and the summary accepts either form.
"""
import abc
import logging

import torch

from . import _lib
from . import model as model_mod


class BatchContext:
    def __init__(self, batch: dict, batch_index: int, sample_offset: int = None) -> None:
        self.input = batch
        self.batch_index = batch_index
        # global index of the batch's first sample in the run's stream of slices / images (the test loop counts them; None: batch_index x the
        # batch's size): what the seeded Dropout2d masks of a stochastic step are keyed by -- a slice's MC sample must not depend on batch_size
        self.sample_offset = sample_offset
        self.output = {}
        self.metrics = {}
        self.score = None
        self.more = {}


class TaskContext:
    def __init__(self, epoch: int = 0, task_data=None, task_data_config=None) -> None:
        self.epoch = epoch
        self.data = task_data
        self.data_config = task_data_config
        self.scores = []
        self.more = {}


class Context:
    """Base of the contexts a step accepts (reference: ctx.TorchTrainContext / ctx.TorchTestContext)."""


class TorchTestContext(Context):
    """The two attributes a step touches: ``model`` and ``device`` (common/trainloop/context.py:256-322).
    The reference hard-codes 'cuda' in its scripts (bin-dl/brats_test_default.py:39)."""

    def __init__(self, device_str: str = 'cuda', model=None) -> None:
        self.device = torch.device(device_str)
        self.model = model
        self.more = {}


def _type_error_msg(obj, *expected):
    names = ','.join(e.__name__ for e in expected)
    return 'expected type is "({})" but object is of type "{}"'.format(names, obj.__class__.__name__)


def _check_context(context):
    # The reference means to raise ValueError here (customsteps.py:17-18); its message helper trips over the
    # tuple argument first (common/utils/messages.py:5) -- the intended ValueError is what we raise.
    if not isinstance(context, Context):
        raise ValueError(_type_error_msg(context, TorchTestContext))


def set_dropout_mode(model, is_train=True):
    """common/utils/torchhelper.py:44-50: toggles only the Dropout modules; BatchNorm stays in eval."""
    # (model.UNet keeps the list of its Dropout2d modules -- its module tree is fixed after construction: no walk over 150 modules four
    # times per batch)
    sites = getattr(model, '_site_modules', None) if isinstance(model, model_mod.UNet) else None
    for m in (model.modules() if sites is None else sites):
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d, torch.nn.Dropout3d)):
            if is_train:
                m.train()
            else:
                m.eval()


def job_seed(seed, step_index, job):
    """Seed of the dropout masks of MC pass ``job`` (1..T) of batch / volume ``step_index``: a function of (seed, batch, pass) only -- not
    of the pass groups, the stream lanes, the rank that runs the pass or the number of ranks."""
    return (int(seed) * 1000003 + int(step_index) * 10007 + int(job)) % (2 ** 63 - 1)


def pass_seed(seed, job):
    """Key of the library's counter-based mask draw (include/rcu.h, rcu_dropout_masks) for MC pass ``job`` (1..T): a function of (seed, pass) --
    the draw's counter carries the sample's GLOBAL index, so a slice's masks do not depend on the batch it is loaded in either."""
    return job_seed(seed, 0, job)


def first_sample_of(batch_context, n):
    """Global index of a batch's first sample: what the test loop counted (``BatchContext.sample_offset``), else batch_index x n."""
    offset = getattr(batch_context, 'sample_offset', None)
    return int(batch_context.batch_index) * int(n) if offset is None else int(offset)


class McStatistics:
    """Per-voxel sufficient statistics of the passes seen so far (the ``stats`` blob of include/rcu.h).
    Plain additive: blobs of disjoint pass subsets merge by ``+`` (one RCCL sum-reduce, see
    rcu_amd.distributed).  ``exact`` (RCU_MC_EXACT; what the predict steps use): float64 planes whose addends are rounded to multiples
    of 2^-40 first -- every addition is then exact, so the sums (and everything finalised from them) carry the same bits whatever the order of
    the passes, the pass groups, the stream lanes, the ranks and the collective's reduction tree."""

    def __init__(self, n, nb_classes, height, width, device, do_mi=False, do_var=False, blob=None, exact=False):
        self.n, self.nb_classes, self.height, self.width = n, nb_classes, height, width
        self.hw = height * width
        self.flags = self.flags_of(do_mi, do_var, exact)
        elems = self.blob_elements(n, nb_classes, self.hw, do_mi, do_var, exact)
        dtype = self.dtype_of(do_var, exact)
        if blob is None:
            blob = torch.empty(elems, device=device, dtype=dtype)
        elif blob.numel() != elems or blob.dtype != dtype or not blob.is_contiguous():
            raise ValueError('statistics blob must be a contiguous {} tensor of {} elements'.format(dtype, elems))
        self.blob = blob
        self.count = 0
        # set by the predict steps: recipe(do_mi, do_var, materialize=False) runs the SAME passes again (same inputs, same dropout
        # masks) into fresh statistics with other flags, or into a real [T, N, C, H, W] tensor
        self.recipe = None
        _lib.check(_lib.load().rcu_mc_begin(_lib.ptr(self.blob), n, self.hw, nb_classes, self.flags,
                                            _lib.current_stream()))

    @staticmethod
    def flags_of(do_mi=False, do_var=False, exact=False):
        return (_lib.RCU_MC_MI if do_mi else 0) | (_lib.RCU_MC_VAR if do_var else 0) | (_lib.RCU_MC_EXACT if exact else 0)

    @staticmethod
    def dtype_of(do_var=False, exact=False):
        return torch.float64 if (do_var or exact) else torch.float32

    @staticmethod
    def blob_elements(n, nb_classes, hw, do_mi=False, do_var=False, exact=False):
        flags = McStatistics.flags_of(do_mi, do_var, exact)
        return _lib.load().rcu_mc_stats_bytes(n, hw, nb_classes, flags) // (8 if (do_var or exact) else 4)

    @property
    def exact(self):
        return bool(self.flags & _lib.RCU_MC_EXACT)

    @property
    def do_mi(self):
        return bool(self.flags & _lib.RCU_MC_MI)

    @property
    def do_var(self):
        return bool(self.flags & _lib.RCU_MC_VAR)

    def accumulate(self, tensor, is_probabilities=False):
        """Add one ``[N, C, H, W]`` logits (softmax applied on the fly) or probability volume."""
        t = tensor.to(torch.float32).contiguous()
        if tuple(t.shape) != (self.n, self.nb_classes, self.height, self.width):
            raise ValueError('expected shape {}'.format((self.n, self.nb_classes, self.height, self.width)))
        flags = self.flags | (_lib.RCU_MC_INPUT_PROBS if is_probabilities else 0)
        _lib.check(_lib.load().rcu_mc_accumulate(_lib.ptr(t), _lib.ptr(self.blob), self.n, self.hw, self.nb_classes,
                                                 flags, _lib.current_stream()))
        self.count += 1

    def as_tensor(self):
        """The stacked ``[T, N, C, H, W]`` probabilities the reference keeps under ``multi_probabilities`` (customsteps.py:36): the
        statistics do not hold them, so the passes run again, materialised, under the same masks.  For foreign steps that read
        the key between the predict step and the summary; costs T forward passes and T volumes of HBM."""
        if self.recipe is None:
            raise ValueError('these statistics were not produced by a predict step: the passes cannot be replayed')
        return self.recipe(self.do_mi, self.do_var, materialize=True)

    def finalize(self, do_mi=False, do_var=False, count=None):
        """-> dict with the reference's keys: probabilities, entropy[, mutual_info][, variance]."""
        if do_mi and not self.do_mi:
            raise ValueError('mutual information was not tracked (construct the predict step with do_mi=True)')
        if do_var and not self.do_var:
            raise ValueError('variance was not tracked (construct the predict step with do_var=True)')
        t = self.count if count is None else count
        dev = self.blob.device
        shape1 = (self.n, 1, self.height, self.width)
        mean = torch.empty((self.n, self.nb_classes, self.height, self.width), device=dev, dtype=torch.float32)
        entropy = torch.empty(shape1, device=dev, dtype=torch.float32)
        mi = torch.empty(shape1, device=dev, dtype=torch.float32) if do_mi else None
        var = torch.empty(shape1, device=dev, dtype=torch.float32) if do_var else None
        _lib.check(_lib.load().rcu_mc_finalize(_lib.ptr(self.blob), self.n, self.hw, self.nb_classes, int(t), self.flags,
                                               _lib.ptr(mean), _lib.ptr(entropy), _lib.ptr(mi), _lib.ptr(var),
                                               _lib.current_stream()))
        out = {'probabilities': mean, 'entropy': entropy}
        if do_mi:
            out['mutual_info'] = mi
        if do_var:
            out['variance'] = var
        return out


class StreamLanes:
    """The launches of one batch spread over ``count`` HIP streams in turn (lane 0 = the caller's stream).  Consecutive layers of a
    forward pass depend on each other, so a stream idles while a layer's last workgroups finish and the next layer's first ones
    start; MC passes / ensemble members are independent, and a second lane fills those gaps (tools/stream_overlap_probe.py: 6.63 ->
    6.27 ms per pass on the 160-slice volume).  Every lane needs an activation workspace (``lane=`` of UNet.forward_accumulate) and a
    statistics blob of its own; blobs are plain sums, so the side lanes' are added into lane 0's at the end.  The assignment launch ->
    lane is fixed (round robin), so the result does not depend on timing.

        lanes = StreamLanes(device, 2)
        lane_stats = lanes.begin(stats, make_side_stats, inputs=(images,))
        for each launch: lanes.run(lambda st, lane: model.forward_accumulate(images, st, masks, lane=lane))
        lanes.end(merge)            # merge(stats, side_stats) on the caller's stream
    """

    _side = {}       # (device index, lane) -> stream: side streams are shared by all steps of the process

    def __init__(self, device, count):
        device = torch.device(device)
        self.count = max(1, int(count)) if device.type == 'cuda' else 1
        self.device = device
        self.streams = []
        for lane in range(1, self.count):
            key = (device.index if device.index is not None else torch.cuda.current_device(), lane)
            if key not in StreamLanes._side:
                StreamLanes._side[key] = torch.cuda.Stream(device=device)
            self.streams.append(StreamLanes._side[key])
        self._stats, self._launch, self._current = [], 0, None

    def begin(self, stats, make_side_stats, inputs=(), first=0):
        """``first``: lane of the first launch (a runner that gives a rank only two or three launches per volume rotates it from
        volume to volume, so that the lanes carry the same load over a stream of volumes)."""
        self._stats, self._launch = [stats], int(first) % self.count
        if self.count > 1:
            self._current = torch.cuda.current_stream(self.device)
            for side in self.streams:
                side.wait_stream(self._current)          # the inputs (and whatever else the caller prepared) are ready
                for t in inputs:
                    t.record_stream(side)
                with torch.cuda.stream(side):
                    self._stats.append(make_side_stats())     # zeroed on its own stream
        return self._stats

    def run(self, launch, lane=None):
        """launch(statistics of the lane, lane index) on the next lane (or on ``lane``)."""
        if lane is None:
            lane = self._launch % self.count
            self._launch += 1
        if lane == 0:
            launch(self._stats[0], 0)
        else:
            with torch.cuda.stream(self.streams[lane - 1]):
                launch(self._stats[lane], lane)

    def end(self, merge):
        for side, st in zip(self.streams, self._stats[1:]):
            self._current.wait_stream(side)
            merge(self._stats[0], st)                    # on the caller's stream, behind the lane's last launch
            for t in vars(st).values():
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(self._current)
        self._stats = self._stats[:1]


def balanced_groups(count, group, lanes):
    """Sizes of the pass groups ``count`` MC passes run in, at most ``group`` passes each, for launches that take ``lanes`` stream lanes in
    turn: rounds of one group per lane, every group of a round the same size, the last rounds smaller instead of a full group for one
    lane and nothing for the other -- T = 20 in groups of 4 on two lanes is 4 4 | 4 4 | 2 2 (10 passes per lane), not 4 4 | 4 4 | 4
    (12 against 8: the lanes fill each other's gaps only while both have work).  One lane, or a group size that divides count / lanes:
    plain groups of ``group``."""
    group, lanes, left, sizes = max(1, int(group)), max(1, int(lanes)), int(count), []
    while left > 0:
        for k in range(lanes):         # one round: lane k takes its share of what the lanes from k on still have to run
            if left <= 0:
                break
            sizes.append(min(group, -(-left // (lanes - k))))
            left -= sizes[-1]
    return sizes


def pass_group_size(model, n, h, w, group_pixels):
    """MC passes per launch for batches of n images of h x w: ``group_pixels`` worth of pixels, and no tensor of the plan beyond the 2 GB
    the Winograd kernels address (model.UNet.max_group_samples)."""
    g = max(1, int(group_pixels) // (n * h * w))
    cap = getattr(model, 'max_group_samples', None)
    return g if cap is None else max(1, min(g, cap(h, w) // n))


def reserve_canonical_plans(model, n, h, w, mc_steps, group, lanes):
    """Plans (and workspaces) of ``model`` for batches of n images of h x w whose T = ``mc_steps`` passes run up to ``group`` per launch, on
    ``lanes`` stream lanes: all sized for n * min(group, T) samples, whatever launches this process will really make."""
    plan = n * max(1, min(int(group), max(int(mc_steps), 1)))
    for lane in range(max(1, int(lanes))):
        model.reserve(h, w, plan, lane)
    return plan


def merge_statistics(stats, side):
    """Add the statistics of a side lane into ``stats`` (plain sums, include/rcu.h rcu_mc_*)."""
    stats.blob.add_(side.blob)
    stats.count += side.count
    if getattr(side, 'sigma_sum', None) is not None:
        stats.sigma_sum.add_(side.sigma_sum)


def softmax(logits, out=None):
    """F.softmax(logits, 1) on the HIP path (``out``: a contiguous float32 tensor of the same shape to write into)."""
    logits = logits.to(torch.float32).contiguous()
    n, c, h, w = logits.shape
    if out is None:
        out = torch.empty_like(logits)
    elif out.shape != logits.shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != logits.device:
        raise ValueError('out must be a contiguous float32 tensor of the shape and device of the logits')
    _lib.check(_lib.load().rcu_softmax(_lib.ptr(logits), _lib.ptr(out), n, h * w, c, _lib.current_stream()))
    return out


class BatchStep(abc.ABC):
    @abc.abstractmethod
    def __call__(self, batch_context: BatchContext, task_context: TaskContext, context: Context) -> None:
        pass


def _images_to_device(batch_context, context):
    # non_blocking: a no-op for pageable host memory, asynchronous when the loader pinned the batch (rcu_amd.loops.prefetch)
    images = batch_context.input['images'].float().to(context.device, non_blocking=True)
    if not images.is_contiguous():
        # the volume loader hands the file's channel-last order over (data.VolumeDataset.__getitems__; the copy keeps the strides): the
        # re-ordering to [N, C, H, W] is a kernel behind the upload, not a strided copy on the loader thread
        images = images.contiguous()
    batch_context.input['images'] = images
    return images


class SegmentationPredictStep(BatchStep):

    def __init__(self, has_labels=False, do_probs=False) -> None:
        super().__init__()
        self.has_labels = has_labels
        self.do_probs = do_probs

    def __call__(self, batch_context, task_context, context) -> None:
        _check_context(context)
        images = _images_to_device(batch_context, context)
        if self.has_labels:
            batch_context.input['labels'] = batch_context.input['labels'].long().to(context.device)
        logits = context.model(images)
        batch_context.output['logits'] = logits
        if self.do_probs:
            batch_context.output['probabilities'] = softmax(logits)


class McPredictStep(BatchStep):
    """T stochastic passes (plus the deterministic 'weight scaling' pass the reference always runs
    first, customsteps.py:22-25).
    ``seed``: the Dropout2d factors of pass j for the slice with global index g (``batch_context.sample_offset`` + its position in the batch: the
    test loop counts the slices it has handed out) are drawn under the key ``pass_seed(seed, j)`` at the counter of (g, site, channel)
    (UNet.seeded_masks: one kernel per launch, include/rcu.h rcu_dropout_masks) instead of from the device's default generator: the T samples of a
    SLICE are then a function of (seed, slice, pass) alone -- the same whatever ``batch_size`` the YAML file sets and however the loop coalesces
    batches (round 6; rounds 1-5 keyed the draw by the batch), whatever the pass groups and stream lanes, and the same when the passes are
    sharded over several GPUs (rcu_amd.distributed.ShardedMcPredictStep).  The drop-in scripts pass the YAML file's ``seed``.
    ``exact`` (default): the statistics are exact sums (McStatistics), so that ``MultiPredictionSummary``'s outputs do not depend on
    how the passes were grouped, laned or sharded either; ``exact=False`` keeps float32 sums (float64 with ``do_var``)."""

    # A forward pass fills the GPU from about 160 BraTS slices (3.9 M pixels) on, and the deep levels of the U-Net -- few, long
    # work items per launch -- only from two to four times that (their last round of workgroups is 75-88 % full at 160 slices, 94 % at
    # 320, full at 640); the shipped configs use batch_size 32.  The T passes of a batch are independent, so the fused path runs them in
    # groups of g = GROUP_PIXELS // (N*H*W) as one batch of N * g samples (include/rcu.h: rcu_unet_forward_accumulate_passes) -- same
    # statistics bit for bit; the workspace grows to that of a 640-slice batch (12 GB per lane of the 288), not beyond, and
    # pass_group_size keeps every tensor below the 2 GB the kernels' 32-bit buffer offsets reach (640 BraTS slices x 32 channels: 2.01e9 bytes).
    GROUP_PIXELS = 4 * 160 * 192 * 128
    LANES = 2      # HIP streams the pass groups of a batch alternate over (StreamLanes); the ``lanes`` argument overrides it

    def __init__(self, mc_steps, do_mi=False, do_var=False, materialize=False, masks=None, ws_pass=True,
                 group_pixels=None, lanes=None, seed=None, exact=True) -> None:
        super().__init__()
        self.mc_steps = mc_steps
        self.do_mi, self.do_var = do_mi, do_var
        self.materialize = materialize
        self.masks = masks          # optional: list (one per pass) of mask sets to inject instead of sampling
        self.ws_pass = ws_pass
        self.group_pixels = self.GROUP_PIXELS if group_pixels is None else group_pixels
        self.lanes = self.LANES if lanes is None else max(1, int(lanes))
        self.seed = seed
        self.exact = bool(exact) and mc_steps <= _lib.RCU_MC_EXACT_MAX_PASSES      # (beyond the exact form's 2,048 passes: plain float sums)

    def _seeded_masks(self, model, images, first_sample, job):
        """The mask tensor of MC pass ``job`` (1..T) of the batch whose first sample has global index ``first_sample`` under ``self.seed``
        (dropout mode is on)."""
        return model.seeded_masks(images.shape[0], images.device, [pass_seed(self.seed, job)], first_sample)

    def _launch_masks(self, model, images, first_sample, first, count):
        """``masks`` argument of the launch that runs the passes first .. first + count - 1 (0-based): injected sets, seeded draws, or
        None = drawn inside the launch from the device's default generator."""
        if self.masks is not None:
            return self.masks[first] if count == 1 else self.masks[first:first + count]
        if self.seed is None:
            return None
        # one kernel for the launch's passes, on the stream that reads them, already in the launch's layout (UNet.seeded_masks)
        return model.seeded_masks(images.shape[0], images.device, [pass_seed(self.seed, j + 1) for j in range(first, first + count)], first_sample)

    def __call__(self, batch_context, task_context, context) -> None:
        _check_context(context)
        images = _images_to_device(batch_context, context)
        model = context.model
        k = first_sample_of(batch_context, images.shape[0])      # global index of the batch's first sample: what the seeded masks are keyed by

        if isinstance(model, model_mod.UNet):
            # The CANONICAL plan of the batch, on every lane, before the first forward: which kernel a layer gets depends on the batch its plan
            # is sized for, and a pass's bits are a property of the plan -- so every way of running the T passes (any grouping, one lane or
            # two, fused or materialised, one process or the ranks of rcu_amd.distributed) sizes its plans for the same n * min(group, T).
            n, _, h, w = images.shape
            reserve_canonical_plans(model, n, h, w, self.mc_steps, pass_group_size(model, n, h, w, self.group_pixels),
                                    1 if self.materialize else min(self.lanes, max(self.mc_steps, 1)))
        fused = not self.materialize and isinstance(model, model_mod.UNet)
        if self.ws_pass and not fused:
            batch_context.output['ws_probabilities'] = softmax(model(images))

        def ws_pass():      # fused path: issued on lane 0 once the lanes are set up, so that the side lane starts its first group beside it
            set_dropout_mode(model, is_train=False)
            batch_context.output['ws_probabilities'] = softmax(model(images))
            set_dropout_mode(model, is_train=True)

        set_dropout_mode(model, is_train=True)
        try:
            if not fused:
                probs = []
                for i in range(self.mc_steps):
                    masks = self._launch_masks(model, images, k, i, 1) if isinstance(model, model_mod.UNet) else None
                    logits = model(images) if masks is None else model(images, masks)
                    probs.append(softmax(logits))
                batch_context.output['multi_probabilities'] = torch.stack(probs)
            else:
                batch_context.output['multi_probabilities'] = self._fused_passes(model, images, self.do_mi, self.do_var, k,
                                                                                 before=ws_pass if self.ws_pass else None)
        finally:
            set_dropout_mode(model, is_train=False)   # reset to eval for the next batch (customsteps.py:39)

    def _fused_passes(self, model, images, do_mi, do_var, first_sample=0, before=None):
        """The T passes into per-voxel statistics (dropout mode is on).  The statistics carry a recipe that replays the passes
        -- same images, same masks (seeded: the same draws again; unseeded: the device generator is put back to where the sampling
        started) -- so that ``MultiPredictionSummary(do_mi / do_var)`` decides alone which outputs exist, as in the reference
        (customsteps.py:44-48)."""
        n, _, h, w = images.shape
        dev = images.device
        unseeded = self.masks is None and self.seed is None
        rng_state = torch.cuda.get_rng_state(dev) if (unseeded and dev.type == 'cuda') else None

        def fresh():
            return McStatistics(n, model.nb_classes, h, w, dev, do_mi, do_var, exact=self.exact)

        stats = fresh()
        group = pass_group_size(model, n, h, w, self.group_pixels)
        # (every lane gets work whenever there are two passes: T = 20 on batches of 32 slices is 10 | 10 on two lanes, not one launch of 20 on one)
        lanes = StreamLanes(dev, min(self.lanes, max(self.mc_steps, 1)))
        sizes = balanced_groups(self.mc_steps, group, lanes.count)
        lanes.begin(stats, fresh, inputs=(images,))
        if before is not None:
            before()                   # (the weight-scaling pass, on the caller's stream)
        i = 0
        for g in sizes:
            # the masks are drawn INSIDE the launch, i.e. on the lane's stream -- the stream that reads them (seeded: from the step's generator;
            # unseeded: by forward_accumulate from the device's, in launch order whatever the lane)
            lanes.run(lambda st, lane, i=i, g=g: model.forward_accumulate(images, st, self._launch_masks(model, images, first_sample, i, g),
                                                                       passes=g, lane=lane))
            i += g
        lanes.end(merge_statistics)

        def recipe(mi, var, materialize=False):
            now = torch.cuda.get_rng_state(dev) if rng_state is not None else None
            if rng_state is not None:
                torch.cuda.set_rng_state(rng_state, dev)
            set_dropout_mode(model, is_train=True)
            try:
                if not materialize:
                    return self._fused_passes(model, images, mi, var, first_sample)
                probs = []
                t = 0
                for g in sizes:                      # the same draws as the fused path makes: one per pass group
                    if not unseeded:
                        sets = [self._launch_masks(model, images, first_sample, j, 1) for j in range(t, t + g)]
                    elif g == 1:
                        sets = [None]
                    else:                            # rows [site][pass * n + i]: split the group's draw into its passes
                        flat = model.sample_masks(n * g, dev)
                        per_site = torch.split(flat, [n * g * c for _, c in model.dropout_sites()])
                        sets = [[ps.view(g, n, -1)[k] for ps in per_site] for k in range(g)]
                    for ms in sets:
                        probs.append(softmax(model(images) if ms is None else model(images, ms)))
                    t += g
                return torch.stack(probs)
            finally:
                set_dropout_mode(model, is_train=False)
                if now is not None:
                    torch.cuda.set_rng_state(now, dev)

        stats.recipe = recipe
        return stats


def share_member_workspaces(members):
    """Members 2..K of an ensemble borrow the activation workspaces of the first (model.UNet.share_workspace): the K models the
    reference keeps resident (bin-dl/brats_test_ensemble.py:44-57) differ in their weights only.  Members of another architecture,
    foreign modules and members that already borrow are left alone."""
    donor = members[0] if members and isinstance(members[0], model_mod.UNet) else None
    if donor is None:
        return
    if donor._donor is not None:
        donor = donor._donor         # the first member borrows already (the same models in another order): its owner lends to the rest
    for m in members[1:]:
        if isinstance(m, model_mod.UNet) and m is not donor and m._donor is None:
            try:
                m.share_workspace(donor)
            except ValueError:
                pass        # a different architecture keeps its own workspace


class EnsemblePredictionStep(BatchStep):
    """context.model plus ``additional_models``, all in eval mode (brats_test_ensemble.py:78-94)."""

    def __init__(self, additional_models, do_mi=False, do_var=False, materialize=False, share_workspace=True, lanes=None,
                 exact=True) -> None:
        super().__init__()
        self.additional_models = additional_models
        self.do_mi, self.do_var = do_mi, do_var
        self.materialize = materialize
        self.share_workspace = share_workspace
        self.lanes = McPredictStep.LANES if lanes is None else max(1, int(lanes))
        self.exact = bool(exact)       # exact sums (McStatistics): the member order / lanes / ranks do not change the bits

    def __call__(self, batch_context, task_context, context) -> None:
        _check_context(context)
        images = _images_to_device(batch_context, context)
        members = [context.model] + list(self.additional_models)
        fused = not self.materialize and all(isinstance(m, model_mod.UNet) for m in members)
        if self.share_workspace:
            share_member_workspaces(members)
        if fused:
            n, _, h, w = images.shape
            def run(mi, var, materialize=False):
                if materialize:
                    return torch.stack([softmax(m(images)) for m in members])
                exact = self.exact and len(members) <= _lib.RCU_MC_EXACT_MAX_PASSES
                st = McStatistics(n, members[0].nb_classes, h, w, images.device, mi, var, exact=exact)
                lanes = StreamLanes(images.device, min(self.lanes, len(members)))
                lanes.begin(st, lambda: McStatistics(n, members[0].nb_classes, h, w, images.device, mi, var, exact=exact), inputs=(images,))
                for m in members:      # (a member keeps to its lane from batch to batch: one workspace per member)
                    lanes.run(lambda s_, lane, m=m: m.forward_accumulate(images, s_, lane=lane))
                lanes.end(merge_statistics)
                st.recipe = run
                return st
            batch_context.output['multi_probabilities'] = run(self.do_mi, self.do_var)
        else:
            batch_context.output['multi_probabilities'] = torch.stack([softmax(m(images)) for m in members])


class MultiPredictionSummary(BatchStep):
    _replay_warned = False

    def __init__(self, do_mi=False, do_var=False, remove_multi_probs=True) -> None:
        super().__init__()
        self.do_mi = do_mi
        self.do_var = do_var
        self.remove_multi_probs = remove_multi_probs

    def __call__(self, batch_context, task_context, context) -> None:
        if self.remove_multi_probs:
            multi = batch_context.output.pop('multi_probabilities')
        else:
            multi = batch_context.output['multi_probabilities']
        if multi is None:      # a rank other than the root of a sharded predict step (rcu_amd.distributed): the root alone has the merged statistics
            return
        if hasattr(multi, 'finalize_when_merged'):
            # the root of a sharded predict step: the statistics are merged by a collective that is still in flight.  The finalize runs on a
            # side stream behind it; the outputs carry an event (wait_for_outputs) instead of holding the compute stream up
            out, event = multi.finalize_when_merged(self.do_mi, self.do_var)
            batch_context.output.update(out)
            if event is not None:
                batch_context.more['outputs_ready'] = event
            return
        if isinstance(multi, McStatistics):
            stats = multi
            if (self.do_mi and not stats.do_mi) or (self.do_var and not stats.do_var):
                # the predict step did not track what this summary asks for (the reference's summary alone decides,
                # customsteps.py:44-48): replay the passes with the flags of both
                if stats.recipe is None:
                    raise ValueError('the statistics lack {} and cannot be replayed'.format(
                        'the entropy sum (do_mi)' if self.do_mi and not stats.do_mi else 'the squared sums (do_var)'))
                if not MultiPredictionSummary._replay_warned:
                    MultiPredictionSummary._replay_warned = True
                    logging.getLogger(__name__).warning(
                        'MultiPredictionSummary(do_mi=%s, do_var=%s) asks for more than the predict step tracked: the passes run a '
                        'second time under the same masks (construct the predict step with the same flags to avoid it)',
                        self.do_mi, self.do_var)
                stats = stats.recipe(self.do_mi or stats.do_mi, self.do_var or stats.do_var)
        else:
            t, n, c, h, w = multi.shape
            # (exact sums: a materialised stack gives the bits the fused statistics of the same passes give)
            stats = McStatistics(n, c, h, w, multi.device, self.do_mi, self.do_var, exact=t <= _lib.RCU_MC_EXACT_MAX_PASSES)
            for i in range(t):
                stats.accumulate(multi[i], is_probabilities=True)
        out = stats.finalize(self.do_mi, self.do_var)
        batch_context.output['probabilities'] = out['probabilities']
        batch_context.output['entropy'] = out['entropy']
        if self.do_mi:
            batch_context.output['mutual_info'] = out['mutual_info']
        if self.do_var:
            batch_context.output['variance'] = out['variance']


class AleatoricPredictStep(BatchStep):

    def __init__(self, is_log_sigma=False) -> None:
        super().__init__()
        self.is_log_sigma = is_log_sigma

    def __call__(self, batch_context, task_context, context) -> None:
        _check_context(context)
        images = _images_to_device(batch_context, context)
        mean_logits, sigma_raw = context.model(images)
        batch_context.output['logits'] = mean_logits
        n, c, h, w = mean_logits.shape
        probs = torch.empty_like(mean_logits)
        sigma = torch.empty_like(mean_logits)
        _lib.check(_lib.load().rcu_aleatoric(_lib.ptr(mean_logits), _lib.ptr(sigma_raw.contiguous()), n, h * w, c,
                                             int(self.is_log_sigma), _lib.ptr(probs), _lib.ptr(sigma), None, None,
                                             _lib.current_stream()))
        batch_context.output['sigma'] = sigma
        batch_context.output['probabilities'] = probs


class AleatoricMcPredictStep(BatchStep):
    """EXTENSION -- BASELINE config "aleatoric + MC" (sigma-head U-Net, T stochastic passes); the reference has no such step
    (McPredictStep cannot take the (logits, sigma) tuple, customsteps.py:32-33).  It is the composition of the reference's
    pieces: per pass t, with dropout on, ``logits_t, raw_t = model(x)``; ``p_t = softmax(logits_t)`` goes into the MC statistics
    (-> ``multi_probabilities`` for MultiPredictionSummary, as McPredictStep); ``sigma_t = |raw_t|`` or ``exp(raw_t)``
    (AleatoricPredictStep, brats_test_aleatoric.py:66-69) is averaged over the passes -> ``sigma`` [N, C, H, W].  The
    deterministic pass that McPredictStep runs first gives ``ws_probabilities`` and ``ws_sigma``."""

    def __init__(self, mc_steps, is_log_sigma=False, do_mi=False, do_var=False, masks=None, ws_pass=True, lanes=None, exact=True) -> None:
        super().__init__()
        self.mc_steps = mc_steps
        self.is_log_sigma = is_log_sigma
        self.do_mi, self.do_var = do_mi, do_var
        self.masks = masks
        self.ws_pass = ws_pass
        self.lanes = McPredictStep.LANES if lanes is None else max(1, int(lanes))
        self.exact = bool(exact) and mc_steps <= _lib.RCU_MC_EXACT_MAX_PASSES       # the probability statistics; the sigma sums stay float32 (unbounded addends)

    def __call__(self, batch_context, task_context, context) -> None:
        _check_context(context)
        images = _images_to_device(batch_context, context)
        model = context.model
        if not isinstance(model, model_mod.UNet) or not model.sigma_out:
            raise ValueError('AleatoricMcPredictStep needs a rcu_amd.model.UNet built with sigma_out=True')
        n, _, h, w = images.shape
        c = model.nb_classes
        if self.ws_pass:
            logits, sigma_raw = model(images)
            probs = torch.empty_like(logits)
            sigma = torch.empty_like(logits)
            _lib.check(_lib.load().rcu_aleatoric(_lib.ptr(logits), _lib.ptr(sigma_raw.contiguous()), n, h * w, c,
                                                 int(self.is_log_sigma), _lib.ptr(probs), _lib.ptr(sigma), None, None,
                                                 _lib.current_stream()))
            batch_context.output['ws_probabilities'] = probs
            batch_context.output['ws_sigma'] = sigma
        set_dropout_mode(model, is_train=True)
        try:
            dev = images.device

            def fresh():
                st = McStatistics(n, c, h, w, dev, self.do_mi, self.do_var, exact=self.exact)
                st.sigma_sum = torch.zeros((n, c, h, w), device=dev, dtype=torch.float32)
                return st

            stats = fresh()
            sigma_sum = stats.sigma_sum
            # pass groups and stream lanes as in McPredictStep: g passes per launch, launches alternating over two HIP streams
            group = pass_group_size(model, n, h, w, McPredictStep.GROUP_PIXELS)
            lanes = StreamLanes(dev, min(self.lanes, self.mc_steps))
            lanes.begin(stats, fresh, inputs=(images,))
            i = 0
            for g in balanced_groups(self.mc_steps, group, lanes.count):
                masks = None if self.masks is None else (self.masks[i] if g == 1 else self.masks[i:i + g])
                lanes.run(lambda st, lane, masks=masks, g=g: model.forward_accumulate_sigma(images, st, st.sigma_sum, masks, self.is_log_sigma,
                                                                                         lane=lane, passes=g))
                i += g
            lanes.end(merge_statistics)
            batch_context.output['multi_probabilities'] = stats
            batch_context.output['sigma'] = sigma_sum.div_(float(max(self.mc_steps, 1)))
        finally:
            set_dropout_mode(model, is_train=False)


def wait_for_outputs(batch_context):
    """Order the current stream behind outputs a step produced on a stream of its own (``batch_context.more['outputs_ready']``, set by
    MultiPredictionSummary for the root of a sharded predict step).  Steps and hooks that read ``batch_context.output`` tensors on the
    compute stream call this first; the test loop's download waits for the same event on its own stream."""
    event = batch_context.more.get('outputs_ready')
    if event is not None:
        torch.cuda.current_stream().wait_event(event)
        for value in batch_context.output.values():
            if torch.is_tensor(value) and value.is_cuda:
                value.record_stream(torch.cuda.current_stream())


def prediction_and_foreground(probabilities):
    """``[N, C, H, W]`` probabilities -> (uint8 argmax ``[N, H, W]``, float32 foreground probability
    ``[N, H, W]``): what the reference's writers derive with numpy before saving
    ``*_prediction`` / ``*_probabilities`` (bin-dl/brats_test_default.py:96-99)."""
    p = probabilities.to(torch.float32).contiguous()
    n, c, h, w = p.shape
    pred = torch.empty((n, h, w), device=p.device, dtype=torch.uint8)
    fg = torch.empty((n, h, w), device=p.device, dtype=torch.float32)
    _lib.check(_lib.load().rcu_prediction_and_foreground(_lib.ptr(p), n, h * w, c, _lib.ptr(pred), _lib.ptr(fg),
                                                         _lib.current_stream()))
    return pred, fg


def sigma_of_prediction(logits, sigma_raw, is_log_sigma=False):
    """sigma of the predicted class per voxel (bin-dl/brats_test_aleatoric.py:95-97) -> (prediction u8, sigma f32)."""
    logits = logits.to(torch.float32).contiguous()
    sigma_raw = sigma_raw.to(torch.float32).contiguous()
    n, c, h, w = logits.shape
    pred = torch.empty((n, h, w), device=logits.device, dtype=torch.uint8)
    sp = torch.empty((n, h, w), device=logits.device, dtype=torch.float32)
    _lib.check(_lib.load().rcu_aleatoric(_lib.ptr(logits), _lib.ptr(sigma_raw), n, h * w, c, int(is_log_sigma), None,
                                         None, _lib.ptr(pred), _lib.ptr(sp), _lib.current_stream()))
    return pred, sp


def channel_to_end(tensor):
    """NCHW -> NHWC view, as the test loop applies before ``.cpu().numpy()`` (torchhelper.py:10-23; loops.py:214-220)."""
    dims = tensor.dim()
    return tensor.permute(0, *range(2, dims), 1)
