"""The reference's test / evaluation command lines, running on librcu_hip.

Each function is the ``main`` of the same-named reference script (flags unchanged):
    bin-dl/brats_test_default.py   -config_file / -config_id      (bin-dl/brats_test_default.py:22-61, 113-117)
    bin-dl/brats_test_ensemble.py  -config_file                   (bin-dl/brats_test_ensemble.py:25-69)
    bin-dl/brats_test_aleatoric.py -config_file                   (bin-dl/brats_test_aleatoric.py:24-48)
    bin-dl/isic_test_default.py / isic_test_ensemble.py / isic_test_aleatoric.py
    bin-dl/{brats,isic}_test_auxiliary_feat.py -config_file       (bin-dl/brats_test_auxiliary_feat.py:23-61)
    bin-dl/{brats,isic}_test_auxiliary_segm.py -config_file       (bin-dl/brats_test_auxiliary_segm.py:23-47)
    bin-eval/eval_uncertainty.py   --ds --ids --act               (bin-eval/eval_uncertainty.py:13-50, 248-288)
Config ids map to the same YAML names under ``<project>/config``; paths inside the YAML stay relative to the
working directory, as in the reference.  Datasets: see rcu_amd.data (volume directories instead of pymia HDF5).
"""
import logging
import os

import numpy as np
import torch

from . import data as data_mod
from . import distributed as rdist
from . import evalrun
from . import evaluation as ev
from . import loops
from . import management as mgt
from . import nifti
from . import steps

PROJECT_DIR = os.environ.get('RCU_PROJECT_DIR', os.getcwd())
CONFIG_DIR = os.path.join(PROJECT_DIR, 'config')


def _config_path(dataset, config_id, default='baseline'):
    if config_id in ('baseline', 'baseline_mc', 'center', 'center_mc'):
        return os.path.join(CONFIG_DIR, 'test_{}_{}.yaml'.format(dataset, config_id))
    if config_id in ('cv0', 'cv1', 'cv2', 'cv3', 'cv4'):
        return os.path.join(CONFIG_DIR, 'baseline_cv', 'test_{}_baseline_cv{}.yaml'.format(dataset, config_id[-1]))
    return os.path.join(CONFIG_DIR, 'test_{}_{}.yaml'.format(dataset, default))


class EvalSubjectStep(loops.SubjectStep):
    """Dice of the arg-max prediction (brats_test_default.py:63-77; isic_test_default.py:70-86)."""

    def __init__(self, squeeze_labels=False, keep_prediction=False):
        self.evaluate = ev.ComposeEvaluation([ev.DiceNumpy()])
        self.squeeze_labels, self.keep_prediction = squeeze_labels, keep_prediction

    def __call__(self, subject_context, task_context, context) -> None:
        probabilities = subject_context.subject_data['probabilities']
        per_slice = subject_context.subject_data.pop('confusion', None)
        if per_slice is not None:
            # the counts were taken slice by slice on the GPU, behind the predict step (ConfusionOnDeviceStep), and came to the host with the
            # batches: integers, so their sum IS the confusion matrix of the assembled subject.  Nothing here touches the GPU -- and a
            # volume's arg-max for the *_prediction file stays with the writer thread
            tp, _, fp, fn = (int(v) for v in np.asarray(per_slice).reshape(-1, 4).sum(axis=0))
            subject_context.metrics.update({'dice': ev.dice_from_counts(tp, fp, fn)})
            made = subject_context.subject_data.get('prediction')       # 2-D subjects: the arg-max map came with the batch too
            if made is not None:
                made = np.asarray(made, dtype=np.uint8)
                made = made[..., 0] if (made.ndim == np.ndim(probabilities) and made.shape[-1] == 1) else made
                subject_context.more['prediction'] = (made, probabilities)
            if self.keep_prediction:
                if made is None:
                    made = nifti.argmax_last(probabilities, np.uint8)
                    subject_context.more['prediction'] = (made, probabilities)
                subject_context.subject_data['prediction'] = made.astype(np.int64)
            elif made is not None:
                del subject_context.subject_data['prediction']
            return
        prediction = nifti.argmax_last(probabilities, np.uint8)      # (class indices: uint8 holds them; np.argmax's int64 where it is kept)
        if self.keep_prediction:
            subject_context.subject_data['prediction'] = prediction.astype(np.int64)
        # the writer hook takes it from here instead of a second arg-max -- as long as ``probabilities`` still is the array it was taken from
        # (a later subject step may replace it: the reference's writer arg-maxes at write time, brats_test_default.py:96-99)
        subject_context.more['prediction'] = (prediction, probabilities)
        target = subject_context.subject_data['labels']
        if self.squeeze_labels:
            target = target.squeeze(-1)
        results = {}
        self.evaluate({'prediction': prediction, 'probabilities': probabilities, 'target': target}, results)
        subject_context.metrics.update(results)


class WriteHook(loops.TestLoopHook):
    """``{subject}_probabilities|_prediction[|_sigma].nii.gz`` from a background thread
    (brats_test_default.py:80-108; brats_test_aleatoric.py:76-110; isic_test_default.py:89-124)."""

    def __init__(self, with_sigma=False, link_inputs=False, in_background=True):
        self.with_sigma, self.link_inputs, self.in_background = with_sigma, link_inputs, in_background

    def on_test_subject_end(self, subject_context, task_context, context):
        if not isinstance(context, loops.TorchTestContext):
            raise ValueError('expected type is "TorchTestContext" but object is of type "{}"'.format(type(context).__name__))
        data = subject_context.subject_data
        subject = data.get('subject', subject_context.subject_index)
        cached = subject_context.more.get('prediction')
        prediction = cached[0] if (isinstance(cached, tuple) and cached[1] is data['probabilities']) else None
        nifti.write_subject(context.test_dir, subject, data['probabilities'], data.get('properties'),
                            data['sigma'] if self.with_sigma else None, in_background=self.in_background,
                            prediction=prediction)
        if self.link_inputs:   # ISIC: symlink image and label next to the outputs
            files = context.test_data.dataset.get_files_by_id(subject_context.subject_index)
            for key in ('label_paths', 'image_paths'):
                src = os.path.abspath(files[key])
                dst = os.path.join(context.test_dir, os.path.basename(src))
                if not os.path.lexists(dst):
                    os.symlink(src, dst)

    def on_termination(self, context):
        nifti.join_all()


class ConfusionOnDeviceStep(steps.BatchStep):
    """The Dice counts of the subjects' evaluation (EvalSubjectStep: brats_test_default.py:63-77, isic_test_default.py:70-86) taken at BATCH
    time, on the compute stream, right behind the predict steps: tp / tn / fp / fn of every slice (image) of the batch against its labels ->
    ``output['confusion']`` ``[N, 4]`` int64, which travels to the host with the batch's other kept entries and is assembled like them
    (``[D, 4]`` per volume, ``[4]`` per 2-D subject).  2-D subjects also get ``output['prediction']`` -- the arg-max map (uint8) their
    evaluation and their writer want.
    Why: at subject level the same counts cost the test loop's MAIN thread three small synchronous GPU operations, and next to a loop that
    has run ahead -- every CU held by a persistent conv kernel -- each of them waits its turn: 72-94 ms per BraTS subject, 6.8 ms per ISIC
    image (218 ms per batch of 32, whose GPU work takes 61: tools/host_costs_probe.py, tools/isic_script_throughput.py).  Here they are
    stream-ordered work like any kernel of the batch and nobody waits for them.
    Labels reach the device through a ring of pinned buffers (a copy may still be queued behind the batches the loop has run ahead with):
    a volume's labels once per subject -- from the dataset, sliced on the device --, a 2-D batch's own ``labels`` entry once per batch."""

    RING = 16          # > loops.Test.MAX_INFLIGHT: the copy that used a buffer last has long run when its turn comes again

    def __init__(self):
        self._pinned = [None] * self.RING      # (pinned uint8 tensor, event of the copy that read it last)
        self._next = 0
        self._volumes = {}         # subject index -> device uint8 [D, H, W]
        self._side = None          # the root of a sharded run: the stream the counts are taken on (behind the batch's collective)

    def _upload(self, labels, device):
        """host uint8 array -> device tensor of its shape, asynchronously on the current stream."""
        slot = self._next % self.RING
        self._next += 1
        entry = self._pinned[slot]
        if entry is None or entry[0].numel() < labels.size:
            entry = (torch.empty(labels.size, dtype=torch.uint8, pin_memory=True), torch.cuda.Event())
        else:
            entry[1].synchronize()       # the copy that used this buffer RING uploads ago has run
        self._pinned[slot] = entry
        host = entry[0][:labels.size]
        # (numpy's copy, not torch's: a torch CPU copy of this size starts an OpenMP team on the calling thread -- beside the loader thread's
        # own team the 4 MB took 35 ms, tools/host_costs_probe.py)
        np.copyto(host.numpy(), labels.reshape(-1))
        dev = torch.empty(labels.size, dtype=torch.uint8, device=device)
        dev.copy_(host, non_blocking=True)
        entry[1].record()
        return dev.view(labels.shape)

    def _labels_of(self, subject, dataset, device):
        vol = self._volumes.get(subject)
        if vol is None:
            labels = dataset.direct_extract(subject, ('labels',)).get('labels')
            if labels is None:           # a dataset without labels: nothing to count (EvalSubjectStep will say so, as in the reference)
                return None
            labels = np.ascontiguousarray(labels, dtype=np.uint8)
            if labels.ndim == 4 and labels.shape[-1] == 1:
                labels = labels[..., 0]
            vol = self._volumes[subject] = self._upload(labels, device)
            for other in [k for k in self._volumes if k != subject][:-1]:      # this subject and the one before it (a batch may span two)
                del self._volumes[other]
        return vol

    def __call__(self, batch_context, task_context, context) -> None:
        probabilities = batch_context.output.get('probabilities')
        if probabilities is None or not probabilities.is_cuda:
            return            # (a rank other than the root of a sharded run has nothing to evaluate)
        batch = batch_context.input
        dataset = getattr(getattr(task_context, 'data', None), 'dataset', None)
        volumes = 'slice_index' in batch and hasattr(dataset, 'direct_extract')
        labels = None if volumes else batch.get('labels')
        if not volumes and not (torch.is_tensor(labels) and labels.numel() == probabilities.shape[0] * probabilities.shape[2] * probabilities.shape[3]):
            return            # (no labels of the batch's shape at hand: EvalSubjectStep evaluates the assembled subject itself)
        # The root of a sharded run: the summary's outputs come from a side stream BEHIND the batch's collective.  Counting on the compute stream
        # would put that stream -- and with it the next batch's passes -- behind every batch's reduce and finalize, on the one rank every reduce
        # lands on; the counts are taken on a stream of their own that waits for the outputs, and the download waits for the counts instead.
        ready = batch_context.more.get('outputs_ready')
        if ready is None:
            self._count(batch_context, task_context, probabilities, batch, dataset, volumes, labels)
            return
        if self._side is None or self._side.device != probabilities.device:
            self._side = torch.cuda.Stream(device=probabilities.device)
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            if self._count(batch_context, task_context, probabilities, batch, dataset, volumes, labels):
                probabilities.record_stream(self._side)
                done = torch.cuda.Event()
                done.record(self._side)
                batch_context.more['outputs_ready'] = done      # (behind the old event by construction)

    def _count(self, batch_context, task_context, probabilities, batch, dataset, volumes, labels):
        """The arg-max map and the slice-wise confusion counts on the CURRENT stream -> whether they were taken."""
        prediction, _ = steps.prediction_and_foreground(probabilities)
        if volumes:
            subjects = [int(v) for v in batch['subject_index']]
            slices = [int(v) for v in batch['slice_index']]
            pieces, b, n = [], 0, len(subjects)
            while b < n:                                                        # runs of consecutive slices of one subject
                e = b + 1
                while e < n and subjects[e] == subjects[b] and slices[e] == slices[e - 1] + 1:
                    e += 1
                volume = self._labels_of(subjects[b], dataset, prediction.device)
                if volume is None:
                    return False
                pieces.append(volume[slices[b]:slices[b] + (e - b)])
                b = e
            target = pieces[0] if len(pieces) == 1 else torch.cat(pieces)
        else:
            # the cast evaluation._to_dev makes of the assembled subject's labels (the shipped ISIC configs rescale them to float 0 / 1)
            host = labels.detach().cpu().numpy()
            host = host if host.dtype == np.uint8 else host.astype(np.uint8)
            target = self._upload(np.ascontiguousarray(host).reshape(prediction.shape), prediction.device)
            batch_context.output['prediction'] = prediction.unsqueeze(1)      # (outputs carry the channel dim at 1: the loop moves it to the end)
        batch_context.output['confusion'] = ev.confusion_counts_on_device(prediction, target)
        return True


class CollectOnDeviceStep(steps.BatchStep):
    """Part of ``others.device_metrics`` (DeviceMetricsHook): right behind the predict steps, the batch's foreground probability and
    arg-max prediction are copied -- device to device, on the compute stream -- into per-subject volumes that stay in HBM until the subject
    has been evaluated there."""

    def __init__(self, store):
        self.store = store          # subject key -> {'p': [D, H, W] float32, 'prediction': [D, H, W] uint8} (2D subjects: [H, W])

    def __call__(self, batch_context, task_context, context) -> None:
        probabilities = batch_context.output.get('probabilities')
        if probabilities is None or not probabilities.is_cuda:
            return            # (a rank other than the root of a sharded run has nothing to evaluate)
        steps.wait_for_outputs(batch_context)      # (the root of a sharded run: the summary's outputs come from a side stream)
        pred, fg = steps.prediction_and_foreground(probabilities)            # what the writer derives on the host (brats_test_default.py:96-99)
        batch = batch_context.input
        if 'slice_index' in batch:                                          # slices of volumes
            subjects = [int(v) for v in batch['subject_index']]
            slices = [int(v) for v in batch['slice_index']]
            b, n = 0, len(subjects)
            while b < n:                                                    # runs of consecutive slices of one subject: one copy each
                e = b + 1
                while e < n and subjects[e] == subjects[b] and slices[e] == slices[e - 1] + 1:
                    e += 1
                vol = self.store.get(subjects[b])
                if vol is None:
                    depth = int(batch['shape'][b][0])
                    vol = self.store[subjects[b]] = {'p': fg.new_empty((depth,) + tuple(fg.shape[1:])),
                                                     'prediction': pred.new_empty((depth,) + tuple(pred.shape[1:]))}
                vol['p'][slices[b]:slices[b] + (e - b)] = fg[b:e]
                vol['prediction'][slices[b]:slices[b] + (e - b)] = pred[b:e]
                b = e
        else:                                                               # every sample is a subject (ISIC)
            for b, id_ in enumerate(batch['ids']):
                self.store[id_] = {'p': fg[b].clone(), 'prediction': pred[b].clone()}


class DeviceMetricsHook(loops.TestLoopHook):
    """OPT-IN (YAML ``others.device_metrics``, an rcu_amd extension): the calibration / uncertainty-error metrics of
    bin-eval/eval_uncertainty.py computed on every subject while its maps are still in HBM -- no .nii.gz round trip
    (bin-dl/brats_test_default.py:96-108 -> rechun/eval/analysis.py:75-107) -- and written as the CSV files the evaluation script would
    write from the files of this run (same names, columns, rows: the float32 round trip through NIfTI is lossless).

        others:
          device_metrics:
            actions: [minmax, ece_dice, calib, bnf_ue]     # default: all four
            run_id: baseline_mc                            # the test_id column and the file names (default: the config's test_name)
            gt_dir: /data/Brats18/Training                 # ground-truth tree (ISIC: the dataset prefix) the evaluation reads target and T2
                                                           # mask from; without it: the dataset's labels, no mask
            out_dir: null                                  # default: <test_dir>/eval

    The reference's contract stays the file round trip; this hook is the fused path SURVEY.md 8(f)1 leaves optional."""

    def __init__(self, dataset, spec, store):
        spec = dict(spec or {}) if not isinstance(spec, (list, tuple)) else {'actions': list(spec)}
        self.dataset = dataset
        self.action_names = list(spec.get('actions') or ['minmax', 'ece_dice', 'calib', 'bnf_ue'])
        self.run_id, self.gt_dir, self.out_dir = spec.get('run_id'), spec.get('gt_dir'), spec.get('out_dir')
        self.store = store
        self.actions, self.rows, self.gts = [], {}, None
        self.batch = None

    def on_test_start(self, task_context, context):
        base = self.out_dir or os.path.join(context.test_dir, 'eval')
        details = 'foreground' if (self.dataset == 'brats' and self.gt_dir) else ''      # eval_uncertainty.py:19-26
        self.actions = evalrun.get_actions(self.action_names, os.path.join(base, evalrun.MINMAX_NAME), base, details)
        entry = evalrun.EvalData(self.run_id or context.config.test_name, context.test_dir, 'probabilities')
        for action in self.actions:
            action.setup_eval(entry)
        if not evalrun._fusable(entry, self.actions):
            raise ValueError('others.device_metrics: unsupported actions {}'.format(self.action_names))
        for action in self.actions:
            action.start_eval()
        self.want, self.thresholds, self.want_mask = evalrun.metrics_wanted(self.actions)
        if self.gt_dir:
            gts = (evalrun.collect_brats_ground_truth if self.dataset == 'brats' else evalrun.collect_isic_ground_truth)(self.gt_dir)
            self.gts = {sf.subject: sf for sf in gts}

    def on_test_subject_end(self, subject_context, task_context, context):
        vol = self.store.pop(subject_context.subject_index, None)
        if vol is None:
            raise ValueError('others.device_metrics: subject {} was not collected on the device'.format(subject_context.subject_index))
        name = str(loops._subject_name(subject_context))
        mask = None
        if self.gts is not None:             # the evaluation's own inputs (analysis.py:88-89, 118-125)
            sf = self.gts[name]
            target = (evalrun.read_label_image(sf.categories['labels']['gt']) > 0).astype(np.uint8)
            if self.want_mask:
                mask = nifti.read(sf.categories['images']['t2'])[0] > 0
        else:
            target = (np.asarray(subject_context.subject_data['labels']) > 0).astype(np.uint8)
        n = vol['p'].numel()
        target = np.squeeze(target) if target.size == n and target.shape != tuple(vol['p'].shape) else target
        if self.batch is None or self.batch.n != n:
            self.batch = ev.SubjectBatch(1, n, device=vol['p'].device, with_mask=self.want_mask)
        self.batch.used = 0
        self.batch.put(0, vol['p'], vol['prediction'], target, mask)
        self.batch.upload()
        self.rows[name] = (self.batch.metrics(thresholds=self.thresholds, want=self.want), target.ndim)

    def on_test_end(self, task_context, context):
        for name in sorted(self.rows):       # the evaluation script's subject order (rechun/eval/evaldata.py: sorted by subject)
            res, n_dim = self.rows[name]
            evalrun.record_subject(self.actions, name, res, 0, n_dim)
        for action in self.actions:
            action.finish_eval()
        self.rows = {}


class PrepareSubjectStep(steps.BatchStep):
    def __call__(self, batch_context, task_context, context) -> None:
        batch_context.output['labels'] = batch_context.input['labels'].unsqueeze(1)   # isic_test_default.py:62-66


def _other(context, key, default=None):
    """A key of the YAML file's free-form ``others`` (rcu_amd extensions, absent from the reference's configs: coalesce_pixels, pipelined,
    max_inflight, stream_lanes, device_metrics)."""
    return getattr(context.config.others, key, default)


def _mask_seed(context, world):
    """Seed of the per-(pass, slice) Dropout2d masks: the YAML file's ``seed`` (the reference seeds torch with it,
    common/trainloop/loops.py:183-185).  Without one a one-process run draws from the device generator; a sharded run agrees on a
    random seed (the masks of a pass must not depend on the rank that runs it)."""
    seed = context.config.seed
    if seed is None and world.world > 1:
        import torch
        import torch.distributed as dist
        t = torch.randint(2 ** 31 - 1, (1,), dtype=torch.int64)
        if world.backend == 'nccl':
            t = t.to(world.device)
        dist.broadcast(t, src=0)
        seed = int(t.item())
    return seed


def _default_steps(context, world):
    if hasattr(context.config.others, 'mc'):
        lanes = _other(context, 'stream_lanes')
        # ``others.group_pixels`` / ``others.exact`` (rcu_amd extensions): the memory / reproducibility trade of the MC step.  Every stream lane of every
        # rank sizes its plan for n x min(group_pixels // (n H W), T) samples (canonical plans: 24 GB per lane for four passes of a 160-slice BraTS
        # batch) and ``exact`` keeps the statistics as exact float64 sums; a smaller ``group_pixels`` (or ``stream_lanes: 1``) shrinks the workspace,
        # ``exact: false`` halves the statistics and the reduce at the price of byte-identity across lanes, groups and world sizes.
        group_pixels, exact = _other(context, 'group_pixels'), bool(_other(context, 'exact', True))
        if world.world > 1:     # the T + 1 passes of every batch sharded over the ranks, one sum-reduce per batch (rcu_amd.distributed)
            return [rdist.ShardedMcPredictStep(context.config.others.mc, world, seed=_mask_seed(context, world), lanes=lanes,
                                               group_pixels=group_pixels, exact=exact),
                    steps.MultiPredictionSummary()]
        return [steps.McPredictStep(context.config.others.mc, seed=_mask_seed(context, world), lanes=lanes, group_pixels=group_pixels, exact=exact),
                steps.MultiPredictionSummary()]
    return [steps.SegmentationPredictStep(do_probs=True)]


def _hooks(write_hook, extra=()):
    hooks = [loops.ConsoleTestLogHook(), loops.WriteTestMetricsCsvHook('metrics.csv'), write_hook] + list(extra)
    return loops.ReducedComposeTestLoopHook(hooks)


def _load_additional_models(context):
    """others.model_dir (list) + others.test_at -> extra ensemble members (brats_test_ensemble.py:37-57)."""
    others = context.config.others
    if not hasattr(others, 'model_dir') or not hasattr(others, 'test_at'):
        raise ValueError('missing "model_dir" or "test_at" entry in the configuration (others)')
    model_dirs = [others.model_dir] if isinstance(others.model_dir, str) else list(others.model_dir)
    models = []
    for i, model_dir in enumerate(model_dirs):
        logging.info('load additional model [{}/{}] {}'.format(i + 1, len(model_dirs), os.path.basename(model_dir)))
        mf = mgt.ModelFiles.from_model_dir(model_dir)
        model = mgt.load_model_from_parameters(mf.model_path())
        mgt.load_checkpoint(mgt.find_checkpoint_file(mf.weight_checkpoint_dir, others.test_at), model)
        models.append(model.to(context.device).eval())
    return models


def _loop_options(context):
    """Test-loop options from the YAML file's ``others`` (rcu_amd extensions): ``coalesce_pixels`` -- merge loader batches up to that many
    pixels per step (loops.Test; DEFAULT since round 6: ``loops.Test.COALESCE_PIXELS``, one benchmark-sized BraTS volume -- the shipped
    ``batch_size: 32`` then runs as launches that fill the GPU, and, the seeded Dropout2d masks being keyed by the slice and not by the batch,
    the same YAML file writes the same files for any ``batch_size``; ``0`` switches it off: steps and hooks then see the loader's batches, as in
    the reference; two BraTS volumes, 7864320, keep the launches of an 8-GPU run at 640 samples) --, ``pipelined: false`` (the reference's
    callback order), ``max_inflight``, ``loader_timing: true`` (the loader thread logs where its time went)."""
    value = _other(context, 'coalesce_pixels')
    return dict(coalesce=loops.Test.COALESCE_PIXELS if value is None else int(value), pipelined=_other(context, 'pipelined'),
                max_inflight=_other(context, 'max_inflight'), loader_timing=bool(_other(context, 'loader_timing', False)))


def _context(device, config_file):
    """-> (context, world): under ``python -m torch.distributed.run`` every rank gets its own GPU (rcu_amd.distributed.world_from_env) and
    only rank 0 creates the test directory, logs, assembles, evaluates and writes."""
    world = rdist.world_from_env(device)
    context = loops.TorchTestContext(world.device)
    context.writes_output = world.is_root
    context.load_from_config(config_file)
    return context, world


def _run(context, dataset, test_steps, write_hook, entries, world=None):
    world = world if world is not None else rdist.World()
    sharded = [s_ for s_ in test_steps if isinstance(s_, rdist._ShardedStepBase)]
    if world.world > 1 and not sharded:
        # one deterministic forward pass per batch: nothing to shard over samples -- the run is rank 0's, the other ranks leave at once
        if not world.is_root:
            logging.info('rank {}: this configuration has one forward pass per batch, nothing to shard; rank 0 runs it'.format(world.rank))
            return context
    build = data_mod.BuildData(build_dataset=data_mod.BuildVolumeDataset() if dataset == 'brats' else data_mod.BuildIsicDataset())
    options = _loop_options(context)
    extra_hooks = []
    spec = _other(context, 'device_metrics')
    if spec and world.is_root:      # opt-in: the evaluation's metrics on the maps while they are in HBM (DeviceMetricsHook)
        store = {}
        test_steps = test_steps + [CollectOnDeviceStep(store)]
        extra_hooks.append(DeviceMetricsHook(dataset, spec, store))
    if world.is_root and bool(_other(context, 'device_confusion', True)):
        # the subjects' Dice counts ride with the batches (ConfusionOnDeviceStep; ``others.device_confusion: false`` keeps them at subject level)
        test_steps = test_steps + [ConfusionOnDeviceStep()]
        entries = None if entries is None else tuple(entries) + ('confusion',)
    if dataset != 'brats':
        test_steps = test_steps + [PrepareSubjectStep()]
    if not world.is_root:
        # a rank other than the root of a sharded run: the same loader and the same batch steps, nothing assembled, evaluated or written
        # (the same coalescing: every rank must see the root's batches -- their shapes size the collective, their indices rotate the jobs)
        test = loops.Test(test_steps, [], None, entries=(), coalesce=options['coalesce'], pipelined=options['pipelined'],
                          max_inflight=options['max_inflight'])
        hook = loops.TestLoopHook()
    elif dataset == 'brats':
        test = loops.Test(test_steps, [loops.ExtractSubjectInfoStep(), EvalSubjectStep()], loops.SubjectAssembler(),
                          entries=entries, **options)
        hook = _hooks(write_hook, extra_hooks)
    else:
        test = loops.Test(test_steps, [EvalSubjectStep(squeeze_labels=True, keep_prediction=True)],
                          loops.Subject2dAssembler(), entries=entries, **options)
        hook = _hooks(write_hook, extra_hooks)
    try:
        test(context, build, hook=hook)
    except BaseException:
        if world.world > 1 and sharded:
            # A rank that failed must NOT issue the closing collectives: the other ranks are still in their per-batch reduces, a barrier from
            # here would pair with one of those (mismatched collectives: RCCL hangs until its watchdog fires) and the traceback below would
            # never be printed.  Log it, tear this rank's communicator down so that the peers' pending collectives fail instead of waiting, and
            # let the exception end the process: a non-zero exit makes the launcher (torch.distributed.run) end the other ranks.
            logging.exception('rank {} of {}: the test loop failed; leaving without the closing barrier'.format(world.rank, world.world))
            _abandon_process_group()
        raise
    else:
        if world.world > 1 and sharded:      # (a run that is rank 0's alone has nobody to meet: the other ranks have left)
            import torch
            import torch.distributed as dist
            for s_ in sharded:
                s_.finish()
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dist.barrier()           # no rank leaves (and frees what a collective still reads) before the root has the last batch
    return context


def _abandon_process_group(timeout_s=10.0):
    """Best effort, bounded in time: destroy this rank's process group from a helper thread (a communicator with collectives in flight may not
    come down at once; the exception that brought us here must not wait for it)."""
    import threading
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return

    def destroy():
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 - we are already failing; the original exception is the one to report
            pass

    t = threading.Thread(target=destroy, name='rcu-abandon-process-group', daemon=True)
    t.start()
    t.join(timeout_s)


def test_default(dataset, config_file=None, config_id=None, device='cuda'):
    context, world = _context(device, config_file or _config_path(dataset, config_id))
    entries = ('probabilities',) if dataset == 'brats' else None
    return _run(context, dataset, _default_steps(context, world), WriteHook(link_inputs=dataset == 'isic'), entries, world)


def test_ensemble(dataset, config_file=None, device='cuda'):
    context, world = _context(device, config_file or os.path.join(CONFIG_DIR, 'test_{}_ensemble.yaml'.format(dataset)))
    members = _load_additional_models(context)
    lanes = _other(context, 'stream_lanes')
    if world.world > 1:     # the K members of every batch sharded over the ranks (bin-dl/brats_test_ensemble.py:44-59 on N GPUs)
        test_steps = [rdist.ShardedEnsemblePredictionStep(members, world, lanes=lanes), steps.MultiPredictionSummary()]
    else:
        test_steps = [steps.EnsemblePredictionStep(members, lanes=lanes), steps.MultiPredictionSummary()]
    return _run(context, dataset, test_steps, WriteHook(link_inputs=dataset == 'isic'), None, world)


def test_aleatoric(dataset, config_file=None, device='cuda'):
    context, world = _context(device, config_file or os.path.join(CONFIG_DIR, 'test_{}_aleatoric.yaml'.format(dataset)))
    return _run(context, dataset, [steps.AleatoricPredictStep()], WriteHook(with_sigma=True, link_inputs=dataset == 'isic'),
                None, world)


# ------------------------------------------------------------------------------- auxiliary networks
class AuxiliaryFeatPredictStep(steps.BatchStep):
    """bin-dl/brats_test_auxiliary_feat.py:64-82: segmentation network with provide_features, then the auxiliary
    PostNet (``context.model``) on its features; the features stay in the U-Net's workspace (no copy)."""

    def __init__(self, test_model):
        self.test_model = test_model

    def __call__(self, batch_context, task_context, context) -> None:
        if not isinstance(context, steps.Context):
            raise ValueError('expected type is "TorchTestContext" but object is of type "{}"'.format(type(context).__name__))
        batch_context.input['images'] = batch_context.input['images'].float().to(context.device)
        segm_logits = self.test_model(batch_context.input['images'])
        batch_context.output['segm_probabilities'] = steps.softmax(segm_logits)
        logits = context.model(self.test_model.features)
        batch_context.output['probabilities'] = steps.softmax(logits)


class AuxiliarySegmPredictStep(steps.BatchStep):
    """bin-dl/brats_test_auxiliary_segm.py:50-71: the auxiliary U-Net sees the images plus the segmentation to be
    judged (labels channel 1) and predicts where that segmentation is wrong."""

    def __init__(self, keep_labels=False):
        self.keep_labels = keep_labels       # isic_test_auxiliary_segm.py:80-81

    def __call__(self, batch_context, task_context, context) -> None:
        if not isinstance(context, steps.Context):
            raise ValueError('expected type is "TorchTestContext" but object is of type "{}"'.format(type(context).__name__))
        import torch
        batch_context.input['images'] = batch_context.input['images'].float().to(context.device)
        batch_context.input['labels'] = batch_context.input['labels'].long().to(context.device)
        pred = batch_context.input['labels'][:, 1]
        inpt = torch.cat([batch_context.input['images'], pred.unsqueeze(1).float()], dim=1)
        logits = context.model(inpt)
        batch_context.output['logits'] = logits
        batch_context.output['probabilities'] = steps.softmax(logits)
        batch_context.output['orig_prediction'] = pred.unsqueeze(1)
        if self.keep_labels:
            batch_context.output['labels'] = batch_context.input['labels']


class EvalAuxiliaryFeatStep(loops.SubjectStep):
    """Dice of the SEGMENTATION network's arg-max (brats_test_auxiliary_feat.py:85-99)."""

    def __init__(self, squeeze_labels=False):
        self.evaluate = ev.ComposeEvaluation([ev.DiceNumpy()])
        self.squeeze_labels = squeeze_labels

    def __call__(self, subject_context, task_context, context) -> None:
        probabilities = subject_context.subject_data['segm_probabilities']
        target = subject_context.subject_data['labels']
        if self.squeeze_labels:
            target = target.squeeze(-1)
        results = {}
        self.evaluate({'prediction': np.argmax(probabilities, axis=-1), 'probabilities': probabilities, 'target': target},
                      results)
        subject_context.metrics.update(results)


class EvalAuxiliarySegmStep(loops.SubjectStep):
    """Dice of "predicted wrong" against "is wrong" (brats_test_auxiliary_segm.py:74-91)."""

    def __init__(self, set_score=False):
        self.evaluate = ev.ComposeEvaluation([ev.DiceNumpy()])
        self.set_score = set_score

    def __call__(self, subject_context, task_context, context) -> None:
        probabilities = subject_context.subject_data['probabilities']
        labels = subject_context.subject_data['labels']
        results = {}
        self.evaluate({'prediction': np.argmax(probabilities, axis=-1), 'probabilities': probabilities,
                       'target': labels[..., 1] != labels[..., 0]}, results)
        subject_context.metrics.update(results)
        if self.set_score:
            subject_context.score = results['dice']


class WriteConfidenceHook(loops.TestLoopHook):
    """``{subject}_confidence.nii.gz`` (foreground output of the auxiliary network) + ``{subject}_prediction.nii.gz``
    (brats_test_auxiliary_feat.py:102-135, brats_test_auxiliary_segm.py:94-124; ISIC variants symlink the inputs)."""

    def __init__(self, prediction_entry, link=(), in_background=True):
        self.prediction_entry, self.link, self.in_background = prediction_entry, tuple(link), in_background

    def on_test_subject_end(self, subject_context, task_context, context):
        if not isinstance(context, loops.TorchTestContext):
            raise ValueError('expected type is "TorchTestContext" but object is of type "{}"'.format(type(context).__name__))
        data = subject_context.subject_data
        subject = data.get('subject', subject_context.subject_index)
        confidence = np.ascontiguousarray(data['probabilities'][..., 1], dtype=np.float32)
        if self.prediction_entry == 'segm_probabilities':
            prediction = np.argmax(data['segm_probabilities'], axis=-1).astype(np.uint8)
        else:
            prediction = np.squeeze(data['orig_prediction']).astype(np.uint8)
        props, test_dir = data.get('properties'), context.test_dir

        def work():
            nifti.write(os.path.join(test_dir, '{}_confidence.nii.gz'.format(subject)), confidence, props)
            nifti.write(os.path.join(test_dir, '{}_prediction.nii.gz'.format(subject)), prediction, props)

        nifti.do_work(work, in_background=self.in_background)
        if self.link:
            files = context.test_data.dataset.get_files_by_id(subject_context.subject_index)
            for key in self.link:
                src = os.path.abspath(files[key])
                dst = os.path.join(context.test_dir, os.path.basename(src))
                if not os.path.lexists(dst):
                    os.symlink(src, dst)

    def on_termination(self, context):
        nifti.join_all()


def _load_segmentation_model(context):
    """others.model_dir + others.test_at -> the trained segmentation U-Net with provide_features set
    (brats_test_auxiliary_feat.py:35-46)."""
    others = context.config.others
    if not hasattr(others, 'model_dir') or not hasattr(others, 'test_at'):
        raise ValueError('missing "model_dir" or "test_at" entry in the configuration (others)')
    mf = mgt.ModelFiles.from_model_dir(others.model_dir)
    model = mgt.load_model_from_parameters(mf.model_path())
    model.provide_features = True
    mgt.load_checkpoint(mgt.find_checkpoint_file(mf.weight_checkpoint_dir, others.test_at), model)
    return model.to(context.device).eval()


def _single_rank_only(world):
    """The auxiliary-network runs are one deterministic forward pass per batch: under a multi-process launch they are rank 0's alone."""
    if world.world > 1 and not world.is_root:
        logging.info('rank {}: this configuration has one forward pass per batch, nothing to shard; rank 0 runs it'.format(world.rank))
        return False
    return True


def test_auxiliary_feat(dataset, config_file=None, device='cuda'):
    context, world = _context(device, config_file or os.path.join(CONFIG_DIR, 'test_{}_auxiliary_feat.yaml'.format(dataset)))
    if not _single_rank_only(world):
        return context
    test_model = _load_segmentation_model(context)
    if dataset == 'brats':
        build = data_mod.BuildData(build_dataset=data_mod.BuildVolumeDataset())
        test = loops.Test([AuxiliaryFeatPredictStep(test_model)], [loops.ExtractSubjectInfoStep(), EvalAuxiliaryFeatStep()],
                          loops.SubjectAssembler(), entries=('probabilities', 'segm_probabilities'))
        hook = WriteConfidenceHook('segm_probabilities')
    else:
        build = data_mod.BuildData(build_dataset=data_mod.BuildIsicDataset())
        test = loops.Test([AuxiliaryFeatPredictStep(test_model), PrepareSubjectStep()],
                          [EvalAuxiliaryFeatStep(squeeze_labels=True)], loops.Subject2dAssembler(),
                          entries=('probabilities', 'segm_probabilities', 'labels'))
        hook = WriteConfidenceHook('segm_probabilities', link=('label_paths',))
    test(context, build, hook=_hooks(hook))
    return context


def test_auxiliary_segm(dataset, config_file=None, device='cuda'):
    context, world = _context(device, config_file or os.path.join(CONFIG_DIR, 'test_{}_auxiliary_segm.yaml'.format(dataset)))
    if not _single_rank_only(world):
        return context
    if dataset == 'brats':
        build = data_mod.BuildData(build_dataset=data_mod.BuildVolumeDataset())
        test = loops.Test([AuxiliarySegmPredictStep()], [loops.ExtractSubjectInfoStep(), EvalAuxiliarySegmStep()],
                          loops.SubjectAssembler(), entries=('probabilities', 'orig_prediction'))
        hook = WriteConfidenceHook('orig_prediction')
    else:
        if not hasattr(context.config.others, 'prediction_dir'):
            raise ValueError('"others.prediction_dir" is required in the config')
        build = data_mod.BuildData(build_dataset=data_mod.BuildIsicDataset(),
                                   prediction_dir=context.config.others.prediction_dir)
        test = loops.Test([AuxiliarySegmPredictStep(keep_labels=True)], [EvalAuxiliarySegmStep(set_score=True)],
                          loops.Subject2dAssembler(), entries=('probabilities', 'labels', 'orig_prediction'))
        hook = WriteConfidenceHook('orig_prediction', link=('label_paths', 'image_paths'))
    test(context, build, hook=_hooks(hook))
    return context


for _fn in (test_default, test_ensemble, test_aleatoric, test_auxiliary_feat, test_auxiliary_segm):
    _fn.__test__ = False   # not pytest tests


def eval_uncertainty(dataset, run_dirs: dict, ground_truth_dir, base_dir, actions=('minmax', 'ece_dice', 'calib', 'bnf_ue'),
                     expected_subjects=None, fused=True, batch_subjects=8, timing=None):
    """``run_dirs``: run id (baseline, baseline_mc, ..., aleatoric) -> prediction directory.  BraTS evaluates
    inside the T2 brain mask (``ece_details='foreground'``), ISIC on all pixels (eval_uncertainty.py:19-26).
    ``ground_truth_dir``: the BraTS tree of ``<subject>/<subject>_{t2,seg,...}.nii.gz`` or, for ISIC, the dataset
    prefix (``.../ISIC-2017_Test_v2``) whose ``_Data`` / ``_Part1_GroundTruth`` folders hold the jpg / png files."""
    if dataset not in ('brats', 'isic'):
        raise ValueError('chose "brats" or "isic" as dataset')
    if dataset == 'brats':
        gts = evalrun.collect_brats_ground_truth(ground_truth_dir)
        details = 'foreground'
    else:
        gts = evalrun.collect_isic_ground_truth(ground_truth_dir)
        details = ''
    entries = [evalrun.get_eval_data(run_id, path, gts, expected_subjects) for run_id, path in run_dirs.items()]
    # (fused / batch_subjects / timing: rcu_amd.evalrun.evaluate_runs -- one upload per subject shared by all actions, subjects batched per launch)
    evalrun.evaluate_runs(entries, list(actions), base_dir, details, fused=fused, batch_subjects=batch_subjects, timing=timing)
    return entries
