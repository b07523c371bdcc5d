"""Test loop, contexts, hooks and subject assemblers of the test scripts.

Mirrors
  Test.__call__ / _test_batch            common/trainloop/loops.py:165-235
  TorchTestContext                       common/trainloop/context.py:256-331
  TestLoopHook, ReducedComposeTestLoopHook, ConsoleTestLogHook, WriteTestMetricsCsvHook
                                         common/trainloop/hooks.py:67-98, 116-151, 250-294, 369-400
  SubjectStep, ExtractSubjectInfoStep    common/trainloop/steps.py:92-114
  SubjectAssembler / Subject2dAssembler  pymia 0.2.1 (absent; contract as used at loops.py:222-227: parity unpinned)
The loop hands every kept output entry to the assembler channel-last as numpy (loops.py:214-220), exactly
like the reference, so writer hooks and subject steps see the same arrays.
"""
import collections
import contextlib
import csv
import logging
import os
import random
import shutil
import sys
import time

import numpy as np
import torch

from . import config as cfg
from . import management as mgt
from . import steps as steps_mod
from .steps import BatchContext, TaskContext   # noqa: F401  (re-exported)

logging.basicConfig(format='%(message)s', stream=sys.stdout, level=logging.INFO)


class History:
    """task_context.history (common/trainloop/context.py:366-400)."""

    def __init__(self):
        self.categories = {}

    def add(self, entries: dict, category: str):
        for k, v in entries.items():
            self.categories.setdefault(category, {}).setdefault(k, []).append(v)

    def get_entries_keys(self, category):
        return tuple(self.categories.get(category, {}).keys())

    def get_entries(self, key, category):
        return self.categories[category][key]


class SubjectContext:
    def __init__(self, subject_index, subject_data: dict):
        self.subject_index = subject_index
        self.subject_data = subject_data
        self.metrics = {}
        self.score = None
        self.more = {}


def get_unique_identifier():
    return time.strftime('%y%m%d-%H%M%S')   # common/utils/idhelper.py:4-5


class TorchTestContext(steps_mod.TorchTestContext):

    def __init__(self, device_str: str = 'cuda'):
        super().__init__(device_str)
        self.writes_output = True      # False on the ranks other than the root of a sharded run (rcu_amd.scripts._context): no test directory, no log file
        self.config = None
        self.test_id = self.test_dir = self.log_file = ''
        self.model_files = None
        self.config_file_path = None
        self.test_data = None

    def load_from_config(self, config_file: str):
        self.config_file_path = config_file
        self.config = cfg.load(config_file, cfg.TestConfiguration)
        test_dir = self.config.test_dir
        if not test_dir:   # default: <train dir>/tests
            test_dir = os.path.join(os.path.dirname(self.config.model_dir), 'tests')
        self.test_id = get_unique_identifier()
        self.test_dir = os.path.join(test_dir, '{}_{}'.format(self.test_id, self.config.test_name))
        self.log_file = os.path.join(self.test_dir, 'log.txt')
        self.model_files = mgt.ModelFiles.from_model_dir(self.config.model_dir)

    def setup_directory(self):
        if not self.writes_output:
            return
        os.makedirs(self.test_dir, exist_ok=True)
        cfg.save(os.path.join(self.test_dir, 'config' + os.path.splitext(self.config_file_path)[1]), self.config)
        if self.config.split:
            shutil.copy(self.config.split, os.path.join(self.test_dir, os.path.basename(self.config.split)))

    def setup_logging(self):
        if not self.writes_output:
            return
        handler = logging.FileHandler(self.log_file)
        handler.setFormatter(logging.Formatter('%(asctime)s - %(filename)s:%(funcName)s %(levelname)s: %(message)s'))
        logging.getLogger().addHandler(handler)
        logging.info('Set up logging. Log file: {}'.format(self.log_file))

    def get_seed(self):
        return self.config.seed

    def do_seed(self, seed: int):
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)

    def load_test_data(self, build_test):
        params = {}
        if self.config.split:
            from . import data as data_mod
            _, _, test_entries = data_mod.load_split(self.config.split, getattr(self.config.others, 'split_k', None))
            params['entries'] = test_entries
        self.test_data = build_test(self.config.test_data, **params)

    def get_test_at(self):
        return self.config.test_at

    def get_task_context(self):
        tc = TaskContext(0, self.test_data, self.config.test_data)
        tc.history = History()
        return tc

    def load_from_checkpoint(self, epoch):
        path = mgt.find_checkpoint_file(self.model_files.weight_checkpoint_dir, epoch)
        model = mgt.load_model_from_parameters(self.model_files.model_path())
        mgt.load_checkpoint(path, model)
        self.model = model.to(self.device)
        self.model.eval()
        torch.set_grad_enabled(False)


# --------------------------------------------------------------------------------------- hooks
class TestLoopHook:
    __test__ = False

    def on_startup(self):
        pass

    def end_startup(self, context):
        pass

    def on_termination(self, context):
        pass

    def on_test_start(self, task_context, context):
        pass

    def on_test_end(self, task_context, context):
        pass

    def on_test_batch_start(self, batch_context, task_context, context):
        pass

    def on_test_batch_end(self, batch_context, task_context, context):
        pass

    def on_test_subject_start(self, subject_context, task_context, context):
        pass

    def on_test_subject_end(self, subject_context, task_context, context):
        pass


_HOOK_METHODS = [m for m in vars(TestLoopHook) if m.startswith(('on_', 'end_'))]


class ReducedComposeTestLoopHook(TestLoopHook):
    """Chains only the methods a member overrides (hooks.py:116-133, 148-151)."""

    def __init__(self, hooks: list):
        for name in _HOOK_METHODS:
            fns = [getattr(h, name) for h in hooks if getattr(type(h), name) is not getattr(TestLoopHook, name)]
            setattr(self, name, (lambda fs: (lambda *a, **k: [f(*a, **k) for f in fs] and None))(fns))


def _subject_name(subject_context):
    return subject_context.subject_data.get('subject', subject_context.subject_index)


class ConsoleTestLogHook(TestLoopHook):
    def __init__(self):
        self.t_start = self.t_subject = self.t_eval = None

    def on_startup(self):
        logging.info('startup')
        self.t_start = time.time()

    def end_startup(self, context):
        logging.info('model: \n{}'.format(str(context.model)))
        logging.info('startup finished')

    def on_termination(self, context):
        logging.info('\ntesting completed [{:.3}s]'.format(time.time() - self.t_start))

    def on_test_start(self, task_context, context):
        logging.info('testing')
        self.t_subject = time.time()

    def on_test_subject_start(self, subject_context, task_context, context):
        self.t_eval = time.time()

    def on_test_subject_end(self, subject_context, task_context, context):
        now = time.time()
        metrics = ' | '.join('{}: {:.5f}'.format(k, v) for k, v in subject_context.metrics.items())
        logging.info('[{} {:.3}s ({:.3})] {}'.format(_subject_name(subject_context), now - self.t_subject,
                                                     now - self.t_eval, metrics))
        self.t_subject = now


class WriteTestMetricsCsvHook(TestLoopHook):
    """``subject, <sorted metric names>`` rows into <test_dir>/<file_name> (hooks.py:369-400)."""

    def __init__(self, file_name):
        self.file_name = file_name
        self.subject_names = []

    def on_test_start(self, task_context, context):
        self.subject_names.clear()

    def on_test_subject_end(self, subject_context, task_context, context):
        self.subject_names.append(_subject_name(subject_context))

    def on_test_end(self, task_context, context):
        keys = sorted(task_context.history.get_entries_keys('subject_metrics'))
        with open(os.path.join(context.test_dir, self.file_name), 'w') as f:
            writer = csv.writer(f)
            writer.writerow(['subject'] + keys)
            for i, name in enumerate(self.subject_names):
                writer.writerow([name] + [task_context.history.get_entries(k, 'subject_metrics')[i] for k in keys])


# ---------------------------------------------------------------------------------- assemblers
class SubjectAssembler:
    """Collects slice-wise batch outputs (channel-last numpy ``[B, H, W, C]``) into per-subject volumes
    ``[D, H, W, C]``.  A subject is ready once the batches have moved on to the next subject (or the last
    batch has been added).  Needs ``subject_index``, ``slice_index`` and ``shape`` in the batch."""

    def __init__(self):
        self.volumes = {}            # subject index -> {key: array}
        self.subjects_ready = set()

    def add_batch(self, to_assemble: dict, batch: dict, last_batch=False):
        subjects = [int(v) for v in batch['subject_index']]
        slices = [int(v) for v in batch['slice_index']]
        b, n = 0, len(subjects)
        while b < n:                 # runs of consecutive slices of one subject are copied as blocks
            e = b + 1
            while e < n and subjects[e] == subjects[b] and slices[e] == slices[e - 1] + 1:
                e += 1
            si, k = subjects[b], slices[b]
            store = self.volumes.setdefault(si, {})
            for key, value in to_assemble.items():
                if key not in store:
                    depth = int(batch['shape'][b][0])
                    store[key] = np.zeros((depth,) + value.shape[1:], dtype=value.dtype)
                store[key][k:k + (e - b)] = value[b:e]
            b = e
        current = int(batch['subject_index'][-1])
        for si in self.volumes:
            if last_batch or si != current:
                self.subjects_ready.add(si)

    def get_assembled_subject(self, subject_index):
        self.subjects_ready.discard(subject_index)
        return self.volumes.pop(subject_index)


class Subject2dAssembler:
    """Every sample is a subject (ISIC); the subject key is the sample's id."""

    def __init__(self, id_entry='ids'):
        self.id_entry = id_entry
        self.store = {}
        self.subjects_ready = set()

    def add_batch(self, to_assemble: dict, batch: dict, last_batch=False):
        for b, id_ in enumerate(batch[self.id_entry]):
            self.store[id_] = {key: np.array(value[b]) for key, value in to_assemble.items()}   # a copy: the batch arrays may be reused buffers
            self.subjects_ready.add(id_)

    def get_assembled_subject(self, subject_index):
        self.subjects_ready.discard(subject_index)
        return self.store.pop(subject_index)


# -------------------------------------------------------------------------------- subject steps
class SubjectStep:
    def __call__(self, subject_context, task_context, context) -> None:
        pass


class ExtractSubjectInfoStep(SubjectStep):
    """labels / properties / subject name of the assembled subject from the dataset (steps.py:98-114)."""

    def __call__(self, subject_context, task_context, context) -> None:
        info = task_context.data.dataset.direct_extract(subject_context.subject_index)
        subject_context.subject_data.update(info)


# -------------------------------------------------------------------------------------- the loop
def tensor_to_numpy(tensor):
    return tensor.cpu().numpy()


def merge_batches(batches):
    """Concatenate collated dict batches (rcu_amd.data.CollateDict: stacked tensors + per-sample lists) along the sample axis."""
    if len(batches) == 1:
        return batches[0]
    out = {}
    for key, first in batches[0].items():
        if torch.is_tensor(first):
            out[key] = torch.cat([b[key] for b in batches])
        elif isinstance(first, (list, tuple)):
            out[key] = [v for b in batches for v in b[key]]
        else:
            out[key] = first
    return out


class BatchGroup(list):
    """Consecutive loader batches to be merged into one (see ``coalesced``)."""


def _batch_pixels(batch, entry='images'):
    v = batch.get(entry) if isinstance(batch, dict) else None
    if not torch.is_tensor(v) or v.dim() < 3:
        return None
    return int(v.shape[0]) * int(v.shape[-1]) * int(v.shape[-2])


def _batch_samples(batch, entry='images'):
    """Samples (slices / images) of a loader batch: the length of its image entry (channels-last or channels-first, tensor or array alike)."""
    v = batch.get(entry) if isinstance(batch, dict) else None
    try:
        return int(v.shape[0])
    except (AttributeError, IndexError, TypeError):
        return 0


def coalesced(iterable, max_pixels, entry='images', lazy=False):
    """Merge consecutive loader batches while the merged batch stays within ``max_pixels`` (samples x height x width of ``entry``)
    and the per-sample shapes agree.  The reference's ``batch_size`` is a loader setting: a forward pass is independent per sample
    (tests: batch-split invariance, bit for bit), so the outputs do not depend on how the slices are batched -- but the GPU fills
    only from about 160 BraTS slices on, and the shipped YAML files feed 32.
    ``lazy``: groups of two or more batches are handed on as a ``BatchGroup`` (a list) for the consumer to merge -- ``prefetch`` copies
    the pieces of a pinned entry straight into its staging buffer instead of concatenating first."""
    merge_batches = BatchGroup if lazy else globals()['merge_batches']
    pending, pixels = [], 0
    for batch in iterable:
        px = _batch_pixels(batch, entry)
        if px is None:                              # not a dict batch with an image tensor: passed through
            if pending:
                yield merge_batches(pending)
                pending, pixels = [], 0
            yield batch
            continue
        if pending:
            same = pending[0][entry].shape[1:] == batch[entry].shape[1:] and pending[0].keys() == batch.keys()
            if not same or pixels + px > max_pixels:
                yield merge_batches(pending)
                pending, pixels = [], 0
        pending.append(batch)
        pixels += px
        if pixels >= max_pixels:
            yield merge_batches(pending)
            pending, pixels = [], 0
    if pending:
        yield merge_batches(pending)


class _Staged:
    """One pinned host buffer of the loader's staging ring.  The worker fills it and hands it to the test loop inside a batch; the
    loop gives it back (``release``) with a CUDA event recorded behind the step's host-to-device copy, and the worker waits for
    that event before it writes the buffer again."""

    def __init__(self, shape, dtype, channels_last=False):
        self.tensor = torch.empty(shape, dtype=dtype, pin_memory=True,
                                  memory_format=torch.channels_last if channels_last else torch.contiguous_format)
        self.event = None
        self.busy = False
        self.released = 0        # order of the releases (the worker waits for the oldest one when none has completed)


def prefetch(iterable, depth=2, pin=False, pin_entries=('images',), timing=False):
    """Iterate ``iterable`` from a background thread, ``depth`` items ahead: the loader's work for the next batches (file reads,
    decompression -- zlib and numpy release the GIL -- transforms, collation) overlaps the GPU work of the current one.  ``pin``:
    the ``pin_entries`` of a dict batch (the tensors a step copies to the device) are staged in pinned host memory there too, so
    that the step's host-to-device copy is asynchronous.  Yields ``(item, release)``: call ``release()`` once the copies of the
    item's pinned entries have been ENQUEUED (it records a CUDA event; the buffer is not written again before that event has
    completed).  Entries the steps keep on the host (labels) are never staged: they are the loader's own tensors.
    ``timing``: the worker logs where its time went when the iterable is exhausted (tools/loop_timeline.py).
    Closing the generator (or an exception in the consumer) stops the worker."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    done = object()
    stop = threading.Event()
    cond = threading.Condition()
    ring = {}          # (entry, shape, dtype) -> [_Staged]

    def staged(key, v):
        pieces = v if isinstance(v, list) else [v]          # a list: the pieces of a batch group, merged by the copy itself
        shape = (sum(int(t.shape[0]) for t in pieces),) + tuple(pieces[0].shape[1:])
        # a batch in the file's channel-last order (data.VolumeDataset): staged as it is -- memcpy -- and re-ordered on the GPU
        channels_last = all(t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last) for t in pieces)
        ring_key = (key, shape, pieces[0].dtype, channels_last)
        if ring_key not in ring and len(ring) >= 4:        # batches of many shapes: drop the oldest shape's buffers (those still
            ring.pop(next(iter(ring)))                     # in flight stay alive through the batches that hold them)
        bufs = ring.setdefault(ring_key, [])
        t_a = time.perf_counter()
        with cond:
            while True:
                free = [b for b in bufs if not b.busy]
                # a released buffer may still be read by its host-to-device copy, which is queued behind the kernels of the batches
                # before it (up to Test.MAX_INFLIGHT of them): take one whose copy has completed; failing that grow the ring; failing
                # that wait for the buffer that was released first
                ready = [b for b in free if b.event is None or b.event.query()]
                if ready or len(bufs) < depth + 3:     # depth queued + one being filled + one with the consumer + one in flight
                    free = ready
                    break
                if free:
                    free = [min(free, key=lambda b: b.released)]
                    break
                cond.wait(0.05)
                if stop.is_set():
                    return None
            if free:
                buf = free[0]
            else:
                buf = _Staged(shape, pieces[0].dtype, channels_last)
                bufs.append(buf)
            buf.busy = True
        t_b = time.perf_counter()
        if buf.event is not None:
            buf.event.synchronize()                    # the copy that read this buffer last has finished
            buf.event = None
        t_c = time.perf_counter()
        at = 0
        for t in pieces:
            if channels_last:      # (numpy's memcpy on the underlying [n, H, W, C] arrays: a torch CPU copy starts an OpenMP team on this thread)
                np.copyto(buf.tensor[at:at + t.shape[0]].permute(0, 2, 3, 1).numpy(), t.permute(0, 2, 3, 1).numpy())
            else:
                buf.tensor[at:at + t.shape[0]].copy_(t)
            at += t.shape[0]
        spent['ring'] += t_b - t_a
        spent['sync'] += t_c - t_b
        spent['copy'] += time.perf_counter() - t_c
        return buf

    def put(entry):
        while not stop.is_set():
            try:
                q.put(entry, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    spent = {'load': 0.0, 'stage': 0.0, 'items': 0, 'ring': 0.0, 'sync': 0.0, 'copy': 0.0}

    def timed_iter():
        it = iter(iterable)
        while True:
            t0 = time.perf_counter()
            try:
                item = next(it)
            except StopIteration:
                return
            spent['load'] += time.perf_counter() - t0
            spent['items'] += 1
            yield item

    def worker():
        try:
            for item in (timed_iter() if timing else iterable):
                t_stage = time.perf_counter()
                held = []
                group = item if isinstance(item, BatchGroup) else None
                if group is not None:
                    keys = list(group[0].keys())
                    direct = [k for k in pin_entries if torch.is_tensor(group[0].get(k)) and group[0][k].is_floating_point()
                              and not group[0][k].is_cuda] if (pin and len(group) > 1) else []
                    rest = merge_batches([{k: v for k, v in b.items() if k not in direct} for b in group])
                    item = {}
                    for k in keys:
                        if k in direct:                   # merged by the copy into the staging buffer
                            buf = staged(k, [b[k] for b in group])
                            if buf is None:
                                return
                            item[k] = buf.tensor
                            held.append(buf)
                        else:
                            item[k] = rest[k]
                if pin and isinstance(item, dict):
                    item = dict(item)
                    for k in pin_entries:
                        v = item.get(k)
                        if torch.is_tensor(v) and v.is_floating_point() and not v.is_cuda and not v.is_pinned():
                            buf = staged(k, v)
                            if buf is None:
                                return
                            item[k] = buf.tensor
                            held.append(buf)
                spent['stage'] += time.perf_counter() - t_stage
                if not put((item, held, None)):
                    return
            if timing:
                logging.info('loader thread: {items} items, {load:.3f} s loading + collating, {stage:.3f} s merging + staging (waiting for a staging buffer {ring:.3f} s, for its last copy {sync:.3f} s, copying {copy:.3f} s)'.format(**spent))
            put((done, [], None))
        except BaseException as exc:  # noqa: BLE001 - re-raised in the consumer
            put((done, [], exc))

    thread = threading.Thread(target=worker, daemon=True, name='rcu-loader')
    thread.start()

    released = [0]

    def releaser(held):
        def release():
            if not held:
                return
            event = None
            if torch.cuda.is_available():
                event = torch.cuda.Event()
                event.record()
            with cond:
                for buf in held:
                    buf.event = event
                    buf.busy = False
                    released[0] += 1
                    buf.released = released[0]
                cond.notify_all()
            held.clear()
        return release

    try:
        while True:
            item, held, exc = q.get()
            if item is done:
                if exc is not None:
                    raise exc
                return
            yield item, releaser(held)
    finally:
        stop.set()
        with cond:
            cond.notify_all()
        try:                       # unblock a worker waiting in q.put
            while True:
                q.get_nowait()
        except queue.Empty:
            pass


class _Download:
    """Device -> host copy of a batch's kept output entries on a side stream, into pinned buffers (two sets, used in turn), started
    right behind the batch's kernels: the copy of batch k runs beside the kernels of batch k + 1 instead of making the host wait for
    both.  ``wait()`` returns channel-last numpy arrays, as the reference's ``convert_fn`` does (loops.py:214-220)."""

    _buffers = {}

    def __init__(self, tensors: dict, stream, slot, outputs_ready=None):
        self.done = torch.cuda.Event()
        self.arrays = {}
        ready = torch.cuda.Event()
        ready.record()                                  # behind the batch's kernels on the compute stream
        with torch.cuda.stream(stream):
            stream.wait_event(ready)
            if outputs_ready is not None:               # outputs a step finalised on a stream of its own (the root of a sharded predict step)
                stream.wait_event(outputs_ready)
            for key, value in tensors.items():
                value = steps_mod.channel_to_end(value)
                if not value.is_cuda:                   # an entry a step left on the host (labels kept for the subject steps)
                    self.arrays[key] = value
                    continue
                # one pinned buffer per (slot, entry): a batch of another shape replaces it (images of many sizes must not pile
                # pinned memory up; the arrays of the previous batch of this slot have been consumed by then)
                host = self._buffers.get((slot, key))
                if host is None or host.shape != value.shape or host.dtype != value.dtype:
                    host = self._buffers[(slot, key)] = torch.empty(value.shape, dtype=value.dtype, pin_memory=True)
                host.copy_(value, non_blocking=True)
                value.record_stream(stream)
                self.arrays[key] = host
            self.done.record(stream)

    def wait(self):
        self.done.synchronize()
        return {k: v.numpy() for k, v in self.arrays.items()}


def _finish_oldest_now(pixels, pipelined, budget, max_inflight):
    """The test loop's run-ahead rule.  ``pixels``: samples x height x width of the batches whose GPU work is enqueued, oldest first.  The oldest
    is finished (its outputs awaited, its subjects assembled / evaluated / written) once the batches enqueued BEHIND it are worth ``budget``
    pixels -- the GPU then has work for as long as the host spends on that batch -- or more than ``max_inflight`` batches are enqueued; the plain
    loop finishes every batch at once."""
    if not pixels:
        return False
    if not pipelined or len(pixels) > max_inflight:
        return True
    return len(pixels) > 1 and sum(pixels[1:]) >= budget


def _with_last_flag(iterable):
    """(item, is_last) pairs: one item of lookahead."""
    it = iter(iterable)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, False
        cur = nxt
    yield cur, True


class Test:
    """``pipelined`` (default: on for a CUDA device; the constructor argument, or the YAML key ``others.pipelined`` the scripts pass on,
    switches it off): the next batches are loaded and their GPU work
    enqueued while the outputs of batch k come to the host, so "batch k + 1 start" fires before "subjects of batch k" / "batch k
    end"; with ``pipelined=False`` every callback comes in the reference's order (loops.py:176-235).
    How far the loop runs ahead: until ``INFLIGHT_PIXELS`` (two BraTS volumes) worth of batches is enqueued behind the batch being finished,
    at most ``MAX_INFLIGHT`` batches (two volume-sized batches, ten of 32 slices; ``max_inflight`` / ``inflight_pixels`` of the constructor --
    YAML ``others.max_inflight`` -- bound it for memory-constrained runs: 1 = one batch ahead).  Output entries a run does not keep
    (``entries``) are dropped from the batch context as soon as the kept ones are on their way to the host, so a batch in flight holds its
    kept entries only.  One batch ahead (rounds 2-3) hides the host's work on a batch behind the next batch's kernels only
    when batches are volume-sized: with the shipped ``batch_size: 32`` a batch is 24 ms of GPU work, and the batch that completes a subject
    costs the host 100-190 ms (assembly, the metric seam, argmax, hand-over to the NIfTI writers; tools/loop_timeline.py) -- the GPU idled
    a third of the time; ten batches ahead keep it busy (0.185-0.192 -> 0.133-0.155 s per subject, tools/script_throughput.py 16 20 32 0;
    0.12 with coalescing, which also makes the launches volume-sized).
    ``coalesce`` (the constructor argument; the scripts pass the YAML key ``others.coalesce_pixels`` on and, since round 6, default to
    ``COALESCE_PIXELS`` -- this class itself defaults to 0 = off, as the reference, whose loop never regroups batches): consecutive loader
    batches are merged up to that many samples x height x width before the steps run -- ``COALESCE_PIXELS`` = one BraTS volume, 160 x 192 x 128,
    is what fills the GPU (five batches of the shipped ``batch_size: 32`` become one step of 160 slices: the kernels' small levels run 1.3-2x
    faster per slice).  What it changes: the YAML batch_size no longer is the step's batch; steps and hooks see the MERGED batch (fewer
    ``on_test_batch_*`` calls, renumbered ``batch_index``, fewer ``batch_metrics`` entries); and the activation workspace grows to that of the
    merged batch times the pass group (24 GB per lane for four passes of 160 slices).  What it does not change: the MC samples -- the seeded
    Dropout2d masks of a stochastic step are keyed by a slice's GLOBAL index (``BatchContext.sample_offset``: this loop counts the slices it
    hands out), not by the batch (rounds 1-5), so a run's files are the same for every ``batch_size`` and deterministic steps give the same
    files byte for byte with or without coalescing."""
    __test__ = False
    COALESCE_PIXELS = 160 * 192 * 128
    INFLIGHT_PIXELS = 2 * 160 * 192 * 128
    MAX_INFLIGHT = 12

    def __init__(self, steps: list, subject_steps: list = None, subject_assembler=None, entries: tuple = None,
                 convert_fn=tensor_to_numpy, pipelined=None, coalesce=None, max_inflight=None, inflight_pixels=None, loader_timing=False):
        self.steps = steps
        self.subject_steps = subject_steps or []
        self.subject_assembler = subject_assembler
        self.entries = entries
        self.convert_fn = convert_fn
        self.pipelined = pipelined
        self.coalesce = coalesce
        self.max_inflight = self.MAX_INFLIGHT if max_inflight is None else max(1, int(max_inflight))
        self.inflight_pixels = self.INFLIGHT_PIXELS if inflight_pixels is None else max(1, int(inflight_pixels))
        self.loader_timing = bool(loader_timing)

    def __call__(self, context, build_test, hook: TestLoopHook = TestLoopHook()):
        hook.on_startup()
        context.setup_directory()
        context.setup_logging()
        seed = context.get_seed()
        if seed is not None:
            context.do_seed(seed)
        context.load_test_data(build_test)
        context.load_from_checkpoint(context.get_test_at())
        hook.end_startup(context)

        task_context = context.get_task_context()
        hook.on_test_start(task_context, context)
        # Pipelined form of the reference's loop (loops.py:176-235): batch k + 1 is loaded while batch k computes, and the outputs of
        # batch k come to the host -- and its subjects are assembled, evaluated and written -- while batch k + 1 computes.  Per
        # batch the order of the callbacks is the reference's (batch start, steps, subject start / steps / end, batch end); only
        # "batch k + 1 start" now comes before "subjects of batch k".  A custom convert_fn or a CPU device keeps the plain order.
        pipelined = True if self.pipelined is None else self.pipelined
        # (a loop without an assembler -- a rank other than the root of a sharded run: same loader, same steps, nothing to finish -- is pipelined
        # too: its batches come through the loader thread's pinned staging instead of a synchronous copy from pageable memory per batch)
        pipelined = bool(pipelined) and (self.convert_fn is tensor_to_numpy and getattr(context.device, 'type', 'cpu') == 'cuda')
        side = torch.cuda.Stream(device=context.device) if pipelined else None
        self._subject_stream = torch.cuda.Stream(device=context.device) if pipelined else None
        loader = task_context.data.loader
        coalesce = int(self.coalesce or 0)
        if coalesce > 0 and getattr(context.device, 'type', 'cpu') == 'cuda':
            loader = coalesced(loader, coalesce, lazy=pipelined)
        inflight = collections.deque()        # (batch context, download, pixels, download slot) of the batches whose GPU work is enqueued
        free_slots, slots = [], 0             # download slots (a set of pinned buffers each): taken per batch, back when it is finished

        def finish_oldest():
            batch_context, download, _, slot = inflight.popleft()
            self._finish_batch(batch_context, download, task_context, context, hook)
            free_slots.append(slot)

        batches = prefetch(loader, depth=self.max_inflight if pipelined else 2, pin=pipelined, timing=self.loader_timing)
        try:
            samples_seen = 0                  # the run's stream of slices / images: what the seeded Dropout2d masks are keyed by (steps.McPredictStep)
            for i, ((batch, release), last) in enumerate(_with_last_flag(batches)):
                batch_context = BatchContext(batch, i, sample_offset=samples_seen)
                samples_seen += _batch_samples(batch)
                batch_context.more['last_batch'] = last
                hook.on_test_batch_start(batch_context, task_context, context)
                if free_slots:
                    slot = free_slots.pop()
                else:
                    slot, slots = slots, slots + 1
                download = self._run_steps(batch_context, task_context, context, side, slot)
                release()                     # the step's host-to-device copies are enqueued: the staging buffers may go back
                # (pixels of the batch = the measure of its GPU work the run-ahead is budgeted in; a batch without an image tensor counts as a full budget)
                inflight.append((batch_context, download, _batch_pixels(batch) or self.inflight_pixels, slot))
                # finish the oldest batch once enough work is enqueued behind it to cover the host's share of finishing it
                while _finish_oldest_now([e[2] for e in inflight], pipelined, self.inflight_pixels, self.max_inflight):
                    finish_oldest()
            while inflight:
                finish_oldest()
        finally:
            batches.close()                   # stops the loader thread, also when a step or a hook raised
        hook.on_test_end(task_context, context)
        hook.on_termination(context)

    def _kept(self, batch_context):
        return {key: value for key, value in batch_context.output.items()
                if (self.entries is None or key in self.entries) and isinstance(value, torch.Tensor)}

    def _run_steps(self, batch_context, task_context, context, side, slot):
        for batch_step in self.steps:
            batch_step(batch_context, task_context, context)
        if batch_context.metrics:
            task_context.history.add(batch_context.metrics, 'batch_metrics')
        if side is None or self.subject_assembler is None:
            return None
        kept = self._kept(batch_context)
        download = _Download(kept, side, slot, batch_context.more.get('outputs_ready'))
        # up to max_inflight batches are enqueued before this one is finished: the entries nobody keeps must not stay on the GPU that long
        batch_context.output = {key: value for key, value in batch_context.output.items()
                                if key in kept or not (isinstance(value, torch.Tensor) and value.is_cuda)}
        return download

    def _finish_batch(self, batch_context, download, task_context, context, hook):
        if self.subject_assembler is not None:
            if download is not None:
                to_assemble = download.wait()
            else:
                to_assemble = {}
                steps_mod.wait_for_outputs(batch_context)
                for key, value in self._kept(batch_context).items():
                    value = steps_mod.channel_to_end(value)
                    to_assemble[key] = self.convert_fn(value) if self.convert_fn else value
            last = batch_context.more.get('last_batch')
            if last is None:
                last = batch_context.batch_index == task_context.data.nb_batches - 1
            self.subject_assembler.add_batch(to_assemble, batch_context.input, last_batch=last)

            # Pipelined: the subject steps and hooks run with a stream of their own as the current one.  Their GPU work (the metric
            # seam: Dice counts of the assembled subject) is independent of the batch steps', but on the compute stream every small
            # host-to-device copy of theirs would queue behind the NEXT batch's forward passes, which are already enqueued there.
            with (torch.cuda.stream(self._subject_stream) if download is not None else contextlib.nullcontext()):
                for subject_index in sorted(self.subject_assembler.subjects_ready, key=str):
                    subject_data = self.subject_assembler.get_assembled_subject(subject_index)
                    subject_context = SubjectContext(subject_index, subject_data)
                    hook.on_test_subject_start(subject_context, task_context, context)
                    for subject_step in self.subject_steps:
                        subject_step(subject_context, task_context, context)
                    if subject_context.metrics:
                        task_context.history.add(subject_context.metrics, 'subject_metrics')
                    hook.on_test_subject_end(subject_context, task_context, context)
        hook.on_test_batch_end(batch_context, task_context, context)

    def _test_batch(self, batch_context, task_context, context, hook):
        """One batch in the reference's order (loops.py:196-235), without the pipeline."""
        self._run_steps(batch_context, task_context, context, None, 0)
        self._finish_batch(batch_context, None, task_context, context, hook)
