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
        os.makedirs(self.test_dir, exist_ok=True)
        cfg.save(os.path.join(self.test_dir, 'config' + os.path.splitext(self.config_file_path)[1]), self.config)
        if self.config.split:
            shutil.copy(self.config.split, os.path.join(self.test_dir, os.path.basename(self.config.split)))

    def setup_logging(self):
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
        for b, (si, k) in enumerate(zip(batch['subject_index'], batch['slice_index'])):
            si, k = int(si), int(k)
            store = self.volumes.setdefault(si, {})
            for key, value in to_assemble.items():
                if key not in store:
                    depth = int(batch['shape'][b][0])
                    store[key] = np.zeros((depth,) + value.shape[1:], dtype=value.dtype)
                store[key][k] = value[b]
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


def prefetch(iterable, depth=2, pin=False):
    """Iterate ``iterable`` from a background thread, ``depth`` items ahead: the loader's work for the next batches (file reads,
    decompression -- zlib and numpy release the GIL -- transforms, collation) overlaps the GPU work of the current one.  ``pin``:
    floating-point tensors of a dict batch are moved to pinned host memory there too, so that the step's host-to-device copy is
    asynchronous and the test loop keeps the GPU's queue filled instead of waiting for each copy."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    done = object()

    # pinned staging buffers, allocated once per (entry, shape) and used in turn: depth queued + one with the consumer + one
    # whose copy to the device may still be in flight (pinning per batch costs more than the copy it speeds up)
    ring, turn = {}, [0]

    def staged(key, v):
        bufs = ring.setdefault((key, tuple(v.shape), v.dtype), [])
        if len(bufs) < depth + 2:
            bufs.append(torch.empty(v.shape, dtype=v.dtype, pin_memory=True))
        buf = bufs[turn[0] % len(bufs)]
        buf.copy_(v)
        return buf

    def worker():
        try:
            for item in iterable:
                if pin and isinstance(item, dict):
                    item = {k: (staged(k, v) if torch.is_tensor(v) and v.is_floating_point() and not v.is_cuda else v)
                            for k, v in item.items()}
                    turn[0] += 1
                q.put((item, None))
            q.put((done, None))
        except BaseException as exc:  # noqa: BLE001 - re-raised in the consumer
            q.put((done, exc))

    threading.Thread(target=worker, daemon=True, name='rcu-loader').start()
    while True:
        item, exc = q.get()
        if item is done:
            if exc is not None:
                raise exc
            return
        yield item


class _Download:
    """Device -> host copy of a batch's kept output entries on a side stream, into pinned buffers (two sets, used in turn), started
    right behind the batch's kernels: the copy of batch k runs beside the kernels of batch k + 1 instead of making the host wait for
    both.  ``wait()`` returns channel-last numpy arrays, as the reference's ``convert_fn`` does (loops.py:214-220)."""

    _buffers = {}

    def __init__(self, tensors: dict, stream, slot):
        self.done = torch.cuda.Event()
        self.arrays = {}
        ready = torch.cuda.Event()
        ready.record()                                  # behind the batch's kernels on the compute stream
        with torch.cuda.stream(stream):
            stream.wait_event(ready)
            for key, value in tensors.items():
                value = steps_mod.channel_to_end(value)
                if not value.is_cuda:                   # an entry a step left on the host (labels kept for the subject steps)
                    self.arrays[key] = value
                    continue
                buf_key = (slot, key, tuple(value.shape), value.dtype)
                host = self._buffers.get(buf_key)
                if host is None:
                    host = self._buffers[buf_key] = torch.empty(value.shape, dtype=value.dtype, pin_memory=True)
                host.copy_(value, non_blocking=True)
                value.record_stream(stream)
                self.arrays[key] = host
            self.done.record(stream)

    def wait(self):
        self.done.synchronize()
        return {k: v.numpy() for k, v in self.arrays.items()}


class Test:
    __test__ = False

    def __init__(self, steps: list, subject_steps: list = None, subject_assembler=None, entries: tuple = None,
                 convert_fn=tensor_to_numpy):
        self.steps = steps
        self.subject_steps = subject_steps or []
        self.subject_assembler = subject_assembler
        self.entries = entries
        self.convert_fn = convert_fn

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
        pipelined = (self.convert_fn is tensor_to_numpy and self.subject_assembler is not None and
                     getattr(context.device, 'type', 'cpu') == 'cuda')
        side = torch.cuda.Stream(device=context.device) if pipelined else None
        waiting = None
        for i, batch in enumerate(prefetch(task_context.data.loader, pin=pipelined)):
            batch_context = BatchContext(batch, i)
            hook.on_test_batch_start(batch_context, task_context, context)
            download = self._run_steps(batch_context, task_context, context, side, i & 1)
            if waiting is not None:
                self._finish_batch(*waiting, task_context, context, hook)
            waiting = (batch_context, download)
            if not pipelined:
                self._finish_batch(*waiting, task_context, context, hook)
                waiting = None
        if waiting is not None:
            self._finish_batch(*waiting, task_context, context, hook)
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
        return _Download(self._kept(batch_context), side, slot)

    def _finish_batch(self, batch_context, download, task_context, context, hook):
        if self.subject_assembler is not None:
            if download is not None:
                to_assemble = download.wait()
            else:
                to_assemble = {}
                for key, value in self._kept(batch_context).items():
                    value = steps_mod.channel_to_end(value)
                    to_assemble[key] = self.convert_fn(value) if self.convert_fn else value
            last = batch_context.batch_index == task_context.data.nb_batches - 1
            self.subject_assembler.add_batch(to_assemble, batch_context.input, last_batch=last)

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
