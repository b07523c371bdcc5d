#!/usr/bin/env python3
"""Where the MAIN thread of the drop-in test script spends its time (tools/script_throughput.py's run, the shipped batch_size 32, no coalescing):
wall-clock accumulators around the host-side pieces of a batch and of a subject -- no profiler, so the numbers add up to the loop's own time.
    python tools/host_costs_probe.py [subjects, default 12] [coalesce_pixels, default 0] [confusion=0] [switch=<GIL switch interval, s>]"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
os.environ['RCU_SCRIPT_PROFILE'] = '0'
import threading  # noqa: E402

from rcu_amd import evaluation, loops, model, nifti, scripts, steps  # noqa: E402

acc = collections.defaultdict(lambda: [0, 0.0])
MAIN = threading.main_thread()


def wrap(owner, name, label=None):
    inner = getattr(owner, name)
    label = label or '{}.{}'.format(getattr(owner, '__name__', owner), name)

    def timed(*a, **k):
        if threading.current_thread() is not MAIN:
            return inner(*a, **k)
        t = time.perf_counter()
        try:
            return inner(*a, **k)
        finally:
            e = acc[label]
            e[0] += 1
            e[1] += time.perf_counter() - t
    setattr(owner, name, timed)


wrap(loops.Test, '_run_steps')
wrap(loops.Test, '_finish_batch')
wrap(loops._Download, 'wait')
wrap(loops._Download, '__init__', '_Download.__init__')
wrap(nifti, 'argmax_last')
wrap(nifti, 'write_subject')
wrap(evaluation, 'confusion_matrx')
wrap(evaluation, '_to_dev')
wrap(evaluation, '_zeros_like_map')
wrap(scripts.ConfusionOnDeviceStep, '__call__', 'ConfusionOnDeviceStep.__call__')
wrap(scripts.ConfusionOnDeviceStep, '_labels_of', 'ConfusionOnDeviceStep._labels_of')
wrap(steps, 'prediction_and_foreground')
wrap(evaluation, 'confusion_counts_on_device')
wrap(evaluation, '_uncertainty_counts_device')
wrap(steps, 'wait_for_outputs')
for name in ('_images_to_device', 'reserve_canonical_plans', 'set_dropout_mode', 'merge_statistics', 'pass_group_size', 'softmax'):
    if hasattr(steps, name):
        wrap(steps, name)
for name in ('begin', 'run', 'end'):
    wrap(steps.StreamLanes, name, 'StreamLanes.' + name)
wrap(steps.McStatistics, '__init__', 'McStatistics.__init__')
wrap(model.UNet, 'sample_masks')
wrap(model.UNet, 'group_masks')
wrap(evaluation, 'uncertainty_counts')
wrap(steps.McPredictStep, '_launch_masks')
wrap(steps.McPredictStep, '__call__', 'McPredictStep.__call__')
wrap(steps.MultiPredictionSummary, '__call__', 'MultiPredictionSummary.__call__')
wrap(model.UNet, 'forward_accumulate')
wrap(model.UNet, 'forward')
wrap(scripts.EvalSubjectStep, '__call__', 'EvalSubjectStep.__call__')
wrap(scripts.WriteHook, 'on_test_subject_end', 'WriteHook.on_test_subject_end')
for cls_name in ('SubjectAssembler', 'Subject2dAssembler'):
    if hasattr(loops, cls_name):
        wrap(getattr(loops, cls_name), 'add_batch', cls_name + '.add_batch')
        wrap(getattr(loops, cls_name), 'get_assembled_subject', cls_name + '.get_assembled_subject')

for arg in list(sys.argv):
    if arg == 'confusion=0':          # others.device_confusion: false -- the subjects' Dice counts at subject level (three synchronous GPU operations)
        inner_other = scripts._other
        scripts._other = lambda context, key, default=None: False if key == 'device_confusion' else inner_other(context, key, default)
        sys.argv.remove(arg)
    if arg.startswith('switch='):       # sys.setswitchinterval: how long a thread that wants the GIL waits before the holder is told to yield
        sys.setswitchinterval(float(arg.split('=')[1]))
        sys.argv.remove(arg)
print('GIL switch interval {} s'.format(sys.getswitchinterval()))

import script_throughput  # noqa: E402

subjects = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sys.argv = ['x', str(subjects), '20', '32', sys.argv[2] if len(sys.argv) > 2 else '0']
script_throughput.main()
print('main-thread seconds per subject ({} subjects; nested entries overlap their parents):'.format(subjects))
for label, (calls, seconds) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('  {:<44} {:>6} calls  {:>8.1f} ms per subject'.format(label, calls, seconds * 1e3 / subjects))
