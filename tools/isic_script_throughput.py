#!/usr/bin/env python3
"""End-to-end wall time of the drop-in isic_test_default script (YAML -> jpg / png dataset -> Test loop -> NIfTI + metrics) on synthetic
256 x 256 images with the full-width ISIC model (BASELINE config 2: batch 32, T = 20), and where the loop's main thread spends it:
    python tools/isic_script_throughput.py [images, default 192] [mc, default 20] [batch_size, default 32]"""
import collections
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from rcu_amd import evaluation, loops, nifti, scripts, steps  # noqa: E402
from rcu_amd import model as model_mod  # noqa: E402
from rcu_amd import management as mgt  # noqa: E402
from test_gpu_scripts import ISIC_MC_YAML  # noqa: E402  (the reference's YAML layout)

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


def main():
    from PIL import Image
    n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    mc = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    tmp = tempfile.mkdtemp(prefix='rcu_isic_e2e_')
    prefix = os.path.join(tmp, 'isic', 'ISIC-2017_Test_v2')
    img_dir, lab_dir = prefix + '_Data', prefix + '_Part1_GroundTruth'
    os.makedirs(img_dir)
    os.makedirs(lab_dir)
    rng = np.random.RandomState(6)
    for i in range(n_images):
        id_ = 'ISIC_{:07d}'.format(i)
        Image.fromarray(rng.randint(0, 255, (bench.ISIC_HEIGHT, bench.ISIC_WIDTH, 3)).astype(np.uint8)).save(os.path.join(img_dir, id_ + '.jpg'))
        lab = np.zeros((bench.ISIC_HEIGHT, bench.ISIC_WIDTH), np.uint8)
        lab[40:40 + (i % 100) + 20, 30:200] = 255
        Image.fromarray(lab).save(os.path.join(lab_dir, id_ + '_segmentation.png'))
    model = bench.make_model(20, torch.device('cuda'), params=bench.ISIC_PARAMS)
    mf = mgt.ModelFiles(os.path.join(tmp, 'train'), 'isic')
    mgt.save_model(mf, 'unet', bench.ISIC_PARAMS, {k: v.cpu() for k, v in model.state_dict().items()})
    text = ISIC_MC_YAML.format(test_dir=os.path.join(tmp, 'out'), model_dir=mf.model_dir, dataset=prefix)
    text = text.replace('mc: 2', 'mc: {}'.format(mc)).replace('batch_size: 1', 'batch_size: {}'.format(batch))
    cfg = os.path.join(tmp, 'test_isic_baseline_mc.yaml')
    with open(cfg, 'w') as f:
        f.write(text)
    for owner, name, label in ((loops.Test, '_run_steps', None), (loops.Test, '_finish_batch', None), (loops._Download, 'wait', '_Download.wait'),
                               (scripts.EvalSubjectStep, '__call__', 'EvalSubjectStep.__call__'), (evaluation, 'confusion_matrx', None),
                               (nifti, 'argmax_last', None), (scripts.WriteHook, 'on_test_subject_end', 'WriteHook.on_test_subject_end'),
                               (loops.Subject2dAssembler, 'add_batch', 'Subject2dAssembler.add_batch'),
                               (steps.McPredictStep, '__call__', 'McPredictStep.__call__'), (nifti, 'do_work', None), (nifti, 'write_subject', None),
                               (os, 'symlink', 'os.symlink'), (scripts.ConfusionOnDeviceStep, '__call__', 'ConfusionOnDeviceStep.__call__'),
                               (loops._Download, '__init__', '_Download.__init__'), (steps.MultiPredictionSummary, '__call__', 'MultiPredictionSummary.__call__'),
                               (loops.ConsoleTestLogHook, 'on_test_subject_end', 'ConsoleTestLogHook.on_test_subject_end'),
                               (steps, 'reserve_canonical_plans', None), (steps, '_images_to_device', None), (steps, 'set_dropout_mode', None),
                               (steps, 'softmax', None), (steps, 'merge_statistics', None), (steps.McStatistics, '__init__', 'McStatistics.__init__'),
                               (steps.StreamLanes, 'begin', 'StreamLanes.begin'), (steps.StreamLanes, 'run', 'StreamLanes.run'),
                               (steps.StreamLanes, 'end', 'StreamLanes.end'), (scripts.PrepareSubjectStep, '__call__', 'PrepareSubjectStep.__call__')):
        wrap(owner, name, label)
    for name in ('forward', 'forward_accumulate', 'seeded_masks', '_handle'):
        wrap(model_mod.UNet, name, 'UNet.' + name)
    stamps = []
    inner = scripts.WriteHook.on_test_subject_end

    def stamped(self, subject_context, task_context, context):
        inner(self, subject_context, task_context, context)
        stamps.append(time.perf_counter())

    scripts.WriteHook.on_test_subject_end = stamped
    t0 = time.perf_counter()
    scripts.test_default('isic', cfg, None)
    dt = time.perf_counter() - t0
    first = stamps[batch - 1] if len(stamps) > batch else stamps[0]
    steady = (t0 + dt - first) / max(len(stamps) - batch, 1)
    print('{} images, T={}, batch_size {}: {:.2f} s total; steady state (behind the first batch, incl. the final join): {:.4f} s per image = '
          '{:.0f} MC-sample-images/s ({:.1f} ms per batch of {}); start-up + first batch {:.2f} s'.format(
              n_images, mc, batch, dt, steady, mc / steady, steady * batch * 1e3, batch, first - t0))
    print('main-thread milliseconds per image (nested entries overlap their parents):')
    for label, (calls, seconds) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print('  {:<40} {:>6} calls  {:>8.2f} ms per image'.format(label, calls, seconds * 1e3 / n_images))


if __name__ == '__main__':
    main()
