#!/usr/bin/env python3
"""End-to-end wall time of the drop-in brats_test_default script (YAML -> dataset -> Test loop -> NIfTI + metrics) on
full-size synthetic subjects, with a cProfile breakdown of the host side:
    python tools/script_throughput.py [subjects] [mc] [batch_size] [coalesce_pixels, 0 = off] [timing: the loader thread logs its time split]
RCU_SCRIPT_NATIVE=1: subjects of the reference's real BraTS shape, 155 slices of 240 x 240 (bench.NATIVE_*), instead of the benchmark's 160 x 192 x 128."""
import cProfile
import json
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from rcu_amd import data as data_mod  # noqa: E402
from rcu_amd import management as mgt  # noqa: E402
from rcu_amd import nifti, scripts  # noqa: E402
from test_script_surface_cpu import BRATS_MC_YAML  # noqa: E402  (the reference's YAML layout)


def main():
    n_subjects = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    mc = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    tmp = tempfile.mkdtemp(prefix='rcu_e2e_')
    if os.environ.get('RCU_SCRIPT_NATIVE') == '1':
        bench.SLICES, bench.HEIGHT, bench.WIDTH = bench.NATIVE_SLICES, bench.NATIVE_HEIGHT, bench.NATIVE_WIDTH
    x, mask, target = bench.make_volume(20, bench.SLICES, bench.HEIGHT, bench.WIDTH, bench.SLICES)
    names = []
    for i in range(n_subjects):
        name = 'Brats18_SYN_{:03d}_1'.format(i)
        images = (x + 0.01 * i).permute(0, 2, 3, 1).numpy()            # [D, H, W, C]
        props = nifti.ImageProperties((bench.WIDTH, bench.HEIGHT, bench.SLICES), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
        data_mod.write_volume(os.path.join(tmp, 'ds'), name, images, target.numpy(), props)
        names.append(name)
    aleatoric = os.environ.get('RCU_SCRIPT_ALEATORIC') == '1'      # bin-dl/brats_test_aleatoric.py's surface: sigma head, one deterministic pass per batch
    model = bench.make_model(20, torch.device('cuda'), sigma_out=aleatoric)
    mf = mgt.ModelFiles(os.path.join(tmp, 'train'), 'syn')
    mgt.save_model(mf, 'unet', dict(bench.MODEL_PARAMS, sigma_out=True) if aleatoric else bench.MODEL_PARAMS,
                   {k: v.cpu() for k, v in model.state_dict().items()})
    split = os.path.join(tmp, 'split.json')
    with open(split, 'w') as f:
        json.dump({'train': [], 'valid': [], 'test': names}, f)
    text = BRATS_MC_YAML.format(test_dir=os.path.join(tmp, 'out'), model_dir=mf.model_dir, split=split,
                                dataset=os.path.join(tmp, 'ds'))
    text = text.replace('mc: 20', 'mc: {}'.format(mc)).replace('batch_size: 32', 'batch_size: {}'.format(batch))
    # loader batches merged to one volume per step: opt-in (rcu_amd.loops.Test), through the YAML key the scripts pass on; argv[4] = 0 keeps
    # the reference's batches (one step per 32 slices)
    coalesce = int(sys.argv[4]) if len(sys.argv) > 4 else bench.SLICES * bench.HEIGHT * bench.WIDTH
    # (0 = the loader's batches as they are: since round 6 coalescing is the scripts' default, so the key is written either way)
    text = text.replace('  others:\n', '  others:\n    coalesce_pixels: {}\n'.format(coalesce), 1)
    assert 'coalesce_pixels' in text
    if len(sys.argv) > 5 and sys.argv[5] == 'timing':
        text = text.replace('  others:\n', '  others:\n    loader_timing: true\n', 1)
    # RCU_SCRIPT_ENSEMBLE=K: bin-dl/brats_test_ensemble.py's surface instead -- K members (the first is the config's model_dir), no MC passes
    members = int(os.environ.get('RCU_SCRIPT_ENSEMBLE', '0'))
    if members > 1:
        extra = []
        for k in range(1, members):
            mk = mgt.ModelFiles(os.path.join(tmp, 'train_{}'.format(k)), 'syn{}'.format(k))
            mgt.save_model(mk, 'unet', bench.MODEL_PARAMS, {key: v.cpu() for key, v in bench.make_model(20 + k, torch.device('cuda')).state_dict().items()})
            extra.append(mk.model_dir)
        text = text.replace('    mc: {}\n'.format(mc), '    model_dir:\n' + ''.join('    - {}\n'.format(d) for d in extra) + '    test_at: best\n')
        assert '    model_dir:\n    - ' in text
    cfg = os.path.join(tmp, 'test_brats_baseline_mc.yaml')
    with open(cfg, 'w') as f:
        f.write(text)
    stamps = []
    inner = scripts.WriteHook.on_test_subject_end

    def stamped(self, subject_context, task_context, context):
        inner(self, subject_context, task_context, context)
        stamps.append(time.perf_counter())

    scripts.WriteHook.on_test_subject_end = stamped
    profile = os.environ.get('RCU_SCRIPT_PROFILE', '1') != '0'
    prof = cProfile.Profile()
    t0 = time.perf_counter()
    if profile:
        prof.enable()
    if members > 1:
        scripts.test_ensemble('brats', cfg)
    elif aleatoric:
        scripts.test_aleatoric('brats', cfg)
    else:
        scripts.test_default('brats', cfg, None)
    if profile:
        prof.disable()
    dt = time.perf_counter() - t0
    unit = 'member' if members > 1 else 'MC-sample'
    mc = members if members > 1 else (1 if aleatoric else mc)
    print('{} subjects, T={}, batch_size {}: {:.2f} s total, {:.2f} s per subject ({:.1f} {}-volumes/s end to end, start-up and '
          'the final join of the writers included; subjects of {} x {} x {})'.format(n_subjects, mc, batch, dt, dt / n_subjects, mc * n_subjects / dt, unit,
                                                                                  bench.SLICES, bench.HEIGHT, bench.WIDTH))
    if len(stamps) > 2:
        # steady state: from the hand-over of the first subject to the writers to the end of the run (the last subject's files joined)
        steady = (t0 + dt - stamps[0]) / (len(stamps) - 1)
        print('steady state (subjects 2..{}, incl. the final join): {:.3f} s per subject = {:.1f} {}-volumes/s; start-up + first '
              'subject {:.2f} s'.format(len(stamps), steady, mc / steady, unit, stamps[0] - t0))
    if not profile:
        return
    st = pstats.Stats(prof)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats('tottime').print_stats(14)


if __name__ == '__main__':
    main()
