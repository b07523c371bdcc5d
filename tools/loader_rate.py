#!/usr/bin/env python3
"""How fast the drop-in BraTS script's loader alone delivers batches (no GPU work): the ceiling of an N-GPU run of the script, where every rank
reads every batch.    python tools/loader_rate.py [subjects, default 16] [batch_size, default 32]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402
from rcu_amd import data as data_mod  # noqa: E402
from rcu_amd import loops, nifti  # noqa: E402
from test_script_surface_cpu import BRATS_MC_YAML  # noqa: E402


def main():
    n_subjects = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    tmp = tempfile.mkdtemp(prefix='rcu_loader_')
    x, _, target = bench.make_volume(20)
    names = []
    for i in range(n_subjects):
        name = 'Brats18_SYN_{:03d}_1'.format(i)
        props = nifti.ImageProperties((bench.WIDTH, bench.HEIGHT, bench.SLICES), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
        data_mod.write_volume(os.path.join(tmp, 'ds'), name, (x + 0.01 * i).permute(0, 2, 3, 1).numpy(), target.numpy(), props)
        names.append(name)
    split = os.path.join(tmp, 'split.json')
    with open(split, 'w') as f:
        json.dump({'train': [], 'valid': [], 'test': names}, f)
    text = BRATS_MC_YAML.format(test_dir=os.path.join(tmp, 'out'), model_dir=tmp, split=split, dataset=os.path.join(tmp, 'ds'))
    text = text.replace('batch_size: 32', 'batch_size: {}'.format(batch))
    cfg = os.path.join(tmp, 'cfg.yaml')
    with open(cfg, 'w') as f:
        f.write(text)
    context = loops.TorchTestContext('cpu')
    context.load_from_config(cfg)
    build = data_mod.BuildData(build_dataset=data_mod.BuildVolumeDataset())
    for label in ('first pass over the files', 'second pass'):
        context.load_test_data(build)
        loader = context.test_data.loader
        t0 = time.perf_counter()
        count = 0
        for item in loader:
            count += 1
        dt = time.perf_counter() - t0
        print('{}: {} batches of {} slices in {:.3f} s = {:.1f} ms per batch, {:.1f} ms per subject ({:.0f} subjects/s)'.format(
            label, count, batch, dt, dt / count * 1e3, dt / n_subjects * 1e3, n_subjects / dt))


if __name__ == '__main__':
    main()
