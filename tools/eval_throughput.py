#!/usr/bin/env python3
"""End-to-end time of the evaluation script's hot loop (bin-eval/eval_uncertainty.py:32-47 of the reference: subjects x actions) on
synthetic BraTS-sized subjects written to disk as the test scripts write them (.nii.gz probability / prediction volumes + a ground-truth
tree with T2 and segmentation): seconds per subject for the product's loop, split into file reading / staging / upload + kernels / CSV
rows, next to the reference-ordered loop of the same package (--plain) and to the numpy restatement of the reference (oracle/, on a few
subjects).

    python tools/eval_throughput.py [--subjects 32] [--batch 8] [--act minmax ece_dice calib bnf_ue] [--oracle-subjects 3] [--out FILE.json]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

D, H, W = 160, 192, 128


def make_subjects(root, count):
    """count subjects: <root>/run/<s>_{probabilities,prediction}.nii.gz and <root>/gt/HGG/<s>/<s>_{flair,t2,seg}.nii.gz"""
    from rcu_amd import nifti
    rng = np.random.RandomState(1)
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, D), np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing='ij')
    brain = ((zz / 0.9) ** 2 + (yy / 0.85) ** 2 + (xx / 0.8) ** 2) < 1.0
    os.makedirs(os.path.join(root, 'run'))
    names = []
    for i in range(count):
        name = 'Brats18_SYN_{:03d}_1'.format(i)
        lesion = ((zz / 0.35) ** 2 + ((yy - 0.1 + 0.002 * i) / 0.3) ** 2 + ((xx + 0.1) / 0.3) ** 2) < 1.0
        # a probability map shaped like a network's: confident almost everywhere, a soft rim around the lesion
        p = np.clip(lesion * 0.9 + 0.05 + 0.2 * rng.standard_normal((D, H, W)).astype(np.float32) * (np.abs(rng.standard_normal((D, H, W))) > 1.5), 0, 1)
        p = (p * brain).astype(np.float32)
        gt = os.path.join(root, 'gt', 'HGG', name)
        os.makedirs(gt)
        nifti.write(os.path.join(root, 'run', name + '_probabilities.nii.gz'), p)
        nifti.write(os.path.join(root, 'run', name + '_prediction.nii.gz'), (p > 0.5).astype(np.uint8))
        nifti.write(os.path.join(gt, name + '_flair.nii.gz'), np.zeros((2, 2, 2), np.float32))     # (only its name is used)
        nifti.write(os.path.join(gt, name + '_t2.nii.gz'), (brain * (1 + rng.rand(D, H, W))).astype(np.float32))
        nifti.write(os.path.join(gt, name + '_seg.nii.gz'), (lesion * 4).astype(np.uint8))
        names.append(name)
    return names


def oracle_subject(root, name, actions):
    """What the reference's loop does for one subject and these actions, by the numpy restatement (oracle/calib_oracle.py)."""
    from oracle import calib_oracle as co
    from rcu_amd import nifti
    t0 = time.perf_counter()
    p = nifti.read(os.path.join(root, 'run', name + '_probabilities.nii.gz'))[0]
    pred = nifti.read(os.path.join(root, 'run', name + '_prediction.nii.gz'), np.uint8)[0]
    gt = os.path.join(root, 'gt', 'HGG', name)
    tgt = (nifti.read(os.path.join(gt, name + '_seg.nii.gz'), np.uint8)[0] > 0).astype(np.uint8)
    mask = nifti.read(os.path.join(gt, name + '_t2.nii.gz'))[0] > 0
    t_read = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = {}
    if 'minmax' in actions:
        out['minmax'] = (p.min(), p.max())
    pair = co.add_background_probability(p)
    if 'ece_dice' in actions:
        out['ece'] = co.ece_binary(pair, tgt, mask=mask)
        out['conf'] = co.confusion_counts(pred, tgt)
    if 'calib' in actions:
        bins = {}
        co.ece_binary(pair, tgt, mask=mask, out_bins=bins)
        co.confusion_counts(pred, tgt)
    if 'bnf_ue' in actions:
        unc = co.normalised_entropy(pair)
        for thr in co.UE_THRESHOLDS:
            co.correction_metrics(co.uncertainty_counts(pred.astype(bool), tgt.astype(bool), unc > thr))
    return t_read, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--subjects', type=int, default=32)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--act', nargs='+', default=['minmax', 'ece_dice', 'calib', 'bnf_ue'])
    ap.add_argument('--oracle-subjects', type=int, default=3)
    ap.add_argument('--out', default=None)
    ap.add_argument('--keep', action='store_true')
    args = ap.parse_args()
    import torch
    from rcu_amd import scripts
    root = tempfile.mkdtemp(prefix='rcu_eval_')
    try:
        t0 = time.perf_counter()
        names = make_subjects(root, args.subjects)
        t_make = time.perf_counter() - t0
        record = dict(subjects=args.subjects, voxels_per_subject=D * H * W, actions=args.act, batch_subjects=args.batch,
                      files_per_subject='probabilities f32 + prediction u8 (.nii.gz, written as the test scripts write them) + T2 f32 + seg u8',
                      dataset_written_in_s=t_make)
        # warm: library load, pinned staging, first launches
        scripts.eval_uncertainty('brats', {'baseline_mc': os.path.join(root, 'run')}, os.path.join(root, 'gt'), os.path.join(root, 'warm'),
                                 args.act[:], batch_subjects=args.batch)
        legs = {}
        for tag, kwargs in (('fused', dict(batch_subjects=args.batch)), ('fused_batch1', dict(batch_subjects=1)), ('plain', dict(fused=False))):
            timing = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            scripts.eval_uncertainty('brats', {'baseline_mc': os.path.join(root, 'run')}, os.path.join(root, 'gt'), os.path.join(root, 'eval_' + tag),
                                     args.act[:], timing=timing, **kwargs)
            dt = time.perf_counter() - t0
            leg = dict(total_s=dt, s_per_subject=dt / args.subjects)
            if timing.get('subjects'):
                n = timing['subjects']
                leg['breakdown_s_per_subject'] = {k: timing[k] / n for k in ('wait_for_files_s', 'stage_s', 'upload_and_kernels_s', 'csv_rows_s')}
                leg['read_threads_s_per_subject'] = timing['read_thread_s'] / n
                leg['batches'] = timing['batches']
            legs[tag] = leg
        record['legs'] = legs
        import glob

        def csvs(tag):
            base = os.path.join(root, 'eval_' + tag)
            return {os.path.relpath(f, base): open(f, 'rb').read() for f in glob.glob(os.path.join(base, '**', '*.csv'), recursive=True)}

        record['csv_bytes_equal'] = csvs('fused') == csvs('plain') == csvs('fused_batch1')
        k = min(args.oracle_subjects, args.subjects)
        if k > 0:
            reads, evals = zip(*[oracle_subject(root, n_, args.act) for n_ in names[:k]])
            record['oracle_numpy'] = dict(subjects=k, read_s_per_subject=sum(reads) / k, evaluate_s_per_subject=sum(evals) / k,
                                          s_per_subject=(sum(reads) + sum(evals)) / k, threads=1,
                                          note='oracle/calib_oracle.py (numpy restatement of common/evalutation/numpyfunctions.py) on the same files')
            record['speedup_vs_oracle'] = record['oracle_numpy']['s_per_subject'] / legs['fused']['s_per_subject']
        line = json.dumps(record)
        print(line)
        if args.out:
            with open(args.out, 'w') as f:
                f.write(json.dumps(record, indent=1) + '\n')
    finally:
        if not args.keep:
            shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
