#!/usr/bin/env python3
"""Evaluation driver with the reference's command line (bin-eval/eval_uncertainty.py:248-251): --ds --ids --act, nothing else
required.  The directories come from rcu_amd.directories, the mirror of the reference's rechun/directories.py whose
"required to be set" entries can be given in the environment (RCU_BRATS_ORIG_DATA_DIR, RCU_BRATS_BASELINE_MC_PREDICT, ...)
instead of by editing the module; --pred_dir / --gt_dir / --out_dir override them per call."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--ds', type=str, nargs='?', help='the dataset to evaluate the runs on')
    parser.add_argument('--ids', type=str, nargs='*', help='the ids of the runs to be evaluated')
    parser.add_argument('--act', type=str, nargs='*', help='the names of the evaluation configuration')
    parser.add_argument('--pred_dir', type=str, default=None, help='root with one sub-directory per dataset and run id '
                        '(default: directories.PREDICT_DIR and the per-run *_PREDICT names)')
    parser.add_argument('--gt_dir', type=str, default=None, help='BraTS training tree / ISIC dataset prefix '
                        '(default: directories.BRATS_ORIG_DATA_DIR / ISIC_PREPROCESSED_TEST_DATA_DIR)')
    parser.add_argument('--out_dir', type=str, default=None, help='default: directories.EVAL_DIR')
    parser.add_argument('--batch_subjects', type=int, default=8, help='rcu_amd: subjects of a probability-map run evaluated per GPU launch')
    parser.add_argument('--plain', action='store_true', help='rcu_amd: the reference\'s subject-by-subject, action-by-action loop for every run')
    args = parser.parse_args()
    from rcu_amd import directories as dirs
    from rcu_amd import scripts
    ds = args.ds or 'brats'
    ids = args.ids or list(dirs.RUN_IDS)
    acts = args.act or ['minmax', 'ece_dice', 'calib', 'bnf_ue']
    print('\n**************************************')
    print('dataset: {}'.format(ds))
    print('to_evaluate: {}'.format(ids))
    print('eval_actions: {}'.format(acts))
    print('**************************************\n')
    if ds not in ('brats', 'isic'):
        raise ValueError('chose "brats" or "isic" as dataset')          # eval_uncertainty.py:27-28
    gt_dir = args.gt_dir or dirs.ground_truth_dir(ds)
    if not gt_dir:
        raise SystemExit('the ground-truth directory is not set: export RCU_BRATS_ORIG_DATA_DIR (the reference asks for the same '
                         'entry in rechun/directories.py:7) or pass --gt_dir')
    runs = {i: (os.path.join(args.pred_dir, ds, i) if args.pred_dir else dirs.prediction_dir(ds, i)) for i in ids}
    out_dir = os.path.join(args.out_dir, ds) if args.out_dir else dirs.eval_dir(ds)
    scripts.eval_uncertainty(ds, runs, gt_dir, out_dir, acts, fused=not args.plain, batch_subjects=args.batch_subjects)
