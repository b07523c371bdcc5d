#!/usr/bin/env python3
"""Evaluation driver -- flags of the reference's bin-eval/eval_uncertainty.py (--ds --ids --act) plus the
directories the reference hard-codes in rechun/directories.py (--pred_dir <root with one sub-directory per
run id>, --gt_dir <BraTS training tree, or the ISIC dataset prefix .../ISIC-2017_Test_v2>, --out_dir)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--ds', type=str, nargs='?', default='brats', help='the dataset to evaluate the runs on')
    parser.add_argument('--ids', type=str, nargs='*', help='the ids of the runs to be evaluated')
    parser.add_argument('--act', type=str, nargs='*', help='the names of the evaluation configuration')
    parser.add_argument('--pred_dir', type=str, default=os.path.join('out', 'predictions'))
    parser.add_argument('--gt_dir', type=str, required=True)
    parser.add_argument('--out_dir', type=str, default=os.path.join('out', 'eval'))
    args = parser.parse_args()
    ids = args.ids or ['baseline', 'baseline_mc', 'center', 'center_mc', 'ensemble', 'auxiliary_feat', 'auxiliary_segm',
                       'aleatoric']
    acts = args.act or ['minmax', 'ece_dice', 'calib', 'bnf_ue']
    print('\n**************************************')
    print('dataset: {}'.format(args.ds))
    print('to_evaluate: {}'.format(ids))
    print('eval_actions: {}'.format(acts))
    print('**************************************\n')
    from rcu_amd import scripts
    runs = {i: os.path.join(args.pred_dir, args.ds, i) for i in ids}
    scripts.eval_uncertainty(args.ds, runs, args.gt_dir, os.path.join(args.out_dir, args.ds), acts)
