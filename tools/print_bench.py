#!/usr/bin/env python3
"""One line per bench.py JSON file: value, step time, roofline record, parity.   python tools/print_bench.py out.json [...]"""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.load(open(path))
    except Exception as e:  # noqa: BLE001
        print(path, 'unreadable:', e)
        continue
    r = d['roofline']
    par = d.get('parity') or {}
    res = (d.get('resident') or {}).get('value')
    print('{}: {:.2f} {} (resident {}) | {:.2f} ms/step | lanes {} group {} | dominant {} frac {:.3f} avg launch {:.4f} ms | conv {:.3f} ms/forward frac {:.3f} | dp {} bins {}'.format(
        path, d['value'], d['unit'], None if res is None else round(res, 2), d['ms_per_step'], d['config'].get('lanes'), d['config'].get('pass_group'), r['kernel'], r['frac'], r['avg_launch_ms'],
        r['all_conv_kernels']['ms_per_forward'], r['all_conv_kernels']['frac'], par.get('max_abs_dprobabilities_vs_cpu'), par.get('bin_ids_equal')) +
          ' | ue counts equal {} (max delta {})'.format(par.get('ue_counts_equal'), par.get('ue_max_count_delta')))
