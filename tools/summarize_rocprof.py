#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/, scratch) into the small files kept under profiles/.

  python tools/summarize_rocprof.py stats  <dir>/<prefix>_kernel_stats.csv  profiles/<name>_kernel_stats.csv
  python tools/summarize_rocprof.py pmc    <fetch_dir> <write_dir> [<sq_dir>]  profiles/<name>_pmc.json [--plan <bench json>]

`pmc` also refreshes profiles/pmc_traffic.json (bytes per launch per conv instantiation, read by bench.py):
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE / WRITE_SIZE are in KiB and on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads (/opt/skills/guides/MI355X_MICROARCH.md, section HBM);
the factor is exact for streaming reads (checked here on pack_input_kernel: 62.9 MB read -> 30.7 MKiB reported)
and an upper bound for partially coalesced ones.
"""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    """Demangled rocprof kernel name -> the kernel_name string of rcu_unet_layer_info / bench.py."""
    m = re.search(r'conv_igemm(?:_stream)?<rcu::ConvTile<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, \d+)?>', name)
    if m:
        ts, th, tw, bn, kc, _wm, _wn, taps, db = m.groups()
        tile = ('S{}'.format(ts) if ts != '1' else '') + 'T{}x{}'.format(th, tw)
        base = 'upconv_subpixel_igemm' if taps == '4' else 'conv3x3_igemm'
        return '{}<{},N{},K{}{}>'.format(base, tile, bn, kc, ',db' if db == '1' else '')
    m = re.search(r'(conv|upconv)_wino_stream<rcu::WinoTile<(\d+), (\d+), (\d+), (\d+), \d+, \d+(?:, \d+)*>', name)
    if m:
        kind, ts, th, tw, bn = m.groups()
        tile = ('S{}'.format(ts) if ts != '1' else '') + 'T{}x{}'.format(th, tw)
        # template arguments behind the tile: conv <T, HEAD, PART>, upconv <T, PART> (PART: the padded-level store side, round 6)
        flags = re.search(r'WinoTile<[^>]*>((?:, \w+)*)>', name)
        flags = [f.strip() for f in flags.group(1).split(',') if f.strip()] if flags else []
        head = kind == 'conv' and bool(flags) and flags[0] in ('true', '1')
        return '{}<{},N{},K8>{}'.format('conv3x3_winograd' if kind == 'conv' else 'upconv_winograd', tile, bn, '+head' if head else '')
    m = re.search(r'conv_wino4_stream<rcu::Wino4Tile<(\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\w+))?(?:, (\w+))?>(?:, \d+)?(?:, (\w+))?', name)
    if m:   # block = SB slices x BR x BC tiles of 4x4 pixels, workgroup = WS x WR blocks; the tile's last flag: the folded 12x8 geometry;
            # the kernel's last flag (behind the ablation variant): the classifier head in the epilogue
        sb, br, bc, ws, wr = (int(v) for v in m.groups()[:5])
        ts, th, tw = sb * ws, 4 * br * wr, 4 * bc
        if m.group(7) in ('true', '1'):
            th, tw = 12, 8
        return 'conv3x3_winograd4<{}T{}x{},N32,K8>{}'.format('S{}'.format(ts) if ts != 1 else '', th, tw, '+head' if m.group(8) in ('true', '1') else '')
    if 'conv3x3_first_kernel' in name:   # anonymous namespace of rcu_first.hip; one tile shape
        return 'conv3x3_first<T8x32,K36>'
    m = re.search(r'rcu::(\w+)', name)
    if m:
        return m.group(1)
    return name[:100]


def stats(src, dst):
    rows = list(csv.DictReader(open(src)))
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for r in rows:
            w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'],
                        r['MaxNs']])


def per_kernel(path, wanted):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] not in wanted:
            continue
        k = short(r['Kernel_Name'])
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        calls[k].add(r['Dispatch_Id'])
        agg[k]['_ns'] += 0
    return {k: dict(launches=len(calls[k]), **{c: v / len(calls[k]) for c, v in d.items() if not c.startswith('_')})
            for k, d in agg.items()}


def pmc(args):
    # --plan <bench json of the same build>: its roofline.plan_fingerprint goes into pmc_traffic.json, and bench.py reports the traffic
    # only while its own plan carries that fingerprint
    meta = {}
    if '--plan' in args:
        i = args.index('--plan')
        with open(args[i + 1]) as f:
            line = json.loads(f.read().strip().splitlines()[-1])
        meta = dict(plan_fingerprint=line['roofline'].get('plan_fingerprint'), pass_group=line['config'].get('pass_group'),
                    bench_value=line.get('value'))
        args = args[:i] + args[i + 2:]
    traffic_name = 'pmc_traffic.json'      # --traffic-name: the file bench.py reads for another workload (pmc_traffic_<workload>.json)
    if '--traffic-name' in args:
        i = args.index('--traffic-name')
        traffic_name = args[i + 1]
        args = args[:i] + args[i + 2:]
    out_path = args[-1]
    dirs = args[:-1]
    find = lambda d: [os.path.join(d, f) for f in os.listdir(d) if f.endswith('counter_collection.csv')][0]  # noqa: E731
    fetch = per_kernel(find(dirs[0]), {'FETCH_SIZE'})
    write = per_kernel(find(dirs[1]), {'WRITE_SIZE'})
    result = {}
    for k in fetch:
        if not (k.startswith(('conv3x3', 'upconv')) or k.endswith('_kernel')):
            continue
        f, w = fetch[k]['FETCH_SIZE'], write.get(k, {}).get('WRITE_SIZE', 0.0)
        result[k] = dict(launches=fetch[k]['launches'], fetch_size_kib_per_launch=f, write_size_kib_per_launch=w,
                         hbm_bytes_per_launch=(2 * f + w) * 1024)
    if len(dirs) > 2:
        names = {'SQ_WAVES', 'SQ_BUSY_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE',
                 'SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE'}
        sq = per_kernel(find(dirs[2]), names)
        for k, d in sq.items():
            if k in result:
                result[k]['sq'] = {c: d[c] for c in d if c != 'launches'}
                if d.get('GRBM_GUI_ACTIVE'):
                    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs
                    result[k]['mfma_util'] = d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / (d['GRBM_GUI_ACTIVE'] / 8)
                if d.get('SQ_LDS_IDX_ACTIVE'):
                    result[k]['lds_bank_conflict_frac'] = d.get('SQ_LDS_BANK_CONFLICT', 0) / d['SQ_LDS_IDX_ACTIVE']
    with open(out_path, 'w') as f:
        json.dump(result, f, indent=1, sort_keys=True)
    traffic = {k: v['hbm_bytes_per_launch'] for k, v in result.items() if k.startswith(('conv3x3', 'upconv'))}
    traffic['_meta'] = meta
    with open(os.path.join(os.path.dirname(out_path), traffic_name), 'w') as f:
        json.dump(traffic, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == 'pmc':
        pmc(sys.argv[2:])
    else:
        raise SystemExit(__doc__)
