#!/usr/bin/env python3
"""GPU idle and overlap time from a rocprofv3 kernel trace: how much of the wall time no kernel, one kernel, two or more kernels are
running.    python tools/lane_overlap.py <..._kernel_trace.csv> [skip fraction at the start, default 0.5]

Used on `rocprofv3 --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --lanes {1,2}` to show what the stream
lanes of rcu_amd.steps.StreamLanes fill: the gaps between the dependent kernels of one forward pass."""
import csv
import sys


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path))]
    rows = [r for r in rows if 'conv' in r[2] or 'head_kernel' in r[2] or 'mc_finalize' in r[2]]
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + int((t1 - t0) * skip)          # leave the warm-up / handle creation out: look at the steady part
    rows = [r for r in rows if r[0] >= lo]
    # the bench's legs (timed region, H2D leg, serial leg, head probe) are separated by host work: split at gaps > 2 ms
    segs, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - max(x[1] for x in cur[-8:]) > 2_000_000:
            segs.append(cur)
            cur = []
        cur.append(r)
    segs.append(cur)
    seg = max(segs, key=len)                  # the longest run of back-to-back kernels = the timed region
    ev = sorted([(s, 1) for s, _, _ in seg] + [(e, -1) for _, e, _ in seg])
    depth, last, acc = 0, ev[0][0], {}
    for t, d in ev:
        acc[min(depth, 2)] = acc.get(min(depth, 2), 0) + (t - last)
        depth += d
        last = t
    total = sum(acc.values())
    ker = sum(e - s for s, e, _ in seg)
    print('{}: {} kernels over {:.1f} ms of wall time; sum of kernel durations {:.1f} ms'.format(path.split('/')[-1], len(seg), total / 1e6, ker / 1e6))
    for k, name in ((0, 'no kernel running'), (1, 'one kernel running'), (2, 'two or more running')):
        print('    {:<22} {:>7.2f} ms  {:>5.1f} %'.format(name, acc.get(k, 0) / 1e6, 100.0 * acc.get(k, 0) / total))


if __name__ == '__main__':
    main()
