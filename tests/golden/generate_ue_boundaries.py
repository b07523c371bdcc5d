#!/usr/bin/env python3
"""G20: for which float32 foreground probabilities p does the reference call a voxel "uncertain"?

The 'bnf_ue' action of bin-eval/eval_uncertainty.py (:176-202, thresholds :239) thresholds the normalised entropy the
REFERENCE computes from the float32 probability map:   u(p) = ToEntropy(AddBackgroundProbabilities(p))
(rechun/eval/analysis.py:147-152, 189-203; common/evalutation/numpyfunctions.py:166-168) -- float32 products p*log(p) with numpy's
float32 log, a float64 sum of the two, / log 2.  u is a function of the float32 p ALONE, so "u(p) > tau" is a SET of float32 values per
threshold, and a GPU kernel that knows the set needs neither the entropy map nor a log that matches numpy's in the last ulp.

This script finds the sets by running the reference's own classes over EVERY float32 in [0, 1] (1,065,353,217 values, in chunks over a
process pool) and records, per threshold and per half of the unit interval, a window of bit patterns
    below `lo_first`            : not uncertain          above `hi_last`           : not uncertain
    `lo_solid` .. `hi_solid`    : uncertain              in between (the windows)  : a bit mask, one bit per float32 value
The windows exist because u is not monotone at the ulp level (float32 rounding of q = 1 - p and of the products): the scan is the
proof that outside the windows the predicate is constant, and the monotonicity report (window widths, number of out-of-order values) is
stored with the fixture.  Output: tests/golden/g20_ue_boundaries.npz (the fixture) and
reliability-challenges-uncertainty_amd/csrc/rcu_ue_table.inc (the same table as a C initialiser, compiled into librcu_hip).

    python tests/golden/generate_ue_boundaries.py            (about 10 minutes on 8 cores)
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)

THRESHOLDS = [0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95]      # bin-eval/eval_uncertainty.py:239
ONE_BITS = 0x3F800000           # float32 1.0
HALF_BITS = 0x3F000000          # float32 0.5
CHUNK = 1 << 22
MAX_WINDOW = 256                # bits of mask the table format holds per window (4 x 64)

_ref = None


def reference_uncertainty(p32):
    """u(p) exactly as the evaluation script derives it for the 'probabilities' confidence entry (analysis.py:249-252)."""
    global _ref
    if _ref is None:
        import generate_golden as gg
        gg.install_reference()
        import rechun.eval.analysis as ref_an
        _ref = ref_an
    to_eval = {'probabilities': p32.copy()}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        to_eval = _ref.AddBackgroundProbabilities()(to_eval)
        to_eval = _ref.ToEntropy()(to_eval)
    u = to_eval['uncertainty']
    assert u.dtype == np.float64 and u.shape == p32.shape
    return u


def scan_chunk(start):
    """Bit patterns start .. start + CHUNK - 1 (clipped to 1.0): per threshold the extreme indices of the predicate in each half."""
    stop = min(start + CHUNK, ONE_BITS + 1)
    bits = np.arange(start, stop, dtype=np.uint32)
    u = reference_uncertainty(bits.view(np.float32))
    lower = bits <= HALF_BITS
    out = []
    for tau in THRESHOLDS:
        s = u > tau
        lo_t, lo_f = bits[lower & s], bits[lower & ~s]
        hi_t, hi_f = bits[~lower & s], bits[~lower & ~s]
        out.append((int(lo_t.min()) if lo_t.size else -1, int(lo_f.max()) if lo_f.size else -1,
                    int(hi_f.min()) if hi_f.size else -1, int(hi_t.max()) if hi_t.size else -1, int(s.sum())))
    return out


def main():
    import multiprocessing as mp
    starts = list(range(0, ONE_BITS + 1, CHUNK))
    with mp.Pool(int(os.environ.get('RCU_GOLDEN_WORKERS', '8'))) as pool:
        results = pool.map(scan_chunk, starts, chunksize=1)
    n_thr = len(THRESHOLDS)
    lo_first = np.full(n_thr, -1, np.int64)     # smallest p (bits) with u > tau
    lo_lastf = np.full(n_thr, -1, np.int64)     # largest p <= 0.5 with u <= tau
    hi_firstf = np.full(n_thr, -1, np.int64)    # smallest p > 0.5 with u <= tau
    hi_last = np.full(n_thr, -1, np.int64)      # largest p with u > tau
    total = np.zeros(n_thr, np.int64)
    for res in results:
        for k, (a, b, c, d, cnt) in enumerate(res):
            if a >= 0:
                lo_first[k] = a if lo_first[k] < 0 else min(lo_first[k], a)
            lo_lastf[k] = max(lo_lastf[k], b)
            if c >= 0:
                hi_firstf[k] = c if hi_firstf[k] < 0 else min(hi_firstf[k], c)
            hi_last[k] = max(hi_last[k], d)
            total[k] += cnt
    assert (lo_first > 0).all() and (hi_last > HALF_BITS).all() and (hi_firstf > HALF_BITS).all()
    # windows: [lo_first, lo_lastf] (empty when lo_lastf < lo_first: a clean step) and [hi_firstf, hi_last]
    lo_width = np.maximum(lo_lastf - lo_first + 1, 0)
    hi_width = np.maximum(hi_last - hi_firstf + 1, 0)
    assert lo_width.max() <= MAX_WINDOW and hi_width.max() <= MAX_WINDOW, (lo_width, hi_width)
    words = MAX_WINDOW // 64
    lo_mask = np.zeros((n_thr, words), np.uint64)
    hi_mask = np.zeros((n_thr, words), np.uint64)
    out_of_order = np.zeros((n_thr, 2), np.int64)
    for k, tau in enumerate(THRESHOLDS):
        for first, width, mask, col in ((lo_first[k], lo_width[k], lo_mask, 0), (hi_firstf[k], hi_width[k], hi_mask, 1)):
            if width == 0:
                continue
            bits = np.arange(first, first + width, dtype=np.uint32)
            s = reference_uncertainty(bits.view(np.float32)) > tau
            for i in np.nonzero(s)[0]:
                mask[k, i // 64] |= np.uint64(1) << np.uint64(i % 64)
            # values inside the window that break the order a monotone u would give (true left of false on the rising side, ...)
            out_of_order[k, col] = int((~s).sum()) if col == 0 else int(s.sum())
    # consistency of the table with the scan: count of members per threshold
    members = np.zeros(n_thr, np.int64)
    for k in range(n_thr):
        solid = (hi_firstf[k] - 1) - (lo_lastf[k] + 1) + 1 if lo_width[k] else (hi_firstf[k] - 1) - lo_first[k] + 1
        in_lo = sum(bin(int(w)).count('1') for w in lo_mask[k])
        in_hi = sum(bin(int(w)).count('1') for w in hi_mask[k])
        members[k] = solid + in_lo + in_hi
    assert np.array_equal(members, total), (members, total)
    # a probe vector for the GPU test: every window value, 40 neighbours on either side, 0, 0.5, 1 and their neighbours
    probe = [0, 1, HALF_BITS - 1, HALF_BITS, HALF_BITS + 1, ONE_BITS - 1, ONE_BITS]
    for k in range(n_thr):
        probe += list(range(int(lo_first[k]) - 40, int(max(lo_lastf[k], lo_first[k])) + 41))
        probe += list(range(int(hi_firstf[k]) - 40, int(max(hi_last[k], hi_firstf[k])) + 41))
    probe = np.unique(np.array(probe, dtype=np.uint32))
    probe_u = reference_uncertainty(probe.view(np.float32))
    probe_member = np.stack([probe_u > tau for tau in THRESHOLDS]).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, 'g20_ue_boundaries.npz'), thresholds=np.array(THRESHOLDS), lo_first=lo_first, lo_last_false=lo_lastf,
                        hi_first_false=hi_firstf, hi_last=hi_last, lo_mask=lo_mask, hi_mask=hi_mask, lo_width=lo_width, hi_width=hi_width,
                        out_of_order=out_of_order, members=total, values_scanned=np.array(ONE_BITS + 1),
                        probe_bits=probe, probe_uncertainty=probe_u, probe_member=probe_member,
                        numpy_version=np.array(np.__version__), cpu_features=np.array(_cpu_features()))
    write_table(lo_first, lo_width, hi_firstf, hi_width, lo_mask, hi_mask)
    print('scanned {} float32 values'.format(ONE_BITS + 1))
    for k, tau in enumerate(THRESHOLDS):
        print('tau {:4.2f}: uncertain from bits 0x{:08x} ({:.9g}) to 0x{:08x} ({:.9g}); ragged windows {} / {} values, out of order {} / {}'
              .format(tau, lo_first[k], np.uint32(lo_first[k]).view(np.float32), hi_last[k], np.uint32(hi_last[k]).view(np.float32),
                      lo_width[k], hi_width[k], out_of_order[k, 0], out_of_order[k, 1]))


def _cpu_features():
    try:
        from numpy._core._multiarray_umath import __cpu_features__ as feats
        return ' '.join(sorted(k for k, v in feats.items() if v))
    except Exception:  # noqa: BLE001
        return 'unknown'


def write_table(lo_first, lo_width, hi_firstf, hi_width, lo_mask, hi_mask):
    path = os.path.join(ROOT, 'reliability-challenges-uncertainty_amd', 'csrc', 'rcu_ue_table.inc')
    lines = ['// GENERATED by tests/golden/generate_ue_boundaries.py from the reference run over every float32 in [0, 1] -- do not edit.',
             '// {threshold, lo_first, lo_width, hi_first, hi_width, {lo_mask[4]}, {hi_mask[4]}}: p (as its bit pattern b) is "uncertain" for the',
             '// threshold iff  lo_first + lo_width <= b < hi_first,  or b sits in a window [first, first + width) and its mask bit is set.']
    for k, tau in enumerate(THRESHOLDS):
        lines.append('{{{!r}, 0x{:08x}u, {}u, 0x{:08x}u, {}u, {{{}}}, {{{}}}}},'.format(
            tau, int(lo_first[k]), int(lo_width[k]), int(hi_firstf[k]), int(hi_width[k]),
            ', '.join('0x{:016x}ull'.format(int(w)) for w in lo_mask[k]), ', '.join('0x{:016x}ull'.format(int(w)) for w in hi_mask[k])))
    with open(path, 'w') as f:
        f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
