#!/usr/bin/env python3
"""The standalone aggregation kernels of the step seam against the HBM roofline: head_kernel (1x1 classifier + softmax + statistics update on the
32-channel head activations; the path of the sigma head, of more than two classes and of foreign steps), mc_accumulate_kernel (softmax of a
logits volume into the statistics) and mc_finalize_kernel (mean / entropy / mutual information / variance out of them) -- with the default
float32 S = 2 statistics and with every output tracked (float64, S = 5) -- on the 160-slice BraTS volume and on four of them (the small
kernels take 15-20 us on one volume: their launch ramp and tail are a sixth of that, which the larger launch shows).

    python tools/agg_bench.py [reps, default 20]          -> one JSON line

ALGORITHMIC bytes follow SURVEY.md 8d (float32 planes); `moved` is what the kernels really move (float64 planes with all outputs; the
32-channel activations instead of logits for head_kernel)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from rcu_amd import steps  # noqa: E402


def timed(fn, reps):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device('cuda')
    h, w = bench.HEIGHT, bench.WIDTH
    out = {}
    for n in (160, 640):
        vox = n * h * w
        logits = torch.randn((n, 2, h, w), device=dev)
        for tag, flags, S, word in (('default', (False, False), 2, 4), ('all_outputs', (True, True), 5, 8)):
            st = steps.McStatistics(n, 2, h, w, dev, *flags)
            st.count = 1
            ms = timed(lambda: st.accumulate(logits), reps)
            alg, moved = vox * (2 * 4 + 2 * S * 4), vox * (2 * 4 + 2 * S * word)
            out['mc_accumulate_kernel/{}/{}'.format(tag, n)] = dict(us=ms * 1e3, algorithmic_gbs=alg / ms / 1e6, moved_gbs=moved / ms / 1e6,
                                                                   moved_frac=moved / ms / 1e6 / bench.PEAK_HBM_GBS)
            st.count = 20
            outs = (5 if flags[0] else 3)
            ms = timed(lambda: st.finalize(*flags), reps)
            alg, moved = vox * (S * 4 + outs * 4), vox * (S * word + outs * 4)
            out['mc_finalize_kernel/{}/{}'.format(tag, n)] = dict(us=ms * 1e3, algorithmic_gbs=alg / ms / 1e6, moved_gbs=moved / ms / 1e6,
                                                                 moved_frac=moved / ms / 1e6 / bench.PEAK_HBM_GBS)
            del st
        del logits
    # head_kernel on the plan of the headline run (160 slices), fused head off
    model = bench.make_model(20, dev)
    x = bench.make_volume(20)[0].to(dev)
    n = x.shape[0]
    vox = n * h * w
    model.set_fuse_head(False)
    for tag, flags, S, word in (('default', (False, False), 2, 4), ('all_outputs', (True, True), 5, 8)):
        st = steps.McStatistics(n, 2, h, w, dev, *flags)
        model.forward_accumulate(x, st)
        model.profile_begin(h, w, n, reps)
        for _ in range(reps):
            model.forward_accumulate(x, st)
        torch.cuda.synchronize()
        cnt, ms = model.profile_collect(h, w, n)
        t = ms[-1] / max(cnt, 1)
        moved = 4.0 * vox * 32 + 2 * S * word * vox
        out['head_kernel/{}/{}'.format(tag, n)] = dict(us=t * 1e3, moved_gbs=moved / t / 1e6, moved_frac=moved / t / 1e6 / bench.PEAK_HBM_GBS,
                                                       algorithmic_gbs=vox * (2 * 4 + 2 * S * 4) / t / 1e6)
        del st
    print(json.dumps(out))


if __name__ == '__main__':
    main()
