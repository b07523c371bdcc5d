#!/usr/bin/env python3
"""Per-tile s_memtime trace of the F(4x4,3x3) kernel (ablation build only: `make BUILD=_build_ablate OUT=../librcu_hip_ablate.so EXTRA=-DRCU_WINO4_ABLATIONS` in csrc/, then
RCU_HIP_LIBRARY=reliability-challenges-uncertainty_amd/librcu_hip_ablate.so).

    python tools/wino4_trace.py [layer name substring ...]

For every named layer: where a workgroup's time goes between tiles -- the chunk loop, the epilogue, the wait for the next tile's first
chunk (and this tile's stores), the barrier, the first fragment reads.  Ticks of s_memtime (100 MHz on gfx950), medians over all
workgroups and waves of the tiles 0..2 of each workgroup.  (Since the LDS-DMA of a tile's last chunk waits for the next tile's first chunk at its
own barrier, nothing is waited for behind the epilogue: the `vmcnt(0)` and `barrier` rows read ~0 on the current kernel.)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('RCU_W4_VARIANT', '128')   # + 256 stores out of range, + 512 no store instructions, + 1024 no output transform
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from rcu_amd import _lib, steps  # noqa: E402


def main():
    names = sys.argv[1:] or ['down_convs.1.block.block.1', 'down_convs.2.block.block.1', 'up_convs.1.block.block.0', 'down_convs.3.block.block.1']
    dev = torch.device('cuda')
    n, h, w = 160, 192, 128
    m = bench.make_model(20, 'cpu').to(dev)
    steps.set_dropout_mode(m, True)
    xb = bench.make_volume(20)[0][:n].to(dev)
    for _ in range(2):
        m(xb)
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.rcu_debug_w4_trace.restype = ctypes.c_int
    lib.rcu_debug_w4_trace.argtypes = [ctypes.c_void_p]
    buf = np.zeros((256, 4, 4, 8), dtype=np.uint64)
    for L in m.layer_table(h, w, n):
        if 'winograd4' not in L['kernel'] or not any(s in L['name'] for s in names):
            continue
        m.run_layer(h, w, n, L['index'])
        torch.cuda.synchronize()
        assert lib.rcu_debug_w4_trace(buf.ctypes.data) == 0
        t = buf[:, :, :3, :6].astype(np.int64).reshape(-1, 6)
        t = t[t[:, 0] > 0]
        d = np.diff(t, axis=1)
        med = np.median(d, axis=0)
        p90 = np.percentile(d, 90, axis=0)
        print('{:<44} {:>4}->{:<4} {:>3}x{:<3} {}  chunks {}'.format(L['name'][:44], L['cin'], L['cout'], L['height'], L['width'], L['kernel'],
                                                                   L['cin'] // 8))
        for k, lab in enumerate(('chunk loop', 'epilogue', 'vmcnt(0)', 'barrier', 'first fragment reads')):
            print('    {:<22} median {:>8.0f}   p90 {:>8.0f} ticks'.format(lab, med[k], p90[k]))
        # inside the chunk loop: the tile's chunk 0, chunk 1 and the rest (slots 6 / 7 = end of chunk 0 / 1, relative to the tile start)
        t8 = buf[:, :, :3, :8].astype(np.int64).reshape(-1, 8)
        t8 = t8[t8[:, 0] > 0]
        c0, c1, rest = t8[:, 6] - t8[:, 0], t8[:, 7] - t8[:, 6], t8[:, 1] - t8[:, 7]
        nch = L['cin'] // 8
        print('    chunk 0 median {:>6.0f}  chunk 1 {:>6.0f}  chunks 2..{} {:>8.0f} = {:>6.0f} each'.format(
            np.median(c0), np.median(c1), nch - 1, np.median(rest), np.median(rest) / max(nch - 2, 1)))


if __name__ == '__main__':
    main()
