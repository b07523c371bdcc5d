#!/usr/bin/env python3
"""Numerics study on the CPU (no GPU needed): how far is a float32 Winograd F(4x4,3x3) evaluation of the deep conv units
from the float64 network, next to the direct float32 form (what the reference computes) and float32 F(2x2,3x3) (what
csrc/rcu_wino.hip computes)?  Decides whether an F(4x4,3x3) kernel can hold the parity gate (logits <= 2e-5,
probabilities / entropy <= 1e-4 against the oracle).

    python tools/wino43_numerics.py [min_cin for F(4x4,3x3), default 128] [slices, default 2]

Each Winograd stage is evaluated in float32 with the operation order a kernel would use (transform in registers, channel
contraction as a float32 sum, output transform, then bias/BN); the contraction's summation order differs from an MFMA chain but
has the same error size.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import unet_oracle as uo  # noqa: E402

PARAMS = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05)

# F(2x2,3x3)
B2T = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
A2T = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
# F(4x4,3x3), interpolation points 0, +-1, +-2, inf (Lavin & Gray)
B4T = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
               [0, 0, 1]], dtype=np.float64)
A4T = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def wino_conv(x, w, bias, m, BT, G, AT):
    """float32 Winograd F(m x m, 3x3) conv, padding 1.  x [N,C,H,W] float32, H, W multiples of m."""
    n, c, h, wd = x.shape
    a = m + 2
    U = torch.as_tensor(np.einsum('ir,ocrs,js->ijoc', G, w.double().numpy(), G)).float()      # rounded once, as on the host
    xp = F.pad(x, (1, 1, 1, 1))
    # tiles [N, C, th, tw, a, a]
    t = xp.unfold(2, a, m).unfold(3, a, m)
    bt = torch.as_tensor(BT).float()
    v = torch.einsum('ir,nctwrs->nctwis', bt, t)       # float32 row transform
    v = torch.einsum('js,nctwis->nctwij', bt, v)       # float32 column transform
    mm = torch.einsum('ijoc,nctwij->notwij', U, v)     # float32 contraction over channels per position
    at = torch.as_tensor(AT).float()
    y = torch.einsum('pi,notwij->notwpj', at, mm)
    y = torch.einsum('qj,notwpj->notwpq', at, y)       # [N, O, th, tw, m, m]
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], h, wd)
    return y + bias[None, :, None, None]


def forward(state, x, mode, min_cin, dtype=torch.float32):
    """mode: 'direct' | 'f23' | 'f43' (F(4,3) for units with cin >= min_cin, F(2,3) elsewhere where cin >= 32)."""
    plan, _ = uo.unet_plan(**PARAMS)
    st = {k: torch.as_tensor(v).to(dtype) if torch.as_tensor(v).is_floating_point() else torch.as_tensor(v) for k, v in state.items()}
    x = x.to(dtype)
    skips = []
    for op in plan:
        kind = op['kind']
        if kind == 'unit':
            k = op['key']
            w, b = st[k + '.conv.weight'], st[k + '.conv.bias']
            cin = w.shape[1]
            if mode == 'direct' or cin < 32 or dtype == torch.float64:
                x = F.conv2d(x, w, b, padding=1)
            elif mode == 'f43' and cin >= min_cin and x.shape[-1] % 4 == 0 and x.shape[-2] % 4 == 0:
                x = wino_conv(x, w, b, 4, B4T, G4, A4T)
            else:
                x = wino_conv(x, w, b, 2, B2T, G2, A2T)
            x = F.batch_norm(x, st[k + '.bn.running_mean'], st[k + '.bn.running_var'], st[k + '.bn.weight'], st[k + '.bn.bias'],
                             False, 0.0, uo.BN_EPS)
            x = F.relu(x)
        elif kind == 'pool':
            skips.append(x)
            x = F.max_pool2d(x, 2)
        elif kind == 'up':
            skip = skips.pop()
            up = F.interpolate(x, scale_factor=2, mode='nearest')
            up = F.conv2d(up, st[op['key'] + '.weight'], st[op['key'] + '.bias'], padding=1)
            x = torch.cat((up, skip), 1)
        elif kind == 'head':
            return F.conv2d(x, st[op['key'] + '.weight'], st[op['key'] + '.bias'])


def main():
    min_cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    import bench
    model = bench.make_model(20, 'cpu')
    state = {k: v.detach() for k, v in model.state_dict().items()}
    x = bench.make_volume(20)[0][78:78 + n]
    ref = forward(state, x, 'direct', 0, torch.float64)
    pref = torch.softmax(ref, 1)
    href = -(pref * torch.log(pref.clamp_min(1e-300))).sum(1)
    print('logits: |max| {:.3f}, std {:.3f}'.format(float(ref.abs().max()), float(ref.std())))
    for mode in ('direct', 'f23', 'f43'):
        y = forward(state, x, mode, min_cin).double()
        p = torch.softmax(y.float(), 1).double()
        hh = -(p * torch.log(p.clamp_min(1e-300))).sum(1)
        print('{:<7} max|dlogit| {:.3e}  rms {:.3e}   max|dp| {:.3e}   max|dH| {:.3e}'.format(
            mode, float((y - ref).abs().max()), float((y - ref).pow(2).mean().sqrt()), float((p - pref).abs().max()),
            float((hh - href).abs().max())))


if __name__ == '__main__':
    main()
