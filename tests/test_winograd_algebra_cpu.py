"""The algebra the Winograd kernels rely on (csrc/rcu_wino.hip, csrc/rcu_wino_up.hip, weight packing in csrc/rcu_api.hip),
checked in float64 on the CPU against torch's conv2d / nearest interpolation -- the operations of the reference
(common/model/unet.py:8-23, 98-120; common/model/helpers.py:5-16).  No product code runs here: the GPU tests compare the
kernels themselves with the oracle; this file pins the transforms, the sub-pixel fold and the sign fold they implement."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
DT = torch.float64

# F(2x2, 3x3): Y = A^T [ (G g G^T) .* (B^T d B) ] A
BT23 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=DT)
G23 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=DT)
AT23 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=DT)
# F(2x2, 2x2)
BT22 = torch.tensor([[1, -1, 0], [0, 1, 0], [0, -1, 1]], dtype=DT)
G22 = torch.tensor([[1, 0], [1, 1], [0, 1]], dtype=DT)
AT22 = torch.tensor([[1, 1, 0], [0, 1, 1]], dtype=DT)


def test_f23_equals_conv3x3():
    n, cin, cout, h, w = 2, 5, 3, 8, 12
    x = torch.randn(n, cin, h, w, dtype=DT)
    g = torch.randn(cout, cin, 3, 3, dtype=DT)
    ref = F.conv2d(x, g, padding=1)
    xp = F.pad(x, (1, 1, 1, 1))
    patches = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # [n, cin, h/2, w/2, 4, 4], stride 2
    V = torch.einsum('ij,nchwjk,lk->nchwil', BT23, patches, BT23)      # the in-register input transform of the kernel
    U = torch.einsum('ij,ocjk,lk->ocil', G23, g, G23)                  # the host-side weight transform
    M = torch.einsum('nchwil,ocil->nohwil', V, U)                      # 16 GEMMs over the channels (the MFMA part)
    Y = torch.einsum('ij,nohwjk,lk->nohwil', AT23, M, AT23)            # the lane-local output transform
    out = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, cout, h, w)
    assert (out - ref).abs().max() < 1e-12


def _fold(parity, t, d):
    """Kernel row d of the 3x3 window lands on low-resolution row t (0 / 1) of the 2x2 window of output parity `parity`
    (fold_set in csrc/rcu_api.hip)."""
    return (d == 0 if t == 0 else d >= 1) if parity == 0 else (d <= 1 if t == 0 else d == 2)


def _class_taps(g, a, b):
    """2x2 tap weights of parity class (a, b): sums of the 3x3 taps that read the same low-resolution pixel."""
    wc = torch.zeros(g.shape[0], g.shape[1], 2, 2, dtype=DT)
    for ty in range(2):
        for tx in range(2):
            for dy in range(3):
                for dx in range(3):
                    if _fold(a, ty, dy) and _fold(b, tx, dx):
                        wc[:, :, ty, tx] += g[:, :, dy, dx]
    return wc


def test_subpixel_fold_equals_upsample_then_conv3x3():
    n, cin, cout, h, w = 1, 4, 3, 6, 8
    x = torch.randn(n, cin, h, w, dtype=DT)
    g = torch.randn(cout, cin, 3, 3, dtype=DT)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), g, padding=1)
    xp = F.pad(x, (1, 1, 1, 1))
    out = torch.zeros_like(ref)
    for a in range(2):
        for b in range(2):
            wc = _class_taps(g, a, b)
            # output (2y+a, 2x+b) reads low-res rows y+a-1, y+a and columns x+b-1, x+b
            win = xp[:, :, a:a + h + 1, b:b + w + 1]
            out[:, :, a::2, b::2] = F.conv2d(win, wc)
    assert (out - ref).abs().max() < 1e-12


def test_f22_per_class_with_shared_patch_and_sign_fold():
    """The up-conv kernel: one 4x4 low-resolution patch P per 2x2 low-resolution tile; class (a, b) uses rows a..a+2 and
    columns b..b+2.  Rows: rho = (e0 - e1, e1, e2 - e1).  Columns: gamma = (P0 - P1, P1, P2 - P1, P2, P3 - P2); b = 0
    multiplies (gamma0, gamma1, gamma2), b = 1 multiplies (gamma2, gamma3, gamma4) with column 0 of its weights negated."""
    n, cin, cout, h, w = 1, 3, 2, 4, 6
    x = torch.randn(n, cin, h, w, dtype=DT)
    g = torch.randn(cout, cin, 3, 3, dtype=DT)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), g, padding=1)
    xp = F.pad(x, (1, 1, 1, 1))
    out = torch.zeros_like(ref)
    for ty in range(h // 2):
        for tx in range(w // 2):
            P = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]                       # [n, cin, 4, 4]
            gam = torch.stack([P[..., 0] - P[..., 1], P[..., 1], P[..., 2] - P[..., 1], P[..., 2], P[..., 3] - P[..., 2]], -1)
            for a in range(2):
                e0, e1, e2 = gam[:, :, a], gam[:, :, a + 1], gam[:, :, a + 2]        # rows of the column-transformed patch
                rho = torch.stack([e0 - e1, e1, e2 - e1], 2)                        # [n, cin, 3, 5]
                for b in range(2):
                    U = torch.einsum('it,octu,ju->ocij', G22, _class_taps(g, a, b), G22)   # [cout, cin, 3, 3]
                    if b == 1:
                        U[:, :, :, 0] = -U[:, :, :, 0]                               # the sign folded into the packed weights
                    Vab = rho[:, :, :, 2 * b:2 * b + 3]                              # [n, cin, 3, 3]
                    M = torch.einsum('ncij,ocij->noij', Vab, U)
                    Y = torch.einsum('ui,noij,vj->nouv', AT22, M, AT22)              # 2x2 low-res outputs of this class
                    for u in range(2):
                        for v in range(2):
                            out[:, :, 2 * (2 * ty + u) + a, 2 * (2 * tx + v) + b] = Y[:, :, u, v]
    assert (out - ref).abs().max() < 1e-12


def test_f23_float32_is_as_accurate_as_direct_float32():
    """Numerics claim of DESIGN.md 3.1a on one wide layer: against float64, F(2x2,3x3) evaluated in float32 is no worse than
    the direct float32 convolution by more than a small factor."""
    n, cin, cout, h, w = 1, 256, 16, 8, 8
    x = torch.randn(n, cin, h, w, dtype=DT)
    g = torch.randn(cout, cin, 3, 3, dtype=DT) / (3 * cin ** 0.5)
    ref = F.conv2d(x, g, padding=1)
    direct = F.conv2d(x.float(), g.float(), padding=1).double()
    xp = F.pad(x.float(), (1, 1, 1, 1))
    patches = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    V = torch.einsum('ij,nchwjk,lk->nchwil', BT23.float(), patches, BT23.float())
    U = torch.einsum('ij,ocjk,lk->ocil', G23, g, G23).float()          # host: double, rounded once
    M = torch.einsum('nchwil,ocil->nohwil', V, U)
    Y = torch.einsum('ij,nohwjk,lk->nohwil', AT23.float(), M, AT23.float())
    wino = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, cout, h, w).double()
    e_direct = (direct - ref).abs().max().item()
    e_wino = (wino - ref).abs().max().item()
    assert e_wino < 4 * e_direct + 1e-7, (e_wino, e_direct)
