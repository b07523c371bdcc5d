"""Parity of the HIP path (through the C ABI) against the oracle and the golden vectors.
Run on the GPU box: ``python -m pytest tests -m gpu``.  fp32 tolerances are written at each check;
integer / index outputs are compared bit-for-bit."""
import numpy as np
import pytest
import torch

from conftest import golden_params, golden_state

pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-6      # |logit| ~ 0.1..1 after 24 stacked fp32 convs in a different summation order: measured <= 1e-7
PROB_TOL = 1e-4       # north_star: uncertainty maps within 1e-4 of the CPU reference


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a MI355X'
    return torch.device('cuda')


def _model(params, state, dev, **plan_options):
    """``plan_options``: rcu_unet_options fields (include/rcu.h) -- which kernel family / layout the planner may choose."""
    from rcu_amd.model import UNet
    m = UNet(**params)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in state.items()})
    m.plan_options = dict(plan_options)
    return m.to(dev)


def _maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


# ---------------------------------------------------------------------------------- U-Net forward
def test_unet_eval_golden(golden, dev):
    g = golden('g1_unet_eval')
    m = _model(golden_params(g), golden_state(g), dev)
    for tag in ('a', 'b'):
        y = m(torch.from_numpy(g['x_' + tag]).to(dev)).cpu().numpy()
        assert y.shape == g['logits_' + tag].shape
        assert _maxdiff(y, g['logits_' + tag]) < LOGIT_TOL
    g8 = golden('g1_unet_eval_sf8')
    m8 = _model(golden_params(g8), golden_state(g8), dev)
    assert _maxdiff(m8(torch.from_numpy(g8['x']).to(dev)).cpu().numpy(), g8['logits']) < LOGIT_TOL


def test_unet_isic_golden(golden, dev):
    g = golden('g5_unet_isic')
    m = _model(golden_params(g), golden_state(g), dev)
    assert _maxdiff(m(torch.from_numpy(g['x']).to(dev)).cpu().numpy(), g['logits']) < LOGIT_TOL


def test_unet_mc_injected_masks_golden(golden, dev):
    g = golden('g2_unet_mc')
    m = _model(golden_params(g), golden_state(g), dev)
    assert [s[0] for s in m.dropout_sites()] == list(g['sites'])
    x = torch.from_numpy(g['x']).to(dev)
    for t in range(int(g['T'])):
        masks = [g['mask_{}_{}'.format(t, s)] for s in range(len(g['sites']))]
        assert _maxdiff(m(x, masks).cpu().numpy(), g['logits_{}'.format(t)]) < LOGIT_TOL


def test_unet_dropout_center_golden(golden, dev):
    g = golden('g3_unet_center')
    m = _model(golden_params(g), golden_state(g), dev)
    assert [s[0] for s in m.dropout_sites()] == list(g['sites'])
    x = torch.from_numpy(g['x']).to(dev)
    masks = [g['mask_{}'.format(s)] for s in range(len(g['sites']))]
    assert _maxdiff(m(x, masks).cpu().numpy(), g['logits']) < LOGIT_TOL
    assert _maxdiff(m(x).cpu().numpy(), g['logits_eval']) < LOGIT_TOL


def test_unet_sigma_head_golden(golden, dev):
    from rcu_amd import steps
    g = golden('g4_unet_sigma')
    m = _model(golden_params(g), golden_state(g), dev)
    logits, sigma = m(torch.from_numpy(g['x']).to(dev))
    assert _maxdiff(logits.cpu().numpy(), g['logits']) < LOGIT_TOL
    assert _maxdiff(sigma.cpu().numpy(), g['sigma_raw']) < LOGIT_TOL
    bc = steps.BatchContext({'images': torch.from_numpy(g['x'])}, 0)
    steps.AleatoricPredictStep()(bc, None, steps.TorchTestContext('cuda', m))
    assert set(bc.output) == {'logits', 'sigma', 'probabilities'}
    assert _maxdiff(bc.output['sigma'].cpu().numpy(), g['sigma_abs']) < LOGIT_TOL
    assert _maxdiff(bc.output['probabilities'].cpu().numpy(), g['probabilities']) < PROB_TOL
    # exact-input checks of the writer-side selection: feed the golden logits / sigma
    pred, sp = steps.sigma_of_prediction(torch.from_numpy(g['logits']).to(dev), torch.from_numpy(g['sigma_raw']).to(dev))
    assert np.array_equal(pred.cpu().numpy(), g['prediction'])
    assert _maxdiff(sp.cpu().numpy(), g['sigma_pred']) < 1e-6


@pytest.mark.parametrize('shape', [(2, 192, 128), (3, 48, 32), (1, 32, 48), (1, 192, 128), (5, 96, 64), (8, 16, 16), (5, 128, 512)])
def test_unet_full_width_vs_oracle(dev, shape):
    """start_filters=32 (the shipped width; no channel padding anywhere) incl. the BraTS slice size whose
    bottom level (12x8) uses the two-slices-per-workgroup kernel; dropout masks sampled and injected.  128 x 512: an 8 x 32 bottom level, the
    full-width tile of four slices (conv3x3_winograd4<S4T8x32>, round 6) on a ragged batch of five."""
    from oracle import unet_oracle as uo
    n, h, w = shape
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05)
    st = uo.synthetic_state(20, **params)
    m = _model(params, st, dev)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 4, h, w, generator=g)
    _, sites = uo.unet_plan(**params)
    masks = uo.sample_masks(sites, n, 0.3, g)   # heavier dropout than the config to make masks matter
    for mk in (None, masks):
        ref = uo.unet_forward(st, x, mk, **params).numpy()
        out = m(x.to(dev), mk).cpu().numpy()
        assert _maxdiff(out, ref) < LOGIT_TOL
        ps = torch.softmax(torch.from_numpy(out), 1).numpy()
        pr = torch.softmax(torch.from_numpy(ref), 1).numpy()
        assert _maxdiff(ps, pr) < PROB_TOL


def test_winograd_kernels_vs_direct_and_oracle(dev):
    """Every Winograd instantiation (csrc/rcu_wino4.hip: F(4x4,3x3) conv units, csrc/rcu_wino.hip: F(2x2,3x3) conv units,
    csrc/rcu_wino_up.hip: F(2x2,2x2) sub-pixel up-convolutions) on the BraTS slice size with 8 slices -- enough for the work items
    that span 2 and 8 slices: against the oracle, against the direct kernels (plan option conv_winograd=0), and on ragged batches.  Three
    plans: the shipped selection (F(4x4,3x3) wherever it fits -- also the 32-channel full-resolution layers, the 2x2 max-pool in its
    epilogue and the two-source K loop at that tile, and conv_cls.0 with its fused head; head_winograd4=0 keeps that one on F(2x2,3x3)),
    the round-2 selection (conv_winograd4=3: only the layers with >= 64 output channels) and F(2x2,3x3) only (conv_winograd4=0)."""
    from oracle import unet_oracle as uo
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05)
    st = uo.synthetic_state(21, **params)
    g = torch.Generator().manual_seed(6)
    n, h, w = 8, 192, 128
    x = torch.randn(n, 4, h, w, generator=g)
    _, sites = uo.unet_plan(**params)
    masks = uo.sample_masks(sites, n, 0.3, g)
    ref = uo.unet_forward(st, x, masks, **params).numpy()
    head2 = {'conv3x3_winograd<T16x32,N32,K8>'}      # conv_cls.0 on F(2x2,3x3)
    common = {'upconv_winograd<T16x16,N64,K8>', 'upconv_winograd<T16x32,N32,K8>', 'upconv_winograd<S2T8x16,N64,K8>',
              'upconv_winograd<S8T4x8,N64,K8>'}
    w4 = {'conv3x3_winograd4<T32x32,N32,K8>', 'conv3x3_winograd4<S2T16x32,N32,K8>', 'conv3x3_winograd4<S8T8x16,N32,K8>'}
    # the 12x8 bottom level: F(4x4,3x3) with six tiles per slice in the eight slots of the S8 block (round 5) where its long work items fill the
    # chip's rounds -- 640-sample launches -- and always under conv_winograd4=2; F(2x2,3x3) for a plan of 8 slices under the default
    fold, strip = {'conv3x3_winograd4<S8T12x8,N32,K8>'}, {'conv3x3_winograd<S8T4x8,N64,K8>'}
    w2 = {'conv3x3_winograd<T16x16,N64,K8>', 'conv3x3_winograd<S2T8x16,N64,K8>'}
    outs = {}
    for mode, expected in (('1', common | w4 | strip), ('1h', common | w4 | strip | head2), ('2', common | w4 | fold), ('3', common | w4 | strip | head2),
                           ('0', common | w2 | strip | head2)):
        m_w = _model(params, st, dev, conv_winograd4=int(mode[0]), head_winograd4=0 if mode.endswith('h') else 1)
        rows = m_w.layer_table(h, w, n)
        kernels = {row['kernel'] for row in rows}
        assert expected <= kernels, (mode, expected - kernels)
        assert mode in ('1h', '3', '0') or not (head2 & kernels), mode
        n4 = sum('winograd4' in row['kernel'] for row in rows)
        # '3': the 96x64, 48x32 and 24x16 levels; '1h': + three 32-channel 192x128 units; '1': + conv_cls.0; '2': + the 12x8 level
        assert n4 == {'1': 16, '1h': 15, '2': 18, '3': 12, '0': 0}[mode], (mode, n4)
        out_w = m_w(x.to(dev), masks).cpu().numpy()
        assert _maxdiff(out_w, ref) < LOGIT_TOL, mode
        assert _maxdiff(torch.softmax(torch.from_numpy(out_w), 1).numpy(), torch.softmax(torch.from_numpy(ref), 1).numpy()) < PROB_TOL
        assert _maxdiff(m_w(x.to(dev)).cpu().numpy(), uo.unet_forward(st, x, None, **params).numpy()) < LOGIT_TOL, mode
        # ragged batches on the same plan (max_batch 8): slices are independent, so the results are the same bits
        for k in (3, 5):
            out_k = m_w(x[:k].to(dev), [mk[:k] for mk in masks]).cpu().numpy()
            assert np.array_equal(out_k, out_w[:k]), (mode, k)
        outs[mode] = out_w
    # the direct kernels on the same input
    m_d = _model(params, st, dev, conv_winograd=0)
    assert not any('winograd' in row['kernel'] for row in m_d.layer_table(h, w, n))
    out_d = m_d(x.to(dev), masks).cpu().numpy()
    assert _maxdiff(out_d, ref) < LOGIT_TOL
    for mode in outs:
        assert _maxdiff(outs[mode], out_d) < 2e-5, mode


def test_seeded_masks_are_a_function_of_the_seed_alone(dev):
    """UNet.seeded_masks (rcu_dropout_masks: the Dropout2d factors of a launch's passes in one kernel).  A pass's mask must not depend on the
    group it is drawn in (any split of the seeds gives the rows the single-pass calls give, in the group layout -- ``group_masks`` of the
    single draws), nor on the call (same seed, same bits); groups beyond the 32 seeds one launch carries; values in {0, 1 / keep} with the
    Bernoulli(1 - p) law per element and no visible correlation between passes; sites in eval mode get ones, p = 1 zeros
    (sites of mixed state), as ``sample_masks`` gives them."""
    from oracle import unet_oracle as uo
    from rcu_amd import steps
    params = dict(nb_classes=2, in_channels=4, depth=2, start_filters=8, dropout=0.3)
    m = _model(params, uo.synthetic_state(3, **params), dev)
    steps.set_dropout_mode(m, True)
    for n, seeds in ((1, [5, 6]), (3, [11, 12, 13, 14]), (32, list(range(100, 110))), (2, [7]), (2, list(range(1, 72)))):
        singles = [m.seeded_masks(n, dev, [seed]) for seed in seeds]
        want = m.group_masks(singles, n, dev) if len(seeds) > 1 else singles[0]
        got = m.seeded_masks(n, dev, seeds)
        assert got.shape == want.shape and torch.equal(got, want), (n, len(seeds))
        assert torch.equal(m.seeded_masks(n, dev, seeds), got)
        half = len(seeds) // 2
        if half:            # a prefix of the group drawn on its own: the same rows
            assert torch.equal(m.seeded_masks(n, dev, seeds[:half]), m.group_masks(singles[:half], n, dev) if half > 1 else singles[0])
    # ... and they are the oracle's (oracle/mask_oracle.py: Philox4x32-10 pinned by the published known-answer vectors), bit for bit
    from oracle import mask_oracle as mo
    sites = m.dropout_sites()
    for n, seeds, first in ((3, [11, 12, 13, 14], 0), (1, [2 ** 63 + 5], 7), (5, list(range(40, 75)), 2 ** 35 + 3)):
        want = mo.group_masks(seeds, n, [c for _, c in sites], [0.7] * len(sites), first_sample=first)
        assert np.array_equal(m.seeded_masks(n, dev, seeds, first).cpu().numpy(), want), (n, len(seeds))
    # a sample's factors are a function of its GLOBAL index, not of the batch it arrives in (round 6): eight samples at once = three, then five
    whole = m.seeded_masks(8, dev, [21, 22], 100)
    head, tail = m.seeded_masks(3, dev, [21, 22], 100), m.seeded_masks(5, dev, [21, 22], 103)
    at_w = at_h = at_t = 0
    for _, c in sites:
        for t in range(2):
            rows = whole[at_w + t * 8 * c:at_w + (t + 1) * 8 * c].view(8, c)
            assert torch.equal(rows[:3], head[at_h + t * 3 * c:at_h + (t + 1) * 3 * c].view(3, c))
            assert torch.equal(rows[3:], tail[at_t + t * 5 * c:at_t + (t + 1) * 5 * c].view(5, c))
        at_w, at_h, at_t = at_w + 2 * 8 * c, at_h + 2 * 3 * c, at_t + 2 * 5 * c
    big = m.seeded_masks(64, dev, list(range(1000, 1040)))                 # 40 passes x 64 images x 72 channels
    values = torch.unique(big)
    keep = 1.0 - 0.3
    assert values.numel() == 2 and float(values[0]) == 0.0 and abs(float(values[1]) - 1.0 / keep) < 1e-6
    frac = float((big > 0).double().mean())
    assert abs(frac - keep) < 4 * (keep * (1 - keep) / big.numel()) ** 0.5 + 1e-4, frac
    a, b = m.seeded_masks(64, dev, [1000]) > 0, m.seeded_masks(64, dev, [1001]) > 0
    agree = float((a == b).double().mean())                                 # independent draws agree with probability keep^2 + (1 - keep)^2
    assert abs(agree - (keep * keep + (1 - keep) ** 2)) < 0.03, agree
    assert not torch.equal(m.seeded_masks(4, dev, [2 ** 40 + 9]), m.seeded_masks(4, dev, [9]))          # the high word of the seed counts
    steps.set_dropout_mode(m, False)
    assert float(m.seeded_masks(2, dev, [3]).min()) == 1.0 and float(m.seeded_masks(2, dev, [3]).max()) == 1.0
    # sites of mixed state (two Dropout2d modules put back into eval mode by hand): the layout and the ones of the inactive sites
    steps.set_dropout_mode(m, True)
    m._site_modules[1].eval()
    m._site_modules[4].eval()
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    torch_way = m.sample_masks(3, dev, generator=g)                         # (other values: torch's generator; the same sites are constant)
    ours = m.seeded_masks(3, dev, [1])
    assert ours.shape == torch_way.shape and torch.equal(ours == 1.0, torch_way == 1.0)
    assert 0 < int((ours == 1.0).sum()) < ours.numel()
    steps.set_dropout_mode(m, False)


@pytest.mark.parametrize('in_channels,start_filters', [(3, 32), (4, 32), (6, 32), (4, 64), (1, 16)])
def test_first_layer_kernel_vs_tiled_kernel_and_oracle(dev, in_channels, start_filters):
    """csrc/rcu_first.hip (unpadded K = 9 taps x 4 or 8 channels, NCHW input read in place, 32 or 64 output channels) against
    the oracle, against the tiled first-layer kernel + channels-last copy (plan option conv_first=0), on a ragged batch and through a
    pass group (sample t * N + i reads image i)."""
    from oracle import unet_oracle as uo
    from rcu_amd import steps
    params = dict(nb_classes=2, in_channels=in_channels, depth=2, start_filters=start_filters, dropout=0.1)
    st = uo.synthetic_state(23, **params)
    g = torch.Generator().manual_seed(8)
    n, h, w = 3, 24, 64
    x = torch.randn(n, in_channels, h, w, generator=g)
    _, sites = uo.unet_plan(**params)
    masks = uo.sample_masks(sites, n, 0.3, g)
    m = _model(params, st, dev)
    # one plan for everything below (the kernel choice of the other layers depends on the plan's batch size)
    assert m.layer_table(h, w, 2 * n)[0]['kernel'] == 'conv3x3_first<T8x32,K36>'
    ref = uo.unet_forward(st, x, masks, **params).numpy()
    out = m(x.to(dev), masks).cpu().numpy()
    assert _maxdiff(out, ref) < LOGIT_TOL
    assert _maxdiff(m(x.to(dev)).cpu().numpy(), uo.unet_forward(st, x, None, **params).numpy()) < LOGIT_TOL
    assert np.array_equal(m(x[:2].to(dev), [mk[:2] for mk in masks]).cpu().numpy(), out[:2])
    # two passes as one batch of 2 n samples = the same statistics as two single passes, bit for bit
    masks2 = uo.sample_masks(sites, n, 0.3, g)
    s1 = steps.McStatistics(n, 2, h, w, dev)
    m.forward_accumulate(x.to(dev), s1, masks)
    m.forward_accumulate(x.to(dev), s1, masks2)
    s2 = steps.McStatistics(n, 2, h, w, dev)
    m.forward_accumulate(x.to(dev), s2, [masks, masks2], passes=2)
    assert torch.equal(s1.blob, s2.blob)
    m_t = _model(params, st, dev, conv_first=0)
    assert m_t.layer_table(h, w, 2 * n)[0]['kernel'].startswith('conv3x3_igemm')
    out_t = m_t(x.to(dev), masks).cpu().numpy()
    assert _maxdiff(out_t, ref) < LOGIT_TOL
    assert _maxdiff(out, out_t) < LOGIT_TOL


def test_forward_is_bitwise_repeatable(dev):
    """Ten forwards of the BraTS-sized network on the same input and masks give the same bits (no atomics, no races: the
    store-data hazard behind csrc/rcu_wino_common.h: wino_store16 showed up as a few voxels differing from run to run)."""
    from oracle import unet_oracle as uo
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05)
    st = uo.synthetic_state(24, **params)
    g = torch.Generator().manual_seed(9)
    n, h, w = 8, 192, 128
    x = torch.randn(n, 4, h, w, generator=g).to(dev)
    _, sites = uo.unet_plan(**params)
    masks = uo.sample_masks(sites, n, 0.3, g)
    m = _model(params, st, dev)
    packed = m.pack_masks(masks, n, dev)
    first = m(x, packed).clone()
    for _ in range(9):
        assert torch.equal(m(x, packed), first)


@pytest.mark.parametrize('is_log_sigma', [False, True])
def test_aleatoric_mc_step_is_the_composition_of_the_reference_pieces(dev, is_log_sigma):
    """Extension step for BASELINE's "aleatoric + MC" config (csrc: rcu_unet_forward_accumulate_sigma): T stochastic passes of
    a sigma-head U-Net.  Oracle = the pieces the reference has, composed on the CPU: per pass softmax(logits_t) and
    |raw_t| / exp(raw_t) (brats_test_aleatoric.py:66-69), then MultiPredictionSummary's mean / entropy over the passes
    (customsteps.py:50-71) and the mean of the sigmas."""
    from oracle import unet_oracle as uo
    from rcu_amd import steps
    params = dict(nb_classes=2, in_channels=4, depth=3, start_filters=32, dropout=0.05, sigma_out=True)
    st = uo.synthetic_state(26, **params)
    g = torch.Generator().manual_seed(11)
    n, h, w, T = 2, 64, 64, 4
    x = torch.randn(n, 4, h, w, generator=g)
    _, sites = uo.unet_plan(**params)
    mask_sets = [uo.sample_masks(sites, n, 0.3, g) for _ in range(T)]
    m = _model(params, st, dev)
    ctx = steps.TorchTestContext('cuda', m)
    bc = steps.BatchContext({'images': x}, 0)
    steps.AleatoricMcPredictStep(T, is_log_sigma=is_log_sigma, do_mi=True, masks=mask_sets)(bc, None, ctx)
    assert set(bc.output) == {'ws_probabilities', 'ws_sigma', 'multi_probabilities', 'sigma'}
    steps.MultiPredictionSummary(do_mi=True)(bc, None, ctx)
    assert set(bc.output) == {'ws_probabilities', 'ws_sigma', 'sigma', 'probabilities', 'entropy', 'mutual_info'}
    act = (lambda r: r.double().exp()) if is_log_sigma else (lambda r: r.double().abs())
    ps, sg = [], []
    for mk in mask_sets:
        lg, raw = uo.unet_forward(st, x, mk, **params)
        ps.append(torch.softmax(lg.double(), 1))
        sg.append(act(raw))
    p_mean = torch.stack(ps).mean(0)
    ent = -(torch.where(p_mean > 0, p_mean * p_mean.log(), torch.zeros_like(p_mean))).sum(1, keepdim=True)
    ent_t = torch.stack([-(torch.where(p > 0, p * p.log(), torch.zeros_like(p))).sum(1, keepdim=True) for p in ps]).mean(0)
    assert _maxdiff(bc.output['probabilities'].cpu().numpy(), p_mean.numpy()) < 1e-6
    assert _maxdiff(bc.output['entropy'].cpu().numpy(), ent.numpy()) < 1e-6
    assert _maxdiff(bc.output['mutual_info'].cpu().numpy(), (ent - ent_t).numpy()) < 1e-6
    sigma_ref = torch.stack(sg).mean(0).numpy()
    assert _maxdiff(bc.output['sigma'].cpu().numpy(), sigma_ref) < 1e-6 * max(1.0, float(np.abs(sigma_ref).max()))
    lg0, raw0 = uo.unet_forward(st, x, None, **params)
    assert _maxdiff(bc.output['ws_probabilities'].cpu().numpy(), torch.softmax(lg0, 1).numpy()) < 1e-6
    assert _maxdiff(bc.output['ws_sigma'].cpu().numpy(), act(raw0).numpy()) < 1e-6 * max(1.0, float(act(raw0).max()))
    # the sharded runner (world size 1 here; the sigma sums ride in the reduce buffer next to the statistics) gives the same maps
    from rcu_amd import distributed as rdist
    out = rdist.ShardedAleatoricMcRunner(m, T, is_log_sigma=is_log_sigma, do_mi=True).step(x.to(dev), 0, mask_sets)
    assert set(out) == {'probabilities', 'entropy', 'mutual_info', 'sigma', 'ws_probabilities', 'ws_sigma'}
    for key in out:      # (the step ran its passes as one group of T: a plan for 2 * T samples, other kernels at the deep levels)
        assert _maxdiff(out[key].cpu().numpy(), bc.output[key].cpu().numpy()) < 1e-6 * max(1.0, float(out[key].abs().max())), key
    # pass groups of the sigma head (rcu_unet_forward_accumulate_sigma_passes): statistics and sigma sums carry the bits of single passes
    xd = x.to(dev)

    def accumulate(group):
        stats = steps.McStatistics(n, 2, h, w, dev, do_mi=True)
        ssum = torch.zeros((n, 2, h, w), device=dev)
        steps.set_dropout_mode(m, True)
        for t in range(0, T, group):
            sets = mask_sets[t:t + group]
            m.forward_accumulate_sigma(xd, stats, ssum, sets[0] if len(sets) == 1 else sets, is_log_sigma, passes=len(sets))
        steps.set_dropout_mode(m, False)
        return stats.blob.cpu().numpy(), ssum.cpu().numpy()

    m.layer_table(h, w, n * T)          # one plan for every group size
    single = accumulate(1)
    for group in (2, 3, 4):
        blob, ssum = accumulate(group)
        assert np.array_equal(blob, single[0]) and np.array_equal(ssum, single[1]), group
    # errors of the boundary
    with pytest.raises(ValueError):
        steps.AleatoricMcPredictStep(2)(steps.BatchContext({'images': x}, 0), None, object())
    plain = dict(params, sigma_out=False)
    with pytest.raises(ValueError):
        steps.AleatoricMcPredictStep(2)(steps.BatchContext({'images': x}, 0), None,
                                        steps.TorchTestContext('cuda', _model(plain, uo.synthetic_state(26, **plain), dev)))


def test_tensors_beyond_2gb_take_the_64bit_kernels(dev):
    """272 ISIC-sized images in one batch: the full-resolution activations (272 x 256 x 256 x 32 floats = 2.3 GB) are beyond the
    32-bit buffer offsets of the Winograd kernels, so those layers must run on the direct kernels (64-bit addressing) -- and the
    last images of the batch, whose pixels sit above 2 GB, must come out like in a small batch."""
    from oracle import unet_oracle as uo
    params = dict(nb_classes=2, in_channels=3, depth=4, start_filters=32, dropout=0.05)
    st = uo.synthetic_state(25, **params)
    g = torch.Generator().manual_seed(10)
    n = 272
    x = torch.rand(n, 3, 256, 256, generator=g).to(dev)
    m_small = _model(params, st, dev)
    m_big = _model(params, st, dev)
    big = {row['name']: row['kernel'] for row in m_big.layer_table(256, 256, n)}
    small = {row['name']: row['kernel'] for row in m_small.layer_table(256, 256, 8)}
    name = 'down_convs.0.block.block.1.conv2d_batch_relu.conv'
    assert 'winograd' in small[name] and 'igemm' in big[name]
    assert any('winograd' in k for k in big.values())            # the low-resolution levels stay below 2 GB
    out = m_big(x)
    for sl in (slice(0, 8), slice(n - 8, n)):
        assert _maxdiff(out[sl].cpu().numpy(), m_small(x[sl]).cpu().numpy()) < LOGIT_TOL


def test_winograd_sigma_head_and_eval_mode(dev):
    """conv_cls.0 + conv_sigma.0 as one 64-channel Winograd unit with two dropout sites (mask / mask2), eval mode (no
    masks) and a deterministic configuration without dropout modules."""
    from oracle import unet_oracle as uo
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 4, 64, 64, generator=g)
    for params in (dict(nb_classes=2, in_channels=4, depth=3, start_filters=32, dropout=0.05, sigma_out=True),
                   dict(nb_classes=3, in_channels=4, depth=3, start_filters=32, dropout=None)):
        st = uo.synthetic_state(22, **params)
        m = _model(params, st, dev)
        assert any('winograd' in row['kernel'] for row in m.layer_table(64, 64, 2))
        _, sites = uo.unet_plan(**params)
        for mk in ((None, uo.sample_masks(sites, 2, 0.3, g)) if sites else (None,)):
            ref = uo.unet_forward(st, x, mk, **params)
            out = m(x.to(dev), mk)
            if params.get('sigma_out'):
                assert _maxdiff(out[0].cpu().numpy(), ref[0].numpy()) < LOGIT_TOL
                assert _maxdiff(out[1].cpu().numpy(), ref[1].numpy()) < LOGIT_TOL
            else:
                assert _maxdiff(out.cpu().numpy(), ref.numpy()) < LOGIT_TOL


@pytest.mark.parametrize('head_winograd4', [1, 0])
def test_fused_head_matches_head_kernel_bitwise(dev, head_winograd4):
    """conv_cls.0 with the 1x1 classifier + softmax + statistics in its epilogue (csrc/rcu_wino4.hip, wino4_epilogue_head: the shipped plan;
    csrc/rcu_wino.hip, wino_epilogue_head: plan option head_winograd4=0) against the same conv kernel's plain epilogue + the separate head kernel (UNet.set_fuse_head(False) = rcu_unet_set_fuse_head): logits, MC statistics (incl. variance / mutual information) and
    pass groups (the fused epilogue runs the passes of a tile back to back on the workgroup that owns it, in pass order -- the
    order in which head_kernel adds them) must carry the same bits."""
    from oracle import unet_oracle as uo
    from rcu_amd import steps
    params = dict(nb_classes=2, in_channels=4, depth=3, start_filters=32, dropout=0.05)
    st = uo.synthetic_state(23, **params)
    m = _model(params, st, dev, head_winograd4=head_winograd4)
    head_kernel = [row['kernel'] for row in m.layer_table(64, 64, 12) if row['name'].startswith('conv_cls.0')]
    assert head_kernel == ['conv3x3_winograd4<T32x32,N32,K8>' if head_winograd4 else 'conv3x3_winograd<T16x32,N32,K8>']
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 4, 64, 64, generator=g)
    _, sites = uo.unet_plan(**params)
    T = 4
    mask_sets = [uo.sample_masks(sites, 3, 0.3, g) for _ in range(T)]

    def run():
        logits = m(x.to(dev), mask_sets[0]).cpu().numpy()
        bc = steps.BatchContext({'images': x.clone()}, 0)
        ctx = steps.TorchTestContext('cuda', m)
        steps.McPredictStep(T, do_mi=True, do_var=True, masks=mask_sets)(bc, None, ctx)
        steps.MultiPredictionSummary(do_mi=True, do_var=True)(bc, None, ctx)
        return logits, {k: v.cpu().numpy() for k, v in bc.output.items()}

    def run_single_passes(do_mi, do_var, exact=False):   # one forward per pass: the fused epilogue updates the statistics itself
        stats = steps.McStatistics(3, 2, 64, 64, dev, do_mi=do_mi, do_var=do_var, exact=exact)
        for ms in mask_sets:
            m.forward_accumulate(x.to(dev), stats, ms, passes=1)
        return stats.blob.cpu().numpy()

    def run_groups(do_mi, do_var, exact, passes):   # `passes` passes per launch: sample t * 3 + i is image i under mask rows [t * 3 + i]
        stats = steps.McStatistics(3, 2, 64, 64, dev, do_mi=do_mi, do_var=do_var, exact=exact)
        for t in range(0, T, passes):
            group = mask_sets[t:t + passes]
            m.forward_accumulate(x.to(dev), stats, group if len(group) > 1 else group[0], passes=len(group))
        return stats.blob.cpu().numpy()

    flag_sets = ((False, False, False), (True, False, False), (True, True, False), (True, False, True), (True, True, True))   # (mi, var, exact)
    m.set_fuse_head(True)
    lf, of = run()
    bf = [run_single_passes(*f) for f in flag_sets]
    gf = [run_groups(*f, passes) for f in flag_sets for passes in (2, 3, 4)]
    m.set_fuse_head(False)
    lu, ou = run()
    bu = [run_single_passes(*f) for f in flag_sets]
    gu = [run_groups(*f, passes) for f in flag_sets for passes in (2, 3, 4)]
    assert np.array_equal(lf, lu)
    assert set(of) == set(ou)
    for k in of:
        assert np.array_equal(of[k], ou[k]), k
    for a_, b_ in zip(bf, bu):
        assert np.array_equal(a_, b_)
    for i, (a_, b_) in enumerate(zip(gf, gu)):
        assert np.array_equal(a_, b_), i
        assert np.array_equal(a_, bf[i // 3]), i      # and a group adds up to what its passes add one by one
    ref = uo.unet_forward(st, x, mask_sets[0], **params).numpy()
    assert _maxdiff(lf, ref) < LOGIT_TOL


def test_stream_lanes_match_one_lane_and_the_oracle(dev):
    """rcu_amd.distributed: the launches of a volume spread over 2 or 3 HIP streams (a workspace and a statistics blob per lane,
    merged by addition) -- MC passes in pairs with all statistics, the sigma-head extension, ensemble members.  The summary must
    agree with the one-lane run to float32 summation order, be the same bits run after run (the assignment launch -> lane is
    fixed), and the MC case must agree with the oracle under the same masks."""
    from oracle import summary_oracle as so
    from oracle import unet_oracle as uo
    from rcu_amd import distributed as rdist
    params = dict(nb_classes=2, in_channels=4, depth=3, start_filters=32, dropout=0.1)
    st = uo.synthetic_state(29, **params)
    g = torch.Generator().manual_seed(12)
    n, h, w, T = 4, 64, 64, 7
    x = torch.randn(n, 4, h, w, generator=g)
    _, sites = uo.unet_plan(**params)
    mask_sets = [uo.sample_masks(sites, n, 0.1, g) for _ in range(T)]
    m = _model(params, st, dev)
    xd = x.to(dev)

    def run(lanes, **kw):
        r = rdist.ShardedMcRunner(m, T, ws_pass=True, do_mi=True, do_var=True, pass_group=2, lanes=lanes, **kw)
        return {k: v.cpu().numpy() for k, v in r.step(xd, 3, mask_sets).items()}

    one = run(1)
    ws, multi = so.mc_probabilities(lambda xx, mk: uo.unet_forward(st, xx, mk, **params), x, mask_sets)
    ref = so.multi_prediction_summary(multi, do_mi=True, do_var=True)
    ref['ws_probabilities'] = ws
    for lanes in (2, 3):
        a, b = run(lanes), run(lanes)
        assert set(a) == set(one) == set(ref)
        for k in a:
            assert np.array_equal(a[k], b[k]), (lanes, k)
            assert _maxdiff(a[k], one[k]) < 1e-6, (lanes, k)
            assert _maxdiff(a[k], np.asarray(ref[k])) < PROB_TOL, (lanes, k)
    # masks drawn by the runner (seed, volume, pass): the same samples on every lane count
    drawn = [{k: v.cpu().numpy() for k, v in rdist.ShardedMcRunner(m, T, seed=5, pass_group=2, lanes=lanes).step(xd, 1).items()}
             for lanes in (1, 2)]
    for k in drawn[0]:
        assert _maxdiff(drawn[0][k], drawn[1][k]) < 1e-6, k

    # sigma-head extension: the per-pass sigma sums ride with the statistics, lane by lane
    sp = dict(params, sigma_out=True)
    sts = uo.synthetic_state(31, **sp)
    ms = _model(sp, sts, dev)
    _, sigma_sites = uo.unet_plan(**sp)
    sigma_masks = [uo.sample_masks(sigma_sites, n, 0.1, g) for _ in range(T)]
    outs = [{k: v.cpu().numpy() for k, v in rdist.ShardedAleatoricMcRunner(ms, T, lanes=lanes).step(xd, 2, sigma_masks).items()} for lanes in (1, 2)]
    assert 'sigma' in outs[0] and 'ws_sigma' in outs[0]
    for k in outs[0]:
        assert _maxdiff(outs[0][k], outs[1][k]) < 1e-6, k

    # ensemble members on two lanes
    members = [_model(params, uo.synthetic_state(40 + i, **params), dev) for i in range(3)]
    outs = [{k: v.cpu().numpy() for k, v in rdist.ShardedEnsembleRunner(members, do_mi=True, lanes=lanes).step(xd).items()} for lanes in (1, 2)]
    for k in outs[0]:
        assert _maxdiff(outs[0][k], outs[1][k]) < 1e-6, k


def test_unet_g11_reference_digest(golden, dev):
    """Full-width weights rebuilt by replaying the reference constructor's draws; the committed strided
    logits came from the reference itself."""
    from oracle import unet_oracle as uo
    g = golden('g11_fullsize_digest')
    p = golden_params(g)
    st = uo.reference_init_state(int(g['seed']), bn_seed=int(g['seed']) + 1000, **p)
    y = _model(p, st, dev)(torch.from_numpy(g['x']).to(dev)).cpu().numpy().reshape(-1)
    assert _maxdiff(y[::int(g['stride'])], g['logits_strided']) < LOGIT_TOL


def test_unet_plan_cache_is_bounded(dev):
    from oracle import unet_oracle as uo
    from rcu_amd.model import UNet
    params = dict(nb_classes=2, in_channels=3, depth=2, start_filters=4, dropout=None)
    state = uo.synthetic_state(1, **params)
    m = UNet(**params)
    m.load_state_dict(state)
    m = m.to(dev)
    first = None
    for k in range(UNet.MAX_HANDLES + 3):
        x = torch.randn(1, 3, 16 + 4 * k, 32, generator=torch.Generator().manual_seed(k))
        y = m(x.to(dev)).cpu()
        assert _maxdiff(y, uo.unet_forward(state, x, None, **params)) < LOGIT_TOL
        first = first if first is not None else (x, y)
        assert len(m._handles) <= UNet.MAX_HANDLES
    assert torch.equal(m(first[0].to(dev)).cpu(), first[1])      # an evicted plan is rebuilt transparently


def test_unet_batch_split_invariance_and_errors(dev):
    from oracle import unet_oracle as uo
    from rcu_amd import _lib
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05)
    m = _model(params, uo.synthetic_state(3, **params), dev)
    x = torch.randn(6, 4, 48, 32, generator=torch.Generator().manual_seed(1)).to(dev)
    full = m(x)
    part = m(x[2:5].contiguous())
    assert torch.equal(full[2:5], part)          # slices are independent and the kernels deterministic
    assert torch.equal(m(x), full)
    ones = [np.ones((6, c), np.float32) for _, c in m.dropout_sites()]
    assert torch.equal(m(x, ones), full)         # all-ones masks == eval mode, bit for bit
    with pytest.raises(_lib.RcuError):
        m(torch.zeros(1, 4, 8, 32, device=dev))   # smaller than 2^depth: nothing left at the bottom level
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 32, 32))              # CPU tensor: no fallback
    with pytest.raises(ValueError):
        m(x, ones[:-1])


# ---------------------------------------------------------------------------------- aggregation
def test_mc_summary_golden(golden, dev):
    from rcu_amd import steps
    g = golden('g6_mc_summary')
    for case in range(3):
        multi = torch.from_numpy(g['multi_{}'.format(case)]).to(dev)
        bc = steps.BatchContext({}, 0)
        bc.output['multi_probabilities'] = multi
        steps.MultiPredictionSummary(do_mi=True, do_var=True)(bc, None, None)
        assert 'multi_probabilities' not in bc.output
        for k, tol in (('probabilities', 1e-6), ('entropy', 2e-6), ('mutual_info', 2e-6), ('variance', 1e-7)):
            ref = g['{}_{}'.format(k, case)]
            assert bc.output[k].shape == ref.shape
            assert _maxdiff(bc.output[k].cpu().numpy(), ref) < tol, k
        # default flags: float32 statistics, two outputs only
        bc = steps.BatchContext({}, 0)
        bc.output['multi_probabilities'] = multi
        steps.MultiPredictionSummary()(bc, None, None)
        assert set(bc.output) == {'probabilities', 'entropy'}
        assert _maxdiff(bc.output['probabilities'].cpu().numpy(), g['probabilities_{}'.format(case)]) < 1e-6
        assert _maxdiff(bc.output['entropy'].cpu().numpy(), g['entropy_{}'.format(case)]) < 2e-6


def test_mc_step_end_to_end_golden(golden, dev):
    """McPredictStep + MultiPredictionSummary through the fused forward+softmax+accumulate path with the
    reference's own dropout masks injected; also the materialised (reference-shaped) path."""
    from rcu_amd import steps
    g = golden('g7_mc_step')
    m = _model(golden_params(g), golden_state(g), dev)
    T, S = int(g['T']), len(g['sites'])
    mask_sets = [[g['mask_{}_{}'.format(t, s)] for s in range(S)] for t in range(T)]
    ctx = steps.TorchTestContext('cuda', m)
    for materialize in (False, True):
        bc = steps.BatchContext({'images': torch.from_numpy(g['x'])}, 0)   # float64 input: the step casts
        steps.McPredictStep(T, do_mi=True, do_var=True, materialize=materialize, masks=mask_sets)(bc, None, ctx)
        assert not m.mc_active()                                          # dropout switched back off
        if materialize:
            assert _maxdiff(bc.output['multi_probabilities'].cpu().numpy(), g['multi_probabilities']) < PROB_TOL
        steps.MultiPredictionSummary(do_mi=True, do_var=True)(bc, None, ctx)
        assert set(bc.output) == set(g['out_keys'])
        for k in g['out_keys']:
            assert _maxdiff(bc.output[k].cpu().numpy(), g['out::' + k]) < PROB_TOL, k
    with pytest.raises(ValueError):
        steps.McPredictStep(1)(steps.BatchContext({'images': torch.from_numpy(g['x'])}, 0), None, object())
    # In the reference the summary's flags alone decide which outputs exist (customsteps.py:44-48).  A default McPredictStep tracks
    # mean + entropy only; a summary that asks for more replays the passes under the same masks (injected or sampled: the device
    # generator is put back), so that every output belongs to the same T samples.
    bc = steps.BatchContext({'images': torch.from_numpy(g['x'])}, 0)
    steps.McPredictStep(T, masks=mask_sets)(bc, None, ctx)
    steps.MultiPredictionSummary(do_mi=True, do_var=True)(bc, None, ctx)
    assert set(bc.output) == set(g['out_keys'])
    for k in g['out_keys']:
        assert _maxdiff(bc.output[k].cpu().numpy(), g['out::' + k]) < PROB_TOL, k
    torch.manual_seed(11)
    bc = steps.BatchContext({'images': torch.from_numpy(g['x'])}, 0)
    steps.McPredictStep(4)(bc, None, ctx)            # sampled masks, default statistics
    stats = bc.output['multi_probabilities']
    after = torch.cuda.get_rng_state(dev)
    stacked = stats.as_tensor()                      # what a foreign step reading the key would want: [T, N, C, H, W]
    assert tuple(stacked.shape) == (4,) + tuple(g['x'].shape[:1]) + (2,) + tuple(g['x'].shape[2:])
    assert torch.equal(torch.cuda.get_rng_state(dev), after)       # the replay leaves the generator where it was
    steps.MultiPredictionSummary(do_var=True)(bc, None, ctx)
    assert not m.mc_active()
    assert _maxdiff(bc.output['probabilities'].cpu().numpy(), stacked.mean(0).cpu().numpy()) < 1e-6
    assert _maxdiff(bc.output['variance'].cpu().numpy(), stacked.var(0).mean(1, keepdim=True).cpu().numpy()) < 1e-6
    blob = steps.McStatistics(1, 2, 8, 8, dev)       # statistics without a recipe cannot be replayed
    with pytest.raises(ValueError):
        blob.as_tensor()


def test_mc_sampled_dropout_statistics(dev):
    """Sampled masks follow Bernoulli(1-p)/(1-p) and MC passes differ from the eval pass."""
    from oracle import unet_oracle as uo
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.25)
    m = _model(params, uo.synthetic_state(4, **params), dev)
    from rcu_amd import steps
    steps.set_dropout_mode(m, True)
    torch.manual_seed(0)
    mk = m.sample_masks(64, dev)
    vals = torch.unique(mk).cpu().numpy()
    assert np.allclose(vals, [0.0, 1 / 0.75])
    assert abs(float((mk > 0).float().mean()) - 0.75) < 0.01
    x = torch.randn(2, 4, 32, 32, device=dev)
    a, b = m(x), m(x)
    assert not torch.equal(a, b)
    steps.set_dropout_mode(m, False)
    assert torch.equal(m(x), m(x))


def test_ensemble_step(dev):
    from oracle import summary_oracle as so
    from oracle import unet_oracle as uo
    from rcu_amd import steps
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05)
    states = [uo.synthetic_state(20 + k, **params) for k in range(3)]
    models = [_model(params, st, dev) for st in states]
    x = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(2))
    ref = so.multi_prediction_summary(
        so.ensemble_probabilities([lambda xx, mk, st=st: uo.unet_forward(st, xx, mk, **params) for st in states], x),
        True, True)
    ctx = steps.TorchTestContext('cuda', models[0])
    for materialize in (False, True):
        bc = steps.BatchContext({'images': x.clone()}, 0)
        steps.EnsemblePredictionStep(models[1:], do_mi=True, do_var=True, materialize=materialize)(bc, None, ctx)
        steps.MultiPredictionSummary(do_mi=True, do_var=True)(bc, None, ctx)
        for k in ('probabilities', 'entropy', 'mutual_info', 'variance'):
            assert _maxdiff(bc.output[k].cpu().numpy(), ref[k].numpy()) < PROB_TOL, k


def test_aggregation_properties_full_size(dev):
    """BraTS-sized statistics (160 x 2 x 192 x 128): T identical passes -> mean == p, zero variance and
    mutual information; a permutation of the passes gives the same float64 statistics."""
    from rcu_amd import steps
    n, c, h, w = 160, 2, 192, 128
    g = torch.Generator(device='cuda').manual_seed(0)
    logits = torch.randn(n, c, h, w, device=dev, generator=g) * 4
    p = steps.softmax(logits)
    assert float((p.sum(1) - 1).abs().max()) < 1e-6
    st = steps.McStatistics(n, c, h, w, dev, do_mi=True, do_var=True)
    for _ in range(5):
        st.accumulate(logits)
    out = st.finalize(True, True)
    assert float((out['probabilities'] - p).abs().max()) < 1e-7
    assert float(out['variance'].abs().max()) < 1e-12
    assert float(out['mutual_info'].abs().max()) < 1e-6
    others = [torch.randn(n, c, h, w, device=dev, generator=g) for _ in range(3)]
    sa = steps.McStatistics(n, c, h, w, dev, do_mi=False, do_var=True)
    sb = steps.McStatistics(n, c, h, w, dev, do_mi=False, do_var=True)
    for t in others:
        sa.accumulate(t)
    for t in reversed(others):
        sb.accumulate(t)
    assert float((sa.blob - sb.blob).abs().max()) < 1e-12


# ---------------------------------------------------------------------------------- auxiliary_feat: features + PostNet
def test_postnet_golden_and_features_view(golden, dev):
    """bin-dl/brats_test_auxiliary_feat.py:67-77: segmentation forward with provide_features, PostNet on the
    features -- against the reference's own outputs (G13)."""
    from rcu_amd.model import PostNet, UNet
    g = golden('g13_postnet')
    params = golden_params(g)
    model = UNet(**params).to(dev)
    model.load_state_dict({k[len('unet::'):]: torch.as_tensor(g[k]) for k in g if k.startswith('unet::')})
    x = torch.as_tensor(g['x']).to(dev)
    logits = model(x)
    assert _maxdiff(logits.cpu(), g['segm_logits']) < LOGIT_TOL
    feats = model.features
    assert tuple(feats.shape) == g['features'].shape
    assert _maxdiff(feats.cpu(), g['features']) < LOGIT_TOL
    for tag, nb_convs, classes in (('post', 3, 2), ('post5', 5, 3)):
        post = PostNet(params['start_filters'], classes, nb_convs=nb_convs).to(dev)
        post.load_state_dict({k[len(tag) + 2:]: torch.as_tensor(g[k]) for k in g if k.startswith(tag + '::')})
        ref = g['logits' if tag == 'post' else 'logits5']
        assert _maxdiff(post(model.features).cpu(), ref) < LOGIT_TOL                         # in place on the workspace tensor
        assert _maxdiff(post(torch.as_tensor(g['features']).to(dev)).cpu(), ref) < LOGIT_TOL  # generic NCHW input
    with pytest.raises(RuntimeError):
        post(torch.zeros(1, params['start_filters'], 8, 8))
    with pytest.raises(ValueError):
        post(torch.zeros(1, 7, 8, 8, device=dev))


def test_postnet_full_width_vs_oracle(dev):
    """32 feature channels (start_filters 32, config/train_brats_auxiliary_feat.yaml:7-9) on a BraTS slice batch;
    ragged voxel count (not a multiple of 32) and a single voxel."""
    from oracle import unet_oracle as uo
    from rcu_amd.model import PostNet
    for (n, h, w), classes, nb_convs in (((3, 192, 128), 2, 3), ((1, 5, 7), 2, 3), ((1, 1, 1), 4, 1), ((2, 16, 16), 32, 0)):
        state = uo.postnet_synthetic_state(5 + n, 32, classes, nb_convs)
        post = PostNet(32, classes, nb_convs=nb_convs).to(dev)
        post.load_state_dict(state)
        x = torch.randn(n, 32, h, w, generator=torch.Generator().manual_seed(h))
        ref = uo.postnet_forward(state, x, nb_convs)
        assert _maxdiff(post(x.to(dev)).cpu(), ref) < 2e-5
    with pytest.raises(Exception):
        PostNet(128, 2).to(dev)(torch.zeros(1, 128, 8, 8, device=dev))    # wider than the kernel handles (96 channels): loud


def test_postnet_wide_and_mc_dropout_golden(golden, dev):
    """Reference golden G16: PostNet over 64 / 48 / 40 feature channels (U-Nets with start_filters > 32) and MC-dropout inside
    PostNet (a Dropout2d between conv and BatchNorm of every hidden unit) under the reference's masks; sampled masks too."""
    from rcu_amd import steps
    from rcu_amd.model import PostNet
    g = golden('g16_postnet_wide')
    for tag in ('a', 'b', 'c'):
        c, classes, convs = (int(v) for v in g['shape_' + tag])
        post = PostNet(c, classes, nb_convs=convs).to(dev)
        post.load_state_dict({k[len('post_{}::'.format(tag)):]: torch.as_tensor(v) for k, v in g.items() if k.startswith('post_{}::'.format(tag))})
        assert _maxdiff(post(torch.as_tensor(g['features_' + tag]).to(dev)).cpu(), g['logits_' + tag]) < 2e-5, tag
    post = PostNet(32, 2, nb_convs=3, dropout=0.3).to(dev)
    post.load_state_dict({k[len('post_d::'):]: torch.as_tensor(v) for k, v in g.items() if k.startswith('post_d::')})
    f = torch.as_tensor(g['features_d']).to(dev)
    assert _maxdiff(post(f).cpu(), g['logits_d_eval']) < 2e-5
    masks = [g['mask_d_{}'.format(s)] for s in range(3)]
    assert _maxdiff(post(f, masks).cpu(), g['logits_d_mc']) < 2e-5
    steps.set_dropout_mode(post, True)
    torch.manual_seed(3)
    a, b = post(f), post(f)
    steps.set_dropout_mode(post, False)
    assert float((a - b).abs().max()) > 1e-3                      # two stochastic passes differ ...
    assert _maxdiff(post(f).cpu(), g['logits_d_eval']) < 2e-5     # ... and eval mode is back afterwards


# ---------------------------------------------------------------------------------- calibration
def test_ece_golden_bit_exact_bins(golden, dev):
    from rcu_amd import evaluation as ev
    g = golden('g8_ece')
    for tag in ('a', 'b'):
        assert np.array_equal(ev.bin_ids(g[tag + '_p']).astype(np.int64), g[tag + '_binids'])
    p2 = np.stack([1 - g['a_p'], g['a_p']], -1)
    for tag, mask in (('masked', g['a_mask']), ('nomask', None)):
        bins = {}
        ece = ev.ece_binary(p2, g['a_target'], mask=mask, out_bins=bins)
        assert np.array_equal(bins['bins_count'], g['a_bins_count_' + tag])          # integers: exact
        assert np.array_equal(bins['bins_non_zero'], g['a_bins_non_zero_' + tag])
        assert np.array_equal(bins['bins_positive_fraction'], g['a_bins_positive_fraction_' + tag])
        assert _maxdiff(bins['bins_avg_confidence'], g['a_bins_avg_confidence_' + tag]) < 1e-12   # f64 sum order
        assert abs(ece - float(g['a_ece_' + tag])) < 1e-12
    for tag in ('b', 'c'):
        p = g[tag + '_p']
        bins = {}
        ece = ev.ece_binary(np.stack([1 - p, p], -1), g[tag + '_target'], out_bins=bins)
        assert np.array_equal(bins['bins_count'], g[tag + '_bins_count'])
        assert abs(ece - float(g[tag + '_ece'])) < 1e-12
    for wgt in ('log_proportion', 'power_proportion', 'mean_proportion'):
        assert abs(ev.ece_binary(p2, g['a_target'], mask=g['a_mask'], bin_weighting=wgt) - float(g['d_ece_' + wgt])) < 1e-12
    assert abs(ev.ece_binary(p2, g['a_target'], threshold_range=(0.2, 0.9)) - float(g['d_ece_thresrange'])) < 1e-12
    assert abs(ev.ece_binary(p2, g['a_target'], n_bins=5) - float(g['e_ece_5bins'])) < 1e-12
    res = {}
    ev.EceBinaryNumpy(with_mask=True, return_bins=True)({'target': g['a_target'], 'probabilities': p2,
                                                         'mask': g['a_mask']}, res)
    assert sorted(res.keys()) == list(g['d_keys'])
    assert abs(res['ece'] - float(g['d_ece'])) < 1e-12
    with pytest.raises(ValueError):
        ev.ece_binary(np.zeros((4, 4, 3), np.float32), np.zeros((4, 4), np.uint8))


def test_ece_edge_cases(dev):
    from oracle import calib_oracle as co
    from rcu_amd import evaluation as ev
    rng = np.random.RandomState(0)
    # ragged length (scalar path), everything masked out, single voxel
    for n in (1, 7, 1023, 4097):
        p = rng.rand(n).astype(np.float32)
        t = (rng.rand(n) < 0.5).astype(np.uint8)
        cnt, sc, sp = ev.calibration_histogram(p, t)
        rc, rsc, rsp = co.calibration_histogram(p, t)
        assert np.array_equal(cnt[0], rc) and np.array_equal(sp[0], rsp.astype(np.int64))
        assert _maxdiff(sc[0], rsc) < 1e-10
    p = rng.rand(64).astype(np.float32)
    cnt, sc, sp = ev.calibration_histogram(p, np.ones(64, np.uint8), mask=np.zeros(64, bool))
    assert cnt.sum() == 0 and sp.sum() == 0 and sc.sum() == 0


def test_ece_full_size_batched_properties(dev):
    """8 BraTS-sized volumes in one launch: checksum-of-checksums properties + oracle equality on one volume."""
    from oracle import c_oracle
    from oracle import calib_oracle as co
    from rcu_amd import evaluation as ev
    nv, n = 8, 160 * 192 * 128
    g = torch.Generator(device='cuda').manual_seed(1)
    p = torch.rand(nv, n, device=dev, generator=g)
    p = torch.where(torch.rand(nv, n, device=dev, generator=g) < 0.8, p * 0.05, p)   # mostly background
    t = (torch.rand(nv, n, device=dev, generator=g) < p).to(torch.uint8)
    m = (torch.rand(nv, n, device=dev, generator=g) < 0.4).to(torch.uint8)
    cnt, sc, sp = ev.calibration_histogram(p, t, mask=m, n_volumes=nv)
    assert np.array_equal(cnt.sum(1), m.sum(1).cpu().numpy())
    assert np.array_equal(sp.sum(1), (t * m).sum(1).cpu().numpy())
    assert np.allclose(sc.sum(1), (p.double() * m).sum(1).cpu().numpy(), rtol=1e-12)
    thr = co.float32_thresholds(10)
    rc, rsc, rsp = c_oracle.ece_hist(p[3].cpu().numpy(), t[3].cpu().numpy(), m[3].cpu().numpy(), thr)
    assert np.array_equal(cnt[3], rc.astype(np.int64)) and np.array_equal(sp[3], rsp.astype(np.int64))
    assert np.allclose(sc[3], rsc, rtol=1e-12, atol=0)
    e_gpu = ev.ece_from_histogram(cnt[3], sc[3], sp[3])
    e_ref = co.ece_from_histogram(rc.astype(np.int64), rsc, rsp.astype(np.float64))
    assert abs(e_gpu - e_ref) < 1e-12
    cnt2, sc2, sp2 = ev.calibration_histogram(p, t, mask=m, n_volumes=nv)
    assert np.array_equal(sc, sc2)    # run-to-run deterministic, float sums included


def test_calibration_kernels_do_not_depend_on_blocks_per_workgroup(dev):
    """The workgroups of the histogram / count kernels take several consecutive blocks of a volume on large launches (the launcher decides;
    rcu_calib_set_blocks_per_workgroup forces it).  Integer sums: every choice gives the same numbers -- on sizes with a ragged last
    block, a last workgroup with fewer blocks, sizes not divisible by four (scalar path), and with confidences below 2^-16 (rounded to 2^-40:
    the sum stays within 2^-41 per voxel of the float64 sum)."""
    from oracle import c_oracle
    from oracle import calib_oracle as co
    from rcu_amd import _lib
    from rcu_amd import evaluation as ev
    thr = co.float32_thresholds(10)
    ue_thr = (0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95)
    g = torch.Generator(device='cuda').manual_seed(12)
    try:
        _blocks_per_workgroup_cases(dev, g, thr, ue_thr, _lib, ev, c_oracle)
    finally:
        _lib.check(_lib.load().rcu_calib_set_blocks_per_workgroup(0, 0))      # back to the launchers' own rule


def _blocks_per_workgroup_cases(dev, g, thr, ue_thr, _lib, ev, c_oracle):
    for nv, n in ((3, 16384 * 5 + 4 * 77), (2, 16384 * 3), (1, 16384 * 2 + 3), (2, 999)):
        p = torch.rand(nv, n, device=dev, generator=g)
        p = torch.where(torch.rand(nv, n, device=dev, generator=g) < 0.5, p * 3e-6, p)        # half of them below 2^-16
        t = (torch.rand(nv, n, device=dev, generator=g) < 0.3).to(torch.uint8)
        m = (torch.rand(nv, n, device=dev, generator=g) < 0.6).to(torch.uint8)
        pred = (p > 0.5).to(torch.uint8)
        outs = []
        for k_ece, k_unc in ((1, 1), (2, 3), (2, 8)):
            _lib.check(_lib.load().rcu_calib_set_blocks_per_workgroup(k_ece, k_unc))
            cnt, sc, sp = ev.calibration_histogram(p, t, mask=m, n_volumes=nv)
            c = ev.uncertainty_counts(pred, t, p, ue_thr, mask=m, n_volumes=nv)
            outs.append((cnt, sc, sp, c))
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                assert np.array_equal(a, b), (nv, n)
        cnt, sc, sp, c = outs[0]
        for v in range(nv):
            rc, rsc, rsp = c_oracle.ece_hist(p[v].cpu().numpy(), t[v].cpu().numpy(), m[v].cpu().numpy(), thr)
            assert np.array_equal(cnt[v], rc.astype(np.int64)) and np.array_equal(sp[v], rsp.astype(np.int64))
            assert np.all(np.abs(sc[v] - rsc) <= cnt[v] * 2.0 ** -41 + 1e-12 * np.abs(rsc)), (nv, n, v)
        assert np.array_equal(c[..., :4].sum(-1)[:, 0], m.sum(1).cpu().numpy())


def test_uncertainty_counts_golden(golden, dev):
    from rcu_amd import evaluation as ev
    g = golden('g9_uncertainty')
    c = ev.uncertainty_counts(g['prediction'], g['target'], g['uncertainty'], tuple(g['thresholds']))
    assert np.array_equal(c[0], g['counts'])
    cm = ev.uncertainty_counts(g['prediction'], g['target'], g['uncertainty'], (0.5,), mask=g['mask'])
    assert list(cm[0, 0]) == list(g['masked_counts_thr05'])
    for i, thr in enumerate(g['thresholds']):
        res = {}
        ev.UncertaintyErrorDiceNumpy(float(thr))({'prediction': g['prediction'], 'target': g['target'],
                                                 'uncertainty': g['uncertainty']}, res)
        assert [res['dice'], res['recall'], res['precision']] == list(g['derived'][i])
    tpl = ev.uncertainty(g['prediction'], g['target'], g['uncertainty'] > 0.3)
    assert list(tpl) == list(g['counts'][3])
    # end to end from the float32 probability map (round 5): "uncertain" looked up in the table of the reference's own float32 sets
    # (fixture g20): the reference's counts, integer for integer -- no device log, no entropy map
    assert ev.from_p_supported(tuple(g['thresholds']))
    c2 = ev.uncertainty_counts_from_p(g['prediction'], g['target'], g['p'], tuple(g['thresholds']))[0]
    assert np.array_equal(c2, g['counts'])
    c3 = ev.uncertainty_counts_from_p(g['prediction'], g['target'], g['p'], (0.5,), mask=g['mask'])[0, 0]
    assert list(c3) == list(g['masked_counts_thr05'])
    # ... which is what the preparation recipe + the sweep of the bnf_ue action run (ToEntropy leaves an EntropyOfProbability)
    prep, _ = ev.get_uncertainty_preparation('probabilities', 'run')
    to_eval = prep({'probabilities': g['p'].copy(), 'prediction': g['prediction'], 'target': g['target']})
    assert isinstance(to_eval['uncertainty'], ev.EntropyOfProbability)
    res = {}
    ev.UncertaintyAndCorrectionSweep()(to_eval, res)
    for i, t in enumerate(ev.UE_THRESHOLDS):
        assert [res[t][k] for k in ('tp', 'tn', 'fp', 'fn', 'tpu', 'tnu', 'fpu', 'fnu')] == list(g['counts'][i])
    res = {}
    ev.UncertaintyErrorDiceNumpy(0.3)(to_eval, res)
    assert [res['dice'], res['recall'], res['precision']] == list(g['derived'][3])
    # thresholds outside the table take the entropy map (device logf: float tolerance on the map)
    assert not ev.from_p_supported((0.25,)) and not ev.from_p_supported((0.5, 0.3))
    ent = np.asarray(to_eval['uncertainty'])
    assert ent.dtype == np.float64 and _maxdiff(ent, g['uncertainty']) < 1e-6
    res = {}
    ev.UncertaintyAndCorrectionSweep()({'prediction': g['prediction'], 'target': g['target'],
                                        'uncertainty': g['uncertainty']}, res)
    assert [res[t]['tpu'] for t in ev.UE_THRESHOLDS] == list(g['counts'][:, 4])


def test_uncertainty_counts_full_size_properties(dev):
    from oracle import c_oracle
    from rcu_amd import evaluation as ev
    n = 160 * 192 * 128
    g = torch.Generator(device='cuda').manual_seed(2)
    u = torch.rand(n, device=dev, generator=g, dtype=torch.float64) ** 3
    pr = (torch.rand(n, device=dev, generator=g) < 0.1).to(torch.uint8)
    tg = (torch.rand(n, device=dev, generator=g) < 0.1).to(torch.uint8)
    c = ev.uncertainty_counts(pr, tg, u)[0]
    assert np.all(c[:, :4].sum(1) == n)
    assert np.all(np.diff(c[:, 4:].sum(1)) <= 0)           # fewer uncertain voxels as the threshold grows
    assert np.all(c[:, 4:] <= c[:, :4])
    ref = c_oracle.unc_counts(u.cpu().numpy(), pr.cpu().numpy(), tg.cpu().numpy(), None, ev.UE_THRESHOLDS)
    assert np.array_equal(c, ref.astype(np.int64))


def test_ece_histogram_other_bin_counts_and_threshold_neighbours(dev):
    """The histogram kernel finds the bin by floor(p * n_bins) + one table lookup; it must agree with the
    compare-against-every-threshold definition (bin_ids kernel, oracle digitize) for every n_bins, in
    particular on each float32 threshold and its two neighbours, on 0, 1, just above 1, NaN and inf."""
    from oracle import calib_oracle as co
    from rcu_amd import evaluation as ev
    rng = np.random.RandomState(3)
    for n_bins in (1, 2, 3, 7, 10, 15, 16, 17, 32):
        thr = np.asarray(co.float32_thresholds(n_bins), dtype=np.float32)
        special = [np.float32(0), np.float32(1), np.nextafter(np.float32(1), np.float32(2)), np.float32(1e-30),
                   np.float32(-0.0)]
        for t in thr:
            special += [t, np.nextafter(t, np.float32(0)), np.nextafter(t, np.float32(2))]
        p = np.concatenate([np.asarray(special, np.float32), rng.rand(5000).astype(np.float32),
                            (np.arange(0, n_bins + 1) / n_bins).astype(np.float32)])
        tg = (rng.rand(p.size) < 0.5).astype(np.uint8)
        ids = ev.bin_ids(p, n_bins).astype(np.int64)
        valid = p <= 1.0          # above 1 + 1e-8 the reference indexes past the last bin; the build clamps
        assert np.array_equal(ids[valid], co.bin_ids(p[valid], n_bins))
        assert ids.max() <= n_bins - 1
        cnt, sc, sp = ev.calibration_histogram(p, tg, n_bins=n_bins)
        assert np.array_equal(cnt[0], np.bincount(ids, minlength=n_bins))
        assert np.array_equal(sp[0], np.bincount(ids, weights=tg, minlength=n_bins).astype(np.int64))
        assert np.allclose(sc[0], np.bincount(ids, weights=p.astype(np.float64), minlength=n_bins), rtol=1e-13, atol=0)
    weird = np.asarray([np.nan, np.inf, -np.inf, -1.0, 2.0], np.float32)
    cnt, _, _ = ev.calibration_histogram(weird, np.zeros(5, np.uint8))
    assert np.array_equal(cnt[0], np.bincount(ev.bin_ids(weird).astype(np.int64), minlength=10))


def test_uncertainty_counts_threshold_order_and_dtypes(dev):
    """Ascending thresholds take the private-column kernel, any other order the general one; float32 and
    float64 maps, ragged length, masks -- all against the C oracle."""
    from oracle import c_oracle
    from rcu_amd import evaluation as ev
    rng = np.random.RandomState(4)
    for n in (5, 4096, 70001):
        u64 = rng.rand(n) ** 2
        pr = (rng.rand(n) < 0.3).astype(np.uint8)
        tg = (rng.rand(n) < 0.3).astype(np.uint8)
        m = (rng.rand(n) < 0.7).astype(np.uint8)
        for thr in (ev.UE_THRESHOLDS, tuple(reversed(ev.UE_THRESHOLDS)), (0.5, 0.1, 0.9, 0.1), (0.25,),
                    tuple(np.linspace(0.01, 0.99, 16))):
            for u in (u64, u64.astype(np.float32)):
                for mask in (None, m):
                    got = ev.uncertainty_counts(pr, tg, u, thr, mask=mask)[0]
                    ref = c_oracle.unc_counts(u.astype(np.float64), pr, tg, mask, thr)
                    assert np.array_equal(got, ref.astype(np.int64)), (n, thr, u.dtype, mask is None)


def test_uncertainty_counts_at_threshold_neighbours(dev):
    """float32 uncertainty maps are compared through exact float32 thresholds and a monotone cell table (csrc/rcu_calib.hip):
    values on, next to and between every threshold and every cell boundary, outside the threshold range, NaN; threshold
    sets the table takes (one threshold per cell) and sets it must refuse (two thresholds in one cell, a single one)."""
    from oracle import c_oracle
    from rcu_amd import evaluation as ev
    sets = (ev.UE_THRESHOLDS, tuple(np.linspace(0.01, 0.99, 16)), (0.1, 0.1000001, 0.9), (0.3, 0.300001), (-0.5, 0.25, 1.5),
            (0.5,), (1e-30, 0.5, 0.999999))
    for thr in sets:
        t64 = np.asarray(thr, dtype=np.float64)
        vals = [np.float32(0), np.float32(1), np.float32(-1), np.float32(2), np.float32(np.nan), np.float32(np.inf)]
        for t in t64:
            f = np.float32(t)
            for _ in range(3):
                vals.append(f)
                f = np.nextafter(f, np.float32(np.inf), dtype=np.float32)
            f = np.float32(t)
            for _ in range(3):
                f = np.nextafter(f, np.float32(-np.inf), dtype=np.float32)
                vals.append(f)
        lo, hi = np.float32(t64.min()), np.float32(t64.max())
        if hi > lo:   # the cell boundaries of the table: lo + k * (hi - lo) / 63 and their float neighbours
            for k in range(0, 65):
                b = np.float32(lo + np.float32(k) * (hi - lo) / np.float32(63))
                vals += [b, np.nextafter(b, np.float32(np.inf), dtype=np.float32), np.nextafter(b, np.float32(-np.inf), dtype=np.float32)]
        u = np.asarray(vals, dtype=np.float32)
        u = np.tile(u, 40)[:16384 + 7]            # long enough for the vectorised path, ragged tail
        rng = np.random.RandomState(len(thr))
        pr = (rng.rand(u.size) < 0.5).astype(np.uint8)
        tg = (rng.rand(u.size) < 0.5).astype(np.uint8)
        with np.errstate(invalid='ignore'):
            got = ev.uncertainty_counts(pr, tg, u, thr)[0]
            ref = c_oracle.unc_counts(u.astype(np.float64), pr, tg, None, thr)
        assert np.array_equal(got, ref.astype(np.int64)), thr


def test_preparation_golden(golden, dev):
    from rcu_amd import evaluation as ev
    g = golden('g10_prep')
    pred = g['prediction']
    for entry, idp in (('probabilities', 'run'), ('confidence', 'run_rescale'), ('sigma', 'run_rescale')):
        src = g['prob_prep_in_' + entry]
        prep, id_ = ev.get_probability_preparation(entry, 'run')
        out = prep({entry: src.copy(), 'prediction': pred.copy()})
        assert id_ == idp and np.array_equal(out['probabilities'], g['prob_prep_out_' + entry])
        prep_u, id_u = ev.get_uncertainty_preparation(entry, 'run', rescale_confidence='subject', rescale_sigma='subject')
        out_u = prep_u({entry: src.copy(), 'prediction': pred.copy()})
        assert id_u == idp
        assert _maxdiff(out_u['uncertainty'], g['unc_prep_out_' + entry]) < 1e-6
    with pytest.raises(ValueError):
        ev.add_background_probability(np.array([0.5, 1.5]))
    with pytest.raises(ValueError):
        ev.ToEntropy()({'probabilities': np.zeros((2, 2, 3))})


def test_prediction_and_foreground(dev):
    from rcu_amd import steps
    p = torch.rand(3, 2, 16, 16, device=dev)
    p[0, :, 0, 0] = 0.5     # tie -> first maximum, as np.argmax
    pred, fg = steps.prediction_and_foreground(p)
    assert np.array_equal(pred.cpu().numpy(), np.argmax(p.permute(0, 2, 3, 1).cpu().numpy(), -1).astype(np.uint8))
    assert torch.equal(fg, p[:, 1])


# ---------------------------------------------------------------------------------- eval driver (8f-1, 8f-2)
def test_eval_driver_csvs_against_oracle(dev, tmp_path):
    """The four actions of bin-eval/eval_uncertainty.py on a tiny synthetic BraTS-style tree: NIfTI files in,
    the reference's CSV files out; numbers checked against the numpy oracle."""
    import csv
    from oracle import calib_oracle as co
    from rcu_amd import evalrun, nifti
    rng = np.random.RandomState(5)
    gt_root, run_dir, base = tmp_path / 'gt' / 'HGG', tmp_path / 'pred', tmp_path / 'eval'
    run_dir.mkdir(parents=True)
    truth = {}
    for sub in ('Brats18_A_1', 'Brats18_B_1'):
        (gt_root / sub).mkdir(parents=True)
        t2 = (rng.rand(6, 16, 16) * (rng.rand(6, 16, 16) > 0.3)).astype(np.float32)
        seg = (rng.rand(6, 16, 16) < 0.3).astype(np.uint8) * rng.randint(1, 5, (6, 16, 16)).astype(np.uint8)
        p = rng.rand(6, 16, 16).astype(np.float32)
        pred = (p > 0.5).astype(np.uint8)
        for mod, arr in (('flair', t2), ('t1', t2), ('t2', t2), ('t1ce', t2), ('seg', seg)):
            nifti.write(str(gt_root / sub / '{}_{}.nii.gz'.format(sub, mod)), arr)
        nifti.write(str(run_dir / '{}_probabilities.nii.gz'.format(sub)), p)
        nifti.write(str(run_dir / '{}_prediction.nii.gz'.format(sub)), pred)
        truth[sub] = (p, pred, (seg > 0).astype(np.uint8), t2 > 0)
    gts = evalrun.collect_brats_ground_truth(str(tmp_path / 'gt'))
    entry = evalrun.get_eval_data('baseline_mc', str(run_dir), gts, expected_subjects=list(truth))
    evalrun.evaluate_runs([entry], ['minmax', 'ece_dice', 'calib', 'bnf_ue'], str(base), 'foreground')

    def rows(path):
        with open(path, newline='') as f:
            return list(csv.DictReader(f))

    ece_rows = rows(str(base / 'ece_foreground' / 'eval_ece_baseline_mc.csv'))
    assert [r['subject_name'] for r in ece_rows] == sorted(truth)
    for r in ece_rows:
        p, pred, tgt, mask = truth[r['subject_name']]
        assert abs(float(r['ece']) - co.ece_binary(np.stack([1 - p, p], -1), tgt, mask=mask)) < 1e-12
        tp, tn, fp, fn, n = co.confusion_counts(pred, tgt)
        assert [int(r[k]) for k in ('tp', 'tn', 'fp', 'fn', 'n')] == [tp, tn, fp, fn, n]
        assert abs(float(r['dice']) - co.dice_from_counts(tp, fp, fn)) < 1e-15
    cal = rows(str(base / 'calibration' / 'eval_calibration_baseline_mc.csv'))
    for r in cal:
        p, pred, tgt, mask = truth[r['subject_name']]
        cnt, _, _ = co.calibration_histogram(*co.select_foreground(np.stack([1 - p, p], -1), tgt, mask))
        assert [int(r['bins_count_{:02d}'.format(b)]) for b in range(10)] == list(cnt)
    mm = evalrun.read_min_max(str(base / 'minmax' / 'eval_summary_minmax_baseline_mc.csv'))
    # written as str(np.float32) like the reference does: shortest float32 repr
    assert mm == (float(str(min(v[0].min() for v in truth.values()))), float(str(max(v[0].max() for v in truth.values()))))
    for thr in co.UE_THRESHOLDS:
        name = 'eval_uncertainty_baseline_mc_th{}.csv'.format('{:.2f}'.format(thr).replace('.', ''))
        for r in rows(str(base / 'uncertainty' / name)):
            p, pred, tgt, _ = truth[r['subject_name']]
            unc = co.normalised_entropy(co.add_background_probability(p))
            ref = co.correction_metrics(co.uncertainty_counts(pred.astype(bool), tgt.astype(bool), unc > thr))
            for k in ('tp', 'tn', 'fp', 'fn', 'tpu', 'tnu', 'fpu', 'fnu'):      # exact: the probability table, not a device log (fixture g20)
                assert int(r[k]) == ref[k]
            assert set(r) >= {'corrected_dice', 'corrected_accuracy', 'corrected_add_dice', 'dice_benefit_correct'}
    # the fused loop (one upload per subject shared by all actions, subjects batched per launch, files read ahead: the default) writes the
    # bytes of the reference-ordered loop (subject by subject, action by action), whatever the batch size -- all 15 CSV files
    import glob
    import os

    def all_csv(root):
        return {os.path.relpath(f, root): open(f, 'rb').read() for f in sorted(glob.glob(os.path.join(root, '**', '*.csv'), recursive=True))}

    fused = all_csv(str(base))
    assert len(fused) == 1 + 1 + 1 + 11
    for tag, kwargs in (('plain', dict(fused=False)), ('one', dict(batch_subjects=1))):
        other = tmp_path / ('eval_' + tag)
        timing = {}
        evalrun.evaluate_runs([entry], ['minmax', 'ece_dice', 'calib', 'bnf_ue'], str(other), 'foreground', timing=timing, **kwargs)
        assert all_csv(str(other)) == fused, tag
        assert (timing.get('subjects'), timing.get('batches')) == ((None, None) if tag == 'plain' else (2, 2))
    # subsets of the actions, and no mask (the ISIC form of the script)
    for actions, details in ((['ece_dice'], 'foreground'), (['bnf_ue'], ''), (['minmax', 'calib'], '')):
        a, b = tmp_path / 'sub_a', tmp_path / 'sub_b'
        import shutil
        shutil.rmtree(str(a), ignore_errors=True)
        shutil.rmtree(str(b), ignore_errors=True)
        evalrun.evaluate_runs([entry], actions, str(a), details)
        evalrun.evaluate_runs([entry], actions, str(b), details, fused=False)
        assert all_csv(str(a)) == all_csv(str(b)) and len(all_csv(str(a))) >= 1, actions


def test_pass_groups_bit_identical_to_single_passes(dev):
    """rcu_unet_forward_accumulate_passes: T passes of a small batch run as one batch of N*T samples and must
    give exactly the statistics of T single-pass launches (same masks), float32 and float64 blobs, also through
    McPredictStep with injected masks and with sampled masks (statistically)."""
    from oracle import unet_oracle as uo
    from rcu_amd import steps
    from rcu_amd.model import UNet
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=8, dropout=0.3)
    model = UNet(**params)
    model.load_state_dict(uo.synthetic_state(3, **params))
    model = model.to(dev)
    n, h, w, T = 3, 48, 32, 7
    x = torch.randn(n, 4, h, w, generator=torch.Generator().manual_seed(1)).to(dev)
    _, sites = uo.unet_plan(**params)
    g = torch.Generator().manual_seed(2)
    mask_sets = [uo.sample_masks(sites, n, 0.3, g) for _ in range(T)]
    # one plan for single passes and groups, as McPredictStep makes it (UNet.reserve): which kernel a layer gets depends on the batch
    # the plan is sized for (tiles that span two or eight slices want batches they divide), and bit-identity is a property of a plan
    model.reserve(h, w, n * 4)
    for do_mi, do_var in ((False, False), (True, True)):
        a = steps.McStatistics(n, 2, h, w, dev, do_mi, do_var)
        for ms in mask_sets:
            model.forward_accumulate(x, a, ms)
        b = steps.McStatistics(n, 2, h, w, dev, do_mi, do_var)
        model.forward_accumulate(x, b, mask_sets[:4], passes=4)
        model.forward_accumulate(x, b, mask_sets[4:], passes=3)
        assert a.count == b.count == T
        assert torch.equal(a.blob, b.blob)
    ctx = steps.TorchTestContext('cuda', model)
    outs = []
    for group_pixels in (0, 160 * 192 * 128):
        bc = steps.BatchContext({'images': x}, 0)
        steps.McPredictStep(T, do_mi=True, do_var=True, masks=mask_sets, group_pixels=group_pixels)(bc, None, ctx)
        steps.MultiPredictionSummary(do_mi=True, do_var=True)(bc, None, ctx)
        outs.append(bc.output)
    for key in ('probabilities', 'entropy', 'mutual_info', 'variance', 'ws_probabilities'):
        assert torch.equal(outs[0][key], outs[1][key]), key
    # sampled masks: the grouped draw is a different random stream; the mean over many passes must agree
    torch.manual_seed(0)
    bc = steps.BatchContext({'images': x}, 0)
    steps.McPredictStep(200)(bc, None, ctx)
    steps.MultiPredictionSummary()(bc, None, ctx)
    torch.manual_seed(1)
    bc2 = steps.BatchContext({'images': x}, 0)
    steps.McPredictStep(200, group_pixels=0)(bc2, None, ctx)
    steps.MultiPredictionSummary()(bc2, None, ctx)
    assert float((bc.output['probabilities'] - bc2.output['probabilities']).abs().mean()) < 0.02
    with pytest.raises(Exception):
        model.forward_accumulate(x, steps.McStatistics(n, 2, h, w, dev), passes=0)


def test_unet_residual_blocks_golden(golden, dev):
    """ConvResidualBlock (common/model/unet.py:42-60, ``residual=True``): reference golden G14 -- eval pass, an MC pass under the
    reference's masks, and the combination with dropout_center, a sigma head, three classes and a centre-padded size."""
    g = golden('g14_unet_residual')
    m = _model(golden_params(g), golden_state(g), dev)
    assert [s[0] for s in m.dropout_sites()] == list(g['sites'])
    assert sum(k.endswith('.residual.weight') for k in m.state_dict()) == 7      # 3 down + bottom + 3 up blocks
    x = torch.from_numpy(g['x']).to(dev)
    assert _maxdiff(m(x).cpu().numpy(), g['logits_eval']) < LOGIT_TOL
    masks = [g['mask_{}'.format(s)] for s in range(len(g['sites']))]
    assert _maxdiff(m(x, masks).cpu().numpy(), g['logits_mc']) < LOGIT_TOL
    gb = golden('g14_unet_residual_b')
    mb = _model(golden_params(gb), golden_state(gb), dev)
    logits, sigma = mb(torch.from_numpy(gb['x']).to(dev))
    assert _maxdiff(logits.cpu().numpy(), gb['logits']) < LOGIT_TOL
    assert _maxdiff(sigma.cpu().numpy(), gb['sigma']) < LOGIT_TOL


def test_unet_residual_full_width_vs_oracle(dev):
    """Residual blocks at the shipped width on the BraTS slice size: the residual 1x1 convs take the Winograd kernels (a 3x3 unit
    with only its centre tap set), the adding second units the direct ones; with masks, and through the fused statistics path."""
    from oracle import summary_oracle as so
    from oracle import unet_oracle as uo
    from rcu_amd import steps
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05, residual=True)
    st = uo.synthetic_state(41, **params)
    g = torch.Generator().manual_seed(15)
    n, h, w = 2, 192, 128
    x = torch.randn(n, 4, h, w, generator=g)
    _, sites = uo.unet_plan(**params)
    mask_sets = [uo.sample_masks(sites, n, 0.3, g) for _ in range(2)]
    m = _model(params, st, dev)
    rows = m.layer_table(h, w, n)
    assert len(rows) == 23 + 9                                         # nine residual convs
    assert any(r['name'].endswith('.residual') and 'winograd' in r['kernel'] for r in rows)
    for mk in (None, mask_sets[0]):
        ref = uo.unet_forward(st, x, mk, **params).numpy()
        scale = max(1.0, float(np.abs(ref).max()))                     # un-normalised sums: the logits can be of O(10)
        # 1.5 x: the residual sums feed conv_cls.0 activations well above the plain net's, and that unit runs in F(4x4,3x3) since round 5 --
        # measured 2.3e-6 between the two float32 evaluations; with the unit on F(2x2,3x3) (the plan of rounds 1-4) the gate is the plain one
        assert _maxdiff(m(x.to(dev), mk).cpu().numpy(), ref) < 1.5 * LOGIT_TOL * scale
        assert _maxdiff(_model(params, st, dev, head_winograd4=0)(x.to(dev), mk).cpu().numpy(), ref) < LOGIT_TOL * scale
    bc = steps.BatchContext({'images': x}, 0)
    ctx = steps.TorchTestContext('cuda', m)
    steps.McPredictStep(2, masks=mask_sets)(bc, None, ctx)
    steps.MultiPredictionSummary()(bc, None, ctx)
    ws, multi = so.mc_probabilities(lambda xx, mk: uo.unet_forward(st, xx, mk, **params), x, mask_sets)
    ref = so.multi_prediction_summary(multi)
    assert _maxdiff(bc.output['probabilities'].cpu().numpy(), ref['probabilities'].numpy()) < PROB_TOL
    assert _maxdiff(bc.output['ws_probabilities'].cpu().numpy(), ws.numpy()) < PROB_TOL


def test_unet_centre_pad_golden(golden, dev):
    """Reference golden G15: sizes 2^depth does not divide (unet.py:89, 110-116)."""
    g = golden('g15_unet_centre_pad')
    for tag in ('a', 'b', 'c'):
        params = eval(str(g['params_' + tag]), {'__builtins__': {}}, {'dict': dict})
        st = {k[len('sd_{}::'.format(tag)):]: v for k, v in g.items() if k.startswith('sd_{}::'.format(tag))}
        m = _model(params, st, dev)
        y = m(torch.from_numpy(g['x_' + tag]).to(dev)).cpu().numpy()
        assert y.shape == g['logits_' + tag].shape
        assert _maxdiff(y, g['logits_' + tag]) < LOGIT_TOL, tag


def test_unet_no_batchnorm_no_dropout_golden(golden, dev):
    """Reference golden G17: ``bn=False`` (Conv2dBnRelu without BatchNorm, unet.py:16-17; every block and both heads,
    unet.py:128-164) with dropout -- eval pass and an MC pass under the reference's masks --, ``bn=False, dropout=None`` with a
    sigma head, and ``dropout=None`` with BatchNorm: the model-seam switches no shipped config uses."""
    from rcu_amd import steps
    g = golden('g17_unet_no_bn')

    def tagged(tag):
        params = eval(str(g['params_' + tag]), {'__builtins__': {}}, {'dict': dict})
        st = {k[len('sd_{}::'.format(tag)):]: v for k, v in g.items() if k.startswith('sd_{}::'.format(tag))}
        return params, st

    params, st = tagged('a')
    m = _model(params, st, dev)
    assert not any('.bn.' in k for k in m.state_dict()) and [s[0] for s in m.dropout_sites()] == list(g['sites_a'])
    x = torch.from_numpy(g['x_a']).to(dev)
    assert _maxdiff(m(x).cpu().numpy(), g['logits_a_eval']) < LOGIT_TOL
    masks = [g['mask_a_{}'.format(s)] for s in range(len(g['sites_a']))]
    assert _maxdiff(m(x, masks).cpu().numpy(), g['logits_a_mc']) < LOGIT_TOL
    params, st = tagged('b')
    mb = _model(params, st, dev)
    assert mb.dropout_sites() == [] and not mb.mc_active()
    steps.set_dropout_mode(mb, True)                 # nothing to switch on: still deterministic
    logits, sigma = mb(torch.from_numpy(g['x_b']).to(dev))
    assert _maxdiff(logits.cpu().numpy(), g['logits_b']) < LOGIT_TOL
    assert _maxdiff(sigma.cpu().numpy(), g['sigma_b']) < LOGIT_TOL
    params, st = tagged('c')
    mc = _model(params, st, dev)
    assert _maxdiff(mc(torch.from_numpy(g['x_c']).to(dev)).cpu().numpy(), g['logits_c']) < LOGIT_TOL


def test_unet_no_batchnorm_full_width_vs_oracle(dev):
    """bn=False at the shipped width on the BraTS slice size (the Winograd kernels, the fused classifier head): eval pass, an MC pass
    under injected masks, and the fused statistics path against the oracle."""
    from oracle import summary_oracle as so
    from oracle import unet_oracle as uo
    from rcu_amd import steps
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05, bn=False)
    st = uo.synthetic_state(43, **params)
    assert not any('.bn.' in k for k in st)
    g = torch.Generator().manual_seed(16)
    n, h, w = 2, 192, 128
    x = torch.randn(n, 4, h, w, generator=g)
    _, sites = uo.unet_plan(**params)
    mask_sets = [uo.sample_masks(sites, n, 0.3, g) for _ in range(2)]
    m = _model(params, st, dev)
    for mk in (None, mask_sets[0]):
        ref = uo.unet_forward(st, x, mk, **params).numpy()
        scale = max(1.0, float(np.abs(ref).max()))       # no normalisation anywhere: the logits can be of O(10)
        assert _maxdiff(m(x.to(dev), mk).cpu().numpy(), ref) < LOGIT_TOL * scale
    bc = steps.BatchContext({'images': x}, 0)
    ctx = steps.TorchTestContext('cuda', m)
    steps.McPredictStep(2, masks=mask_sets)(bc, None, ctx)
    steps.MultiPredictionSummary()(bc, None, ctx)
    ws, multi = so.mc_probabilities(lambda xx, mk: uo.unet_forward(st, xx, mk, **params), x, mask_sets)
    ref = so.multi_prediction_summary(multi)
    assert _maxdiff(bc.output['probabilities'].cpu().numpy(), ref['probabilities'].numpy()) < PROB_TOL
    assert _maxdiff(bc.output['ws_probabilities'].cpu().numpy(), ws.numpy()) < PROB_TOL


def test_channel_blocked_layout_gives_the_bits_of_channels_last(dev):
    """The activation layout between the Winograd kernels (channel-blocked [N][C/8][H][W][8]) is an addressing
    choice: the same kernels on channels-last tensors (plan option act_layout=1) give the same bits -- logits with and without masks, the
    fused statistics, the feature tap (which keeps its tensor channels-last in both), a sigma head, and a plan that mixes Winograd
    and direct kernels (a size 2^depth does not divide)."""
    from oracle import unet_oracle as uo
    from rcu_amd import steps
    g = torch.Generator().manual_seed(31)
    for params, shape in ((dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05), (8, 4, 192, 128)),
                          (dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05, sigma_out=True), (2, 4, 64, 64)),
                          (dict(nb_classes=3, in_channels=3, depth=3, start_filters=16, dropout=0.2), (3, 3, 88, 72))):
        st = uo.synthetic_state(45, **params)
        x = torch.randn(*shape, generator=g)
        _, sites = uo.unet_plan(**params)
        masks = uo.sample_masks(sites, shape[0], 0.3, g)
        outs = {}
        for layout in ('blocked', 'nhwc'):
            m = _model(params, st, dev, act_layout=int(layout == 'nhwc'))
            m.provide_features = not params.get('sigma_out', False)
            res = [m(x.to(dev)), m(x.to(dev), masks)]
            res = [t for r in res for t in (r if isinstance(r, tuple) else (r,))]
            if m.provide_features:
                res.append(m.features.clone())
            if params['nb_classes'] == 2 and not params.get('sigma_out', False):
                stats = steps.McStatistics(shape[0], 2, shape[2], shape[3], dev, True, True)
                m.forward_accumulate(x.to(dev), stats, masks)
                res.append(stats.blob.clone())
            outs[layout] = [t.cpu() for t in res]
        assert len(outs['blocked']) == len(outs['nhwc'])
        for a, b in zip(outs['blocked'], outs['nhwc']):
            assert torch.equal(a, b), (params, shape)


def test_unet_stress_golden_wide_activations_and_logits(golden, dev):
    """Numerics stress case, reference golden G18 (tests/golden/generate_golden.py: the reference UNet, common/model/unet.py:128-186, with its
    BatchNorm affines x 2.5 and its classifier x 0.5): interior activations of 1e2..1e3 and logits of +-20 under Dropout2d(0.3) -- the ranges
    trained checkpoints have, where the other parity inputs stay below |logit| 1.  Full width on a 192x128 slice pair, so every F(4x4,3x3)
    instantiation runs.  Logits within a bound RELATIVE to their range (2e-5) of the reference's (strided sample) and of the oracle (everywhere);
    probabilities / entropy / mutual information / variance of the three passes within 1e-4; F(2x2,3x3)-only and direct plans on the same
    input within the same bounds."""
    from oracle import summary_oracle as so
    from oracle import unet_oracle as uo
    from rcu_amd import steps
    REL = 2e-5      # measured (tools/stress_margin.py): 2.0e-6 (eval) / 4.2-8.4e-6 (MC passes) with F(4x4,3x3), 1.4 / 2.1-3.7e-6 with F(2x2,3x3), 1.6 / 3.9-7.3e-6 direct;
                    # probabilities 2.7e-5 at most -- the distance to the torch-CPU oracle is float32 summation order at this dynamic range, not Winograd
    g = golden('g18_unet_stress')
    p = golden_params(g)
    st = uo.stress_state(uo.reference_init_state(int(g['seed']), bn_seed=int(g['seed']) + 1000, **p), float(g['bn_gain']), float(g['head_gain']))
    x = torch.as_tensor(g['x'])
    n, _, h, w = x.shape
    _, sites = uo.unet_plan(**p)
    stride = int(g['stride'])
    mask_sets = [[g['mask{}_{}'.format(t, s)] for s in range(len(sites))] for t in range(3)]
    m = _model(p, st, dev, conv_winograd4=2)
    kernels = {row['kernel'] for row in m.layer_table(h, w, n)}
    assert {'conv3x3_winograd4<T32x32,N32,K8>', 'conv3x3_winograd4<S2T16x32,N32,K8>', 'conv3x3_winograd4<S8T8x16,N32,K8>',
            'conv3x3_winograd4<S8T12x8,N32,K8>'} <= kernels      # (conv_winograd4=2: the folded 12x8 form too, although two slices do not fill a round)
    xd = x.to(dev)
    refs = []
    for tag, mk in [('eval', None)] + [('mc{}'.format(t), mask_sets[t]) for t in range(3)]:
        ref = uo.unet_forward(st, x, mk, **p).numpy()
        scale = float(np.abs(ref).max())
        if mk is not None:
            assert scale >= 10.0
            refs.append(ref)
        for model in (m,) if mk is None else (m, _model(p, st, dev, conv_winograd4=0), _model(p, st, dev, conv_winograd=0)):
            out = model(xd, mk).cpu().numpy()
            assert _maxdiff(out, ref) < REL * scale, (tag, _maxdiff(out, ref) / scale)
            assert _maxdiff(out.reshape(-1)[::stride], g['logits_{}_strided'.format(tag)]) < REL * scale, tag
            ps, pr = torch.softmax(torch.from_numpy(out), 1).numpy(), torch.softmax(torch.from_numpy(ref), 1).numpy()
            assert _maxdiff(ps, pr) < PROB_TOL, tag
    feats_ref = uo.unet_forward(st, x, mask_sets[0], return_features=True, **p)[1]
    assert float(feats_ref.abs().max()) >= 100.0
    # the three passes through the fused statistics path (float32 and, with every output, float64 statistics)
    multi = torch.stack([torch.softmax(torch.from_numpy(r), 1) for r in refs])
    ref_sum = so.multi_prediction_summary(multi, True, True)
    for flags in ((False, False), (True, True)):
        bc = steps.BatchContext({'images': x.clone()}, 0)
        ctx = steps.TorchTestContext('cuda', m)
        steps.McPredictStep(3, do_mi=flags[0], do_var=flags[1], masks=mask_sets)(bc, None, ctx)
        steps.MultiPredictionSummary(do_mi=flags[0], do_var=flags[1])(bc, None, ctx)
        for key in ('probabilities', 'entropy') + (('mutual_info', 'variance') if flags[0] else ()):
            assert _maxdiff(bc.output[key].cpu().numpy(), ref_sum[key].numpy()) < PROB_TOL, (flags, key)
    sat = float(((multi[:, :, 1] < 0.01) | (multi[:, :, 1] > 0.99)).float().mean())
    assert 0.05 < sat < 0.95          # saturated and unsaturated softmax inputs side by side


# ---------------------------------------------------------------------------------- exact statistics (RCU_MC_EXACT)
def _quantise(x):
    """include/rcu.h RCU_MC_EXACT: an addend enters as (x + 6144) - 6144 in float64 = rounded to the nearest multiple of 2^-40."""
    x = np.asarray(x, dtype=np.float64)
    return (x + 6144.0) - 6144.0


def test_exact_statistics_are_sums_of_quantised_addends_in_any_order(dev):
    """The exact statistics hold, bit for bit, the float64 sums of the quantised float32 probabilities (and their squares and entropies) --
    whatever the order the passes were added in; values far below 2^-40 and exact 0 / 1 included."""
    from rcu_amd import steps
    n, c, h, w, T = 2, 2, 16, 24, 9
    rng = np.random.RandomState(5)
    p1 = rng.rand(T, n, 1, h, w).astype(np.float32)
    p1[0, 0, 0, :4] = [[1e-30, 3e-13, 2.0 ** -41, 1.5 * 2.0 ** -41] * 6]     # below / at half the quantum: 0, 0, ties-to-even, rounds up
    p1[1, 0, 0, 0, :3] = [0.0, 1.0, np.float32(1.0) - np.float32(2.0 ** -24)]
    probs = np.concatenate([1 - p1, p1], axis=2).astype(np.float32)           # [T, n, 2, h, w]
    assert _quantise(2.0 ** -41) == 0.0 and _quantise(1.5 * 2.0 ** -41) == 2.0 ** -40 and _quantise(3e-13) == 0.0
    with np.errstate(divide='ignore', invalid='ignore'):
        ent = -np.sum(np.where(probs > 0, probs * np.log(probs), 0).astype(np.float32), axis=2, dtype=np.float32)
    blobs = []
    for order in (range(T), reversed(range(T)), rng.permutation(T)):
        st = steps.McStatistics(n, c, h, w, dev, do_mi=False, do_var=True, exact=True)
        for t in order:
            st.accumulate(torch.from_numpy(probs[t]).to(dev), is_probabilities=True)
        blobs.append(st.blob.cpu().numpy().copy())
        assert st.blob.dtype == torch.float64 and st.exact
    assert all(np.array_equal(blobs[0].view(np.uint64), b.view(np.uint64)) for b in blobs[1:])
    planes = blobs[0].reshape(2 * c, n, h * w)
    want_p = _quantise(probs).sum(0)                                          # [n, c, h, w]; exact sums: any order
    want_q = _quantise(probs.astype(np.float64) ** 2).sum(0)
    for k in range(c):
        assert np.array_equal(planes[k], want_p[:, k].reshape(n, -1))
        assert np.array_equal(planes[c + k], want_q[:, k].reshape(n, -1))
    # without the squares: planes [sum p_c] [sum H]; the device's logf differs from numpy's in the last ulp, so H is checked for its
    # quantum (a multiple of 2^-40) and its value
    st = steps.McStatistics(n, c, h, w, dev, do_mi=True, do_var=False, exact=True)
    for t in range(T):
        st.accumulate(torch.from_numpy(probs[t]).to(dev), is_probabilities=True)
    planes = st.blob.cpu().numpy().reshape(c + 1, n, h * w)
    assert st.blob.numel() == (c + 1) * n * h * w
    for k in range(c):
        assert np.array_equal(planes[k], want_p[:, k].reshape(n, -1))
    scaled = planes[c] * 2.0 ** 40
    assert np.array_equal(scaled, np.round(scaled))
    assert np.max(np.abs(planes[c] - ent.astype(np.float64).sum(0).reshape(n, -1))) < 1e-5
    out = st.finalize(do_mi=True)
    mean = (want_p / T).astype(np.float32)
    assert np.array_equal(out['probabilities'].cpu().numpy(), mean)
    # the pass limit of the exact form is part of the contract
    from rcu_amd import _lib
    with pytest.raises(_lib.RcuError):
        st.finalize(count=_lib.RCU_MC_EXACT_MAX_PASSES + 1)


def test_exact_statistics_do_not_depend_on_groups_lanes_or_the_materialised_stack(dev):
    """McPredictStep's outputs under a seed: the same bits for every pass grouping, one or two stream lanes, and for the materialised
    [T, N, C, H, W] stack summed by MultiPredictionSummary -- masks are a function of (seed, batch, pass), the sums are exact."""
    from oracle import unet_oracle as uo
    from rcu_amd import steps
    from rcu_amd.model import UNet
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=8, dropout=0.3)
    model = UNet(**params)
    model.load_state_dict(uo.synthetic_state(3, **params))
    model = model.to(dev)
    n, h, w, T = 3, 48, 32, 7
    x = torch.randn(n, 4, h, w, generator=torch.Generator().manual_seed(1)).to(dev)
    # one plan for every grouping: a pass's bits are a property of the plan, and a step sizes its (canonical) plans for n * min(group, T)
    # -- equal for equal group_pixels, whatever the lanes; across different group_pixels the plans are made equal here, up front
    for lane in (0, 1):
        model.reserve(h, w, n * T, lane)
    ctx = steps.TorchTestContext('cuda', model)
    outs = []
    for kwargs in (dict(group_pixels=0, lanes=1), dict(group_pixels=0, lanes=2), dict(group_pixels=4 * n * h * w, lanes=2),
                   dict(group_pixels=3 * n * h * w, lanes=1), dict(materialize=True)):
        bc = steps.BatchContext({'images': x}, 5)
        steps.McPredictStep(T, do_mi=True, do_var=True, seed=20, **kwargs)(bc, None, ctx)
        steps.MultiPredictionSummary(do_mi=True, do_var=True)(bc, None, ctx)
        outs.append(bc.output)
    for o in outs[1:]:
        for key in ('probabilities', 'entropy', 'mutual_info', 'variance', 'ws_probabilities'):
            assert torch.equal(outs[0][key], o[key]), key
    # the loop's own count of the slices it has handed out wins over batch_index x n: the same offset, the same bits
    bc = steps.BatchContext({'images': x}, 0, sample_offset=5 * n)
    steps.McPredictStep(T, seed=20)(bc, None, ctx)
    steps.MultiPredictionSummary()(bc, None, ctx)
    assert torch.equal(bc.output['probabilities'], outs[0]['probabilities'])
    # other slices (another batch index): other masks
    bc = steps.BatchContext({'images': x}, 6)
    steps.McPredictStep(T, seed=20)(bc, None, ctx)
    steps.MultiPredictionSummary()(bc, None, ctx)
    assert not torch.equal(bc.output['probabilities'], outs[0]['probabilities'])
    assert torch.equal(bc.output['ws_probabilities'], outs[0]['ws_probabilities'])
    # the summary may ask for more than the step tracked: the replay draws the same seeded masks
    bc = steps.BatchContext({'images': x}, 5)
    steps.McPredictStep(T, seed=20)(bc, None, ctx)
    steps.MultiPredictionSummary(do_mi=True, do_var=True)(bc, None, ctx)
    for key in ('probabilities', 'entropy', 'mutual_info', 'variance'):
        assert torch.equal(outs[0][key], bc.output[key]), key
    # against the oracle under the same masks (float32 sums there): within the float32 noise of T additions
    from oracle import summary_oracle as so
    step = steps.McPredictStep(T, seed=20)
    steps.set_dropout_mode(model, True)
    masks = [step._seeded_masks(model, x, 5 * n, j) for j in range(1, T + 1)]      # batch 5 of n slices: first global slice index 5 n
    steps.set_dropout_mode(model, False)
    sites = model.dropout_sites()
    mask_sets = [[m.cpu() for m in torch.split(ms, [n * c_ for _, c_ in sites])] for ms in masks]
    mask_sets = [[m.view(n, -1) for m in ms] for ms in mask_sets]
    state = {k: v.cpu() for k, v in model.state_dict().items()}
    ws, multi = so.mc_probabilities(lambda xx, m: uo.unet_forward(state, xx, m, **params), x.cpu(), mask_sets)
    ref = so.multi_prediction_summary(multi, True, True)
    for key, v in ref.items():
        assert _maxdiff(outs[0][key].cpu(), v) < 2e-6, key


def test_uncertain_voxel_sets_of_the_reference_g20(golden, dev):
    """rcu_unc_counts_from_p against fixture g20 (the reference run over every float32 in [0, 1]): on the probe vector -- every value of
    the ragged windows, 40 neighbours on either side of every edge, 0, 0.5, 1 and their neighbours -- the kernel must call exactly the
    reference's voxels uncertain, for all eleven thresholds at once, for subsets, per volume of a batch, and with a mask."""
    from rcu_amd import evaluation as ev
    g = golden('g20_ue_boundaries')
    p = g['probe_bits'].view(np.float32)
    member = g['probe_member'].astype(bool)                     # [11, n]
    thr = tuple(float(t) for t in g['thresholds'])
    assert thr == ev.UE_THRESHOLDS and int(g['values_scanned']) == 0x3F800000 + 1
    rng = np.random.RandomState(7)
    pred = (rng.rand(p.size) > 0.5).astype(np.uint8)
    tgt = (rng.rand(p.size) > 0.5).astype(np.uint8)

    def reference(sel, rows):
        out = np.zeros((len(rows), 8), np.int64)
        for i, r in enumerate(rows):
            u = member[r]
            for k, (a, b) in enumerate(((1, 1), (0, 0), (1, 0), (0, 1))):          # tp, tn, fp, fn: (prediction, target)
                cell = sel & (pred == a) & (tgt == b)
                out[i, k], out[i, 4 + k] = cell.sum(), (cell & u).sum()
        return out

    every = np.ones(p.size, bool)
    assert np.array_equal(ev.uncertainty_counts_from_p(pred, tgt, p, thr)[0], reference(every, range(11)))
    mask = rng.rand(p.size) > 0.4
    assert np.array_equal(ev.uncertainty_counts_from_p(pred, tgt, p, thr, mask=mask)[0], reference(mask, range(11)))
    rows = [0, 4, 9, 10]
    assert np.array_equal(ev.uncertainty_counts_from_p(pred, tgt, p, [thr[r] for r in rows])[0], reference(every, rows))
    assert np.array_equal(ev.uncertainty_counts_from_p(pred, tgt, p, (0.95,))[0], reference(every, [10]))
    # a batch of volumes in one launch: 4 volumes of a quarter each (ragged length: the scalar tail path), and a long one (the vector path)
    q = p.size // 4
    got = ev.uncertainty_counts_from_p(pred[:4 * q], tgt[:4 * q], p[:4 * q], thr, n_volumes=4)
    for v in range(4):
        sel = np.zeros(p.size, bool)
        sel[v * q:(v + 1) * q] = True
        assert np.array_equal(got[v], reference(sel, range(11)))
    reps = 40
    big = ev.uncertainty_counts_from_p(np.tile(pred, reps), np.tile(tgt, reps), np.tile(p, reps), thr)[0]
    assert np.array_equal(big, reps * reference(every, range(11)))
    # values a probability map cannot hold count as not uncertain (the reference rejects them before it gets here)
    odd = np.array([-0.0, -1.0, 2.0, np.nan, np.inf, -np.inf], np.float32)
    c = ev.uncertainty_counts_from_p(np.zeros(6, np.uint8), np.zeros(6, np.uint8), odd, thr)[0]
    assert np.array_equal(c[:, 4:], np.zeros((11, 4), np.int64)) and np.all(c[:, 1] == 6)


def test_confusion_dice_accuracy_against_sklearn_g19(golden, dev):
    """a17 on the GPU path (evaluation.confusion_matrx / dice / accuracy: the counts kernel) against scikit-learn's numbers (fixture g19)."""
    from rcu_amd import evaluation as ev
    g = golden('g19_confusion_third_party')
    for name in g['names']:
        name = str(name)
        pred, tgt = g[name + '::prediction'], g[name + '::target']
        assert list(ev.confusion_matrx(pred, tgt)) == list(g[name + '::counts_tp_tn_fp_fn_n']), name
        assert abs(ev.dice(pred, tgt) - float(g[name + '::f1_zero_division_1'])) < 1e-15, name
        assert abs(ev.accuracy(pred, tgt) - float(g[name + '::accuracy'])) < 1e-15, name
        res = {}
        ev.ComposeEvaluation([ev.DiceNumpy(), ev.ConfusionMatrix()])({'prediction': pred, 'target': tgt}, res)
        assert [res[k] for k in ('tp', 'tn', 'fp', 'fn', 'n')] == list(g[name + '::counts_tp_tn_fp_fn_n'])
