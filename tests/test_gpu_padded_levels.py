"""Padded levels (rcu_unet_options.pad_levels, csrc/rcu_api.hip choose_level_extents): shapes whose levels are not whole Winograd tiles --
the reference's real data: BraTS slices are 240 x 240 (scripts/create_brats18_dataset.py:53-72 never crops; levels 240 / 120 / 60 / 30 / 15),
ISIC images 192 x 256 (scripts/prepare_isic_data.py:29-30; bottom level 12 x 16) -- run on the Winograd kernels over level tensors ALLOCATED
with whole-tile extents whose padding holds zeros that no kernel writes.  Checked here: against the oracle, against the direct kernels on the
same input (pad_levels=0), the fused head against the standalone head kernel bit for bit, pass groups, the sigma head, ragged batches, and
that the padding stays zero whatever ran before."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-6
PROB_TOL = 1e-4
WIDE = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a MI355X'
    return torch.device('cuda')


def _model(params, state, dev, **plan_options):
    from rcu_amd.model import UNet
    m = UNet(**params)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in state.items()})
    m.plan_options = dict(plan_options)
    return m.to(dev)


def _maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def _padded_rows(rows):
    return [r for r in rows if (r['grid_height'], r['grid_width']) != ((r['height'] // 2, r['width'] // 2) if r['upsample'] else (r['height'], r['width']))]


# (n, h, w): every level extent is a multiple of 2^depth -- no centre pad -- but not of the Winograd tiles; ragged batches for the kernels whose
# work items span two and eight slices
SHAPES = [(5, 240, 240), (2, 192, 256), (3, 48, 80), (9, 80, 48), (1, 112, 176), (11, 16, 16), (2, 16, 48), (3, 208, 144)]


@pytest.mark.timeout(900)
@pytest.mark.parametrize('shape', SHAPES)
def test_padded_levels_full_width_vs_oracle_and_direct_kernels(dev, shape):
    from oracle import unet_oracle as uo
    n, h, w = shape
    st = uo.synthetic_state(41, **WIDE)
    g = torch.Generator().manual_seed(h * 1000 + w)
    x = torch.randn(n, 4, h, w, generator=g)
    _, sites = uo.unet_plan(**WIDE)
    masks = uo.sample_masks(sites, n, 0.3, g)
    m = _model(WIDE, st, dev)
    rows = m.layer_table(h, w, n)
    if (h, w) != (16, 16):   # (16 x 16: padding the 32-channel level to the 32-pixel-wide tiles would double it -- it stays on the direct kernels)
        assert not any('igemm' in r['kernel'] for r in rows), [(r['name'], r['kernel']) for r in rows if 'igemm' in r['kernel']]
    assert _padded_rows(rows), 'the shape was meant to need padded levels'
    direct = _model(WIDE, st, dev, pad_levels=0)
    assert not _padded_rows(direct.layer_table(h, w, n))
    for mk in (None, masks):
        ref = uo.unet_forward(st, x, mk, **WIDE).numpy()
        out = m(x.to(dev), mk).cpu().numpy()
        assert _maxdiff(out, ref) < LOGIT_TOL
        assert _maxdiff(direct(x.to(dev), mk).cpu().numpy(), out) < LOGIT_TOL
        assert _maxdiff(torch.softmax(torch.from_numpy(out), 1).numpy(), torch.softmax(torch.from_numpy(ref), 1).numpy()) < PROB_TOL


@pytest.mark.timeout(900)
@pytest.mark.parametrize('family', [dict(), dict(conv_winograd4=0), dict(conv_winograd4=2), dict(head_winograd4=0), dict(act_layout=1)])
def test_padded_levels_every_kernel_family(dev, family):
    """The same padded shape through F(4x4,3x3), F(2x2,3x3) only, the folded / forced forms, the F(2x2,3x3) head and channels-last activations."""
    from oracle import unet_oracle as uo
    st = uo.synthetic_state(42, **WIDE)
    g = torch.Generator().manual_seed(77)
    for n, h, w in ((9, 80, 112), (4, 240, 240)):
        x = torch.randn(n, 4, h, w, generator=g)
        _, sites = uo.unet_plan(**WIDE)
        masks = uo.sample_masks(sites, n, 0.3, g)
        ref = uo.unet_forward(st, x, masks, **WIDE).numpy()
        m = _model(WIDE, st, dev, **family)
        rows = m.layer_table(h, w, n)
        assert not any('igemm' in r['kernel'] for r in rows), [(r['name'], r['kernel']) for r in rows]
        assert _maxdiff(m(x.to(dev), masks).cpu().numpy(), ref) < LOGIT_TOL


@pytest.mark.timeout(900)
@pytest.mark.parametrize('head_winograd4', [1, 0])
def test_padded_levels_fused_head_matches_head_kernel_bitwise(dev, head_winograd4):
    """conv_cls.0 with the classifier in its epilogue indexes the caller's logits / statistics over the REAL image while its tiles walk the padded
    level: same bits as the plain epilogue (a compact head tensor) + head_kernel, logits and statistics, single passes and pass groups."""
    from oracle import summary_oracle as so
    from oracle import unet_oracle as uo
    from rcu_amd import steps
    st = uo.synthetic_state(43, **WIDE)
    g = torch.Generator().manual_seed(78)
    n, h, w = 3, 240, 240
    x = torch.randn(n, 4, h, w, generator=g)
    _, sites = uo.unet_plan(**WIDE)
    T = 4
    mask_sets = [uo.sample_masks(sites, n, 0.3, g) for _ in range(T)]
    fused = _model(WIDE, st, dev, head_winograd4=head_winograd4)
    plain = _model(WIDE, st, dev, head_winograd4=head_winograd4)
    plain.set_fuse_head(False)
    assert fused.layer_table(h, w, n * 2)[-1]['head_fusable']
    xd = x.to(dev)
    assert torch.equal(fused(xd, mask_sets[0]), plain(xd, mask_sets[0]))
    results = {}
    for name, m, grouping in (('fused', fused, 1), ('plain', plain, 1), ('fused pairs', fused, 2)):
        m.reserve(h, w, n * 2)
        stats = steps.McStatistics(n, 2, h, w, dev, do_mi=True, do_var=True)
        for t in range(0, T, grouping):
            if grouping == 1:
                m.forward_accumulate(xd, stats, mask_sets[t])
            else:
                m.forward_accumulate(xd, stats, mask_sets[t:t + grouping], passes=grouping)
        results[name] = stats.finalize(True, True)
    for key in results['fused']:
        assert torch.equal(results['fused'][key], results['plain'][key]), key
        assert torch.equal(results['fused'][key], results['fused pairs'][key]), key
    multi = torch.stack([torch.softmax(uo.unet_forward(st, x, mk, **WIDE), 1) for mk in mask_sets])
    ref = so.multi_prediction_summary(multi, True, True)
    for key in ref:
        assert _maxdiff(results['fused'][key].cpu().numpy(), ref[key].numpy()) < PROB_TOL, key


@pytest.mark.timeout(900)
def test_padded_levels_sigma_head_and_narrow_model(dev):
    """The cls + sigma twin unit (64 output channels, standalone head kernel over a compact head tensor) and a narrow model (channels padded to 32)
    on padded levels."""
    from oracle import unet_oracle as uo
    g = torch.Generator().manual_seed(79)
    params = dict(WIDE, sigma_out=True)
    st = uo.synthetic_state(44, **params)
    n, h, w = 2, 240, 240
    x = torch.randn(n, 4, h, w, generator=g)
    m = _model(params, st, dev)
    assert not any('igemm' in r['kernel'] for r in m.layer_table(h, w, n))
    logits, sigma = m(x.to(dev))
    ref_l, ref_s = uo.unet_forward(st, x, None, **params)
    assert _maxdiff(logits.cpu().numpy(), ref_l.numpy()) < LOGIT_TOL
    assert _maxdiff(sigma.cpu().numpy(), ref_s.numpy()) < LOGIT_TOL
    narrow = dict(nb_classes=2, in_channels=3, depth=3, start_filters=8, dropout=0.2)
    stn = uo.synthetic_state(45, **narrow)
    mn = _model(narrow, stn, dev)
    for n, h, w in ((3, 40, 56), (2, 120, 120)):
        x = torch.rand(n, 3, h, w, generator=g)
        _, sites = uo.unet_plan(**narrow)
        masks = uo.sample_masks(sites, n, 0.2, g)
        # (40 x 56: the 5 x 7 bottom level keeps its real extent and the direct kernels -- its up-convolution to 32 channels has one Winograd tile,
        # 16 x 32 -- and that direct up-convolution writes into a PADDED level: the mixed case)
        assert (h, w) == (40, 56) or not any('igemm' in r['kernel'] for r in mn.layer_table(h, w, n))
        assert _maxdiff(mn(x.to(dev), masks).cpu().numpy(), uo.unet_forward(stn, x, masks, **narrow).numpy()) < LOGIT_TOL


@pytest.mark.timeout(900)
def test_padding_stays_zero_whatever_ran_before(dev):
    """The zeros beyond the real image are written once, when the workspace is made.  A plan that has run large batches of large-valued inputs
    gives, on a small batch, the bits a fresh plan gives."""
    from oracle import unet_oracle as uo
    st = uo.synthetic_state(46, **WIDE)
    g = torch.Generator().manual_seed(80)
    h, w = 240, 240
    used = _model(WIDE, st, dev)
    used.reserve(h, w, 9)
    _, sites = uo.unet_plan(**WIDE)
    for n in (9, 4, 7):
        big = (100.0 * torch.randn(n, 4, h, w, generator=g)).to(dev)
        used(big, uo.sample_masks(sites, n, 0.3, g))
    x = torch.randn(3, 4, h, w, generator=g)
    masks = uo.sample_masks(sites, 3, 0.3, g)
    fresh = _model(WIDE, st, dev)
    fresh.reserve(h, w, 9)
    assert torch.equal(used(x.to(dev), masks), fresh(x.to(dev), masks))


@pytest.mark.timeout(900)
def test_feature_tap_on_a_padded_level_and_direct_units_next_to_padded_levels(dev):
    """provide_features (common/model/unet.py:135-136, 178-179): ``features`` is handed out as [voxel][channel] -- on a padded level 0 a compact
    copy of the real pixels is made behind the forward (rcu_unet_features), so the auxiliary-network scripts run the Winograd kernels on the
    reference's 240 x 240 slices too.  And the mixed plans: a level that keeps its real extent on the direct kernels (8 x 8 and 16 x 8 images:
    padding a 32-channel level to the 32-pixel-wide tiles would multiply it) pools INTO a padded level and takes the up-convolution OUT of one."""
    from oracle import unet_oracle as uo
    params = dict(WIDE, provide_features=True)
    st = uo.synthetic_state(47, **WIDE)
    g = torch.Generator().manual_seed(81)
    n, h, w = 2, 240, 240
    x = torch.randn(n, 4, h, w, generator=g)
    _, sites = uo.unet_plan(**WIDE)
    masks = uo.sample_masks(sites, n, 0.3, g)
    m = _model(params, st, dev)
    rows = m.layer_table(h, w, n)
    assert not any('igemm' in r['kernel'] for r in rows) and any((r['grid_height'], r['grid_width']) == (240, 256) for r in rows)
    ref_logits, ref_feat = uo.unet_forward(st, x, masks, return_features=True, **WIDE)
    out = m(x.to(dev), masks)
    assert _maxdiff(out.cpu().numpy(), ref_logits.numpy()) < LOGIT_TOL
    assert tuple(m.features.shape) == tuple(ref_feat.shape)
    assert _maxdiff(m.features.cpu().numpy(), ref_feat.numpy()) < 2e-5 * float(ref_feat.abs().max())
    shallow = dict(WIDE, depth=3)
    st3 = uo.synthetic_state(48, **shallow)
    m3 = _model(shallow, st3, dev)
    _, sites3 = uo.unet_plan(**shallow)
    for n, h, w in ((5, 8, 8), (9, 16, 8)):
        rows = m3.layer_table(h, w, n)
        direct_pooling = [r for r in rows if 'igemm' in r['kernel'] and r['pooled']]
        assert direct_pooling and _padded_rows(rows), [(r['name'], r['kernel']) for r in rows]
        x = torch.randn(n, 4, h, w, generator=g)
        masks = uo.sample_masks(sites3, n, 0.3, g)
        for mk in (None, masks):
            assert _maxdiff(m3(x.to(dev), mk).cpu().numpy(), uo.unet_forward(st3, x, mk, **shallow).numpy()) < LOGIT_TOL


@pytest.mark.timeout(900)
def test_padded_levels_against_the_reference_itself_g21(golden, dev):
    """Fixture g21 (tests/golden/generate_golden.py imports the reference): a 240 x 240 BraTS slice, ISIC's 24 x 32 / 12 x 16 levels, a ragged 48 x 80
    batch under the reference's own captured Dropout2d masks -- logits of the GPU path on padded levels against the reference's."""
    g = golden('g21_unet_real_shapes')

    def tagged(tag):
        params = eval(str(g['params_' + tag]), {'__builtins__': {}}, {'dict': dict})
        return params, {k[len('sd_{}::'.format(tag)):]: v for k, v in g.items() if k.startswith('sd_{}::'.format(tag))}

    pa, sta = tagged('a')
    pb, stb = tagged('b')
    ma, mb = _model(pa, sta, dev), _model(pb, stb, dev)
    for m, tag in ((ma, 'a'), (mb, 'b'), (ma, 'c')):
        x = torch.from_numpy(g['x_' + tag]).to(dev)
        rows = m.layer_table(x.shape[2], x.shape[3], x.shape[0])
        if tag == 'b':      # ISIC's 24 x 32 level as it is, on the full-width tile of four slices (its narrow 12 x 16 level: 0.06 % of the plan -- not worth padding)
            assert any('S4T8x32' in r['kernel'] for r in rows), [r['kernel'] for r in rows]
        else:
            assert _padded_rows(rows), tag
        assert _maxdiff(m(x).cpu().numpy(), g['logits_' + tag]) < LOGIT_TOL, tag
    assert [s[0] for s in ma.dropout_sites()] == list(g['sites_c'])
    masks = [g['mask_c_{}'.format(s)] for s in range(len(g['sites_c']))]
    assert _maxdiff(ma(torch.from_numpy(g['x_c']).to(dev), masks).cpu().numpy(), g['logits_mc_c']) < LOGIT_TOL


@pytest.mark.timeout(1200)
def test_padded_levels_random_shapes_vs_oracle(dev):
    """Fourteen more image sizes (multiples of 16 up to 272, drawn once with a fixed seed) and batch sizes 1..9 through whatever plan the planner
    makes of them -- padded levels, real extents, direct kernels next to either -- against the oracle, eval mode and under masks: the planner
    has a few hundred distinct (level extent, kernel) combinations; the hand-picked shapes above cover the ones the reference's data needs."""
    from oracle import unet_oracle as uo
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=16, dropout=0.1)
    st = uo.synthetic_state(49, **params)
    m = _model(params, st, dev)
    rng = np.random.RandomState(2026)
    g = torch.Generator().manual_seed(82)
    _, sites = uo.unet_plan(**params)
    seen = set()
    for _ in range(14):
        n, h, w = int(rng.randint(1, 10)), 16 * int(rng.randint(1, 18)), 16 * int(rng.randint(1, 18))
        x = torch.randn(n, 4, h, w, generator=g)
        masks = uo.sample_masks(sites, n, 0.3, g)
        rows = m.layer_table(h, w, n)
        seen.update((r['kernel'], (r['grid_height'], r['grid_width']) != ((r['height'] // 2, r['width'] // 2) if r['upsample'] else (r['height'], r['width']))) for r in rows)
        for mk in (None, masks):
            ref = uo.unet_forward(st, x, mk, **params).numpy()
            assert _maxdiff(m(x.to(dev), mk).cpu().numpy(), ref) < LOGIT_TOL, (n, h, w, mk is not None)
    assert sum(1 for _, padded in seen if padded) >= 6 and sum(1 for k, _ in seen if 'igemm' in k) >= 1, sorted(seen)
