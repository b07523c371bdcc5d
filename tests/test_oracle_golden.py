"""The oracle is only trusted once it reproduces every golden vector the reference produced
(tests/golden/*.npz, made by tests/golden/generate_golden.py).  CPU-only."""
import numpy as np
import pytest
import torch

from conftest import golden_params, golden_state
from oracle import calib_oracle as co
from oracle import c_oracle
from oracle import summary_oracle as so
from oracle import unet_oracle as uo


def _close(a, b, tol=2e-6):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    assert np.max(np.abs(a - b)) <= tol, np.max(np.abs(a - b))


def test_g1_unet_eval(golden):
    g = golden('g1_unet_eval')
    p, st = golden_params(g), golden_state(g)
    for tag in ('a', 'b'):
        y = uo.unet_forward(st, g['x_' + tag], None, **p)
        _close(y.numpy(), g['logits_' + tag])
    g8 = golden('g1_unet_eval_sf8')
    _close(uo.unet_forward(golden_state(g8), g8['x'], None, **golden_params(g8)).numpy(), g8['logits'])


def test_g2_unet_mc_masks(golden):
    g = golden('g2_unet_mc')
    p, st = golden_params(g), golden_state(g)
    plan, sites = uo.unet_plan(**p)
    assert [s[0] for s in sites] == list(g['sites'])
    for t in range(int(g['T'])):
        masks = [g['mask_{}_{}'.format(t, s)] for s in range(len(sites))]
        assert any((m == 0).any() for m in masks)  # dropout really active
        _close(uo.unet_forward(st, g['x'], masks, **p).numpy(), g['logits_{}'.format(t)], 5e-6)


def test_g3_dropout_center_sites(golden):
    g = golden('g3_unet_center')
    p, st = golden_params(g), golden_state(g)
    _, sites = uo.unet_plan(**p)
    assert [s[0] for s in sites] == list(g['sites'])
    assert len(sites) == 9
    masks = [g['mask_{}'.format(s)] for s in range(len(sites))]
    _close(uo.unet_forward(st, g['x'], masks, **p).numpy(), g['logits'], 5e-6)
    _close(uo.unet_forward(st, g['x'], None, **p).numpy(), g['logits_eval'], 5e-6)
    g2 = golden('g3_unet_center2_sites')
    _, sites2 = uo.unet_plan(**golden_params(g2))
    assert [s[0] for s in sites2] == list(g2['sites'])


def test_g4_sigma_head(golden):
    g = golden('g4_unet_sigma')
    p, st = golden_params(g), golden_state(g)
    logits, sigma = uo.unet_forward(st, g['x'], None, **p)
    _close(logits.numpy(), g['logits'])
    _close(sigma.numpy(), g['sigma_raw'])
    out = so.aleatoric_outputs(logits, sigma)
    _close(out['sigma'].numpy(), g['sigma_abs'])
    _close(out['probabilities'].numpy(), g['probabilities'])
    _close(so.aleatoric_outputs(logits, sigma, True)['sigma'].numpy(), g['sigma_exp'], 1e-5)
    pred = out['probabilities'].argmax(1).numpy()
    assert np.array_equal(pred.astype(np.uint8), g['prediction'])
    sel = np.take_along_axis(out['sigma'].numpy(), pred[:, None], axis=1)[:, 0]
    _close(sel, g['sigma_pred'])


def test_g5_isic_shape(golden):
    g = golden('g5_unet_isic')
    _close(uo.unet_forward(golden_state(g), g['x'], None, **golden_params(g)).numpy(), g['logits'])


def test_g6_mc_summary(golden):
    g = golden('g6_mc_summary')
    for case in range(3):
        out = so.multi_prediction_summary(torch.from_numpy(g['multi_{}'.format(case)]), True, True)
        for k in ('probabilities', 'entropy', 'mutual_info', 'variance'):
            _close(out[k].numpy(), g['{}_{}'.format(k, case)], 1e-7)
    _close(so.torch_entropy(torch.from_numpy(g['kat_entropy_in']), dim=1).numpy(), g['kat_entropy_out'], 0)
    # KAT: p1 in {.1,.4,.7} -> mean .4, class-mean unbiased variance .09
    multi = torch.tensor([.1, .4, .7]).view(3, 1, 1, 1, 1)
    multi = torch.cat([1 - multi, multi], 2)
    o = so.multi_prediction_summary(multi, do_var=True)
    assert abs(o['probabilities'][0, 1, 0, 0].item() - 0.4) < 1e-6
    assert abs(o['variance'].item() - 0.09) < 1e-6


def test_g7_mc_step_end_to_end(golden):
    g = golden('g7_mc_step')
    p, st = golden_params(g), golden_state(g)
    _, sites = uo.unet_plan(**p)
    T = int(g['T'])
    mask_sets = [[g['mask_{}_{}'.format(t, s)] for s in range(len(sites))] for t in range(T)]
    x = torch.from_numpy(g['x']).float()
    ws, multi = so.mc_probabilities(lambda xx, m: uo.unet_forward(st, xx, m, **p), x, mask_sets)
    _close(ws.numpy(), g['out::ws_probabilities'], 5e-6)
    _close(multi.numpy(), g['multi_probabilities'], 5e-6)
    out = so.multi_prediction_summary(multi, True, True)
    for k in ('probabilities', 'entropy', 'mutual_info', 'variance'):
        _close(out[k].numpy(), g['out::' + k], 5e-6)
    assert set(g['out_keys']) == {'ws_probabilities', 'probabilities', 'entropy', 'mutual_info', 'variance'}
    assert set(g['default_out_keys']) == {'probabilities', 'entropy'}


def test_g8_ece(golden):
    g = golden('g8_ece')
    thr = co.float32_thresholds(10)
    assert [hex(v) for v in thr.view(np.uint32)] == ['0x3dcccccd', '0x3e4ccccd', '0x3e99999a', '0x3ecccccd',
                                                     '0x3f000001', '0x3f19999a', '0x3f333334', '0x3f4ccccd',
                                                     '0x3f666667']
    # raw bin ids: numpy restatement and the float32-threshold C restatement, incl. the boundary vector
    for tag in ('a', 'b'):
        p = g[tag + '_p'].reshape(-1)
        assert np.array_equal(co.bin_ids(p), g[tag + '_binids'])
        assert np.array_equal(c_oracle.bin_ids(p, thr).astype(np.int64), g[tag + '_binids'])
    p2 = np.stack([1 - g['a_p'], g['a_p']], -1)
    for tag, m in (('masked', g['a_mask']), ('nomask', None)):
        bins = {}
        ece = co.ece_binary(p2, g['a_target'], mask=m, out_bins=bins)
        assert ece == g['a_ece_' + tag]
        for k in ('bins_count', 'bins_avg_confidence', 'bins_positive_fraction', 'bins_non_zero'):
            assert np.array_equal(bins[k], g['a_{}_{}'.format(k, tag)])
        # C oracle: identical histogram -> identical ECE
        cnt, sc, sp = c_oracle.ece_hist(g['a_p'], g['a_target'], m, thr)
        assert co.ece_from_histogram(cnt.astype(np.int64), sc, sp.astype(np.float64)) == g['a_ece_' + tag]
    for tag in ('b', 'c'):
        p = g[tag + '_p']
        bins = {}
        assert co.ece_binary(np.stack([1 - p, p], -1), g[tag + '_target'], out_bins=bins) == g[tag + '_ece']
        assert np.array_equal(bins['bins_count'], g[tag + '_bins_count'])
    assert float(g['c_ece']) == pytest.approx(0.2062500030733645, abs=1e-15)  # KAT, SURVEY 8c
    assert list(g['c_bins_count']) == [2, 2, 1, 1, 2]
    for w in ('log_proportion', 'power_proportion', 'mean_proportion'):
        assert co.ece_binary(p2, g['a_target'], mask=g['a_mask'], bin_weighting=w) == g['d_ece_' + w]
    assert co.ece_binary(p2, g['a_target'], threshold_range=(0.2, 0.9)) == g['d_ece_thresrange']
    assert co.ece_binary(p2, g['a_target'], n_bins=5) == g['e_ece_5bins']
    assert co.ece_binary(p2, g['a_target'], mask=g['a_mask']) == g['d_ece']


def test_g9_uncertainty_counts(golden):
    g = golden('g9_uncertainty')
    p = g['p']
    p2 = co.add_background_probability(p)
    assert np.array_equal(p2, g['probabilities2'])
    unc = co.normalised_entropy(p2)
    assert unc.dtype == np.float64
    assert np.array_equal(unc, g['uncertainty'])
    assert np.array_equal(co.numpy_entropy(p2), g['entropy_nat'])
    pred, tgt = g['prediction'].astype(bool), g['target'].astype(bool)
    for i, thr in enumerate(g['thresholds']):
        c = co.uncertainty_counts(pred, tgt, unc > thr)
        assert list(c) == list(g['counts'][i])
        tp, tn, fp, fn, tpu, tnu, fpu, fnu = c
        d = [co.error_dice(fp, fn, tpu, tnu, fpu, fnu), co.error_recall(fp, fn, fpu, fnu),
             co.error_precision(tpu, tnu, fpu, fnu)]
        assert np.array_equal(np.array(d), g['derived'][i])
    assert list(co.uncertainty_counts(pred, tgt, unc > 0.5, mask=g['mask'])) == list(g['masked_counts_thr05'])
    cc = c_oracle.unc_counts(unc, g['prediction'], g['target'], None, g['thresholds'])
    assert np.array_equal(cc.astype(np.int64), g['counts'])
    cm = c_oracle.unc_counts(unc, g['prediction'], g['target'], g['mask'], [0.5])
    assert list(cm[0].astype(np.int64)) == list(g['masked_counts_thr05'])
    assert list(g['undefined_error_metrics']) == [co.error_dice(0, 0, 0, 0, 0, 0), co.error_recall(0, 0, 0, 0),
                                                  co.error_precision(0, 0, 0, 0)]


def test_g9_correction_metrics_from_counts(golden):
    """The derived correction metrics only need the eight counts: check against a brute-force
    evaluation that really edits the prediction (eval.py:205-226), with the unpinned pymia restatement."""
    g = golden('g9_uncertainty')
    pred, tgt, unc = g['prediction'], g['target'], g['uncertainty']
    for thr in (0.2, 0.5, 0.9):
        u = unc > thr
        r = co.correction_metrics(co.uncertainty_counts(pred.astype(bool), tgt.astype(bool), u))
        for key, val in (('corrected', 0), ('corrected_add', 1)):
            cp = pred.copy()
            cp[u] = val
            tp, tn, fp, fn, n = co.confusion_counts(cp, tgt)
            assert r[key + '_dice'] == pytest.approx(co.dice_from_counts(tp, fp, fn), abs=1e-15)
            assert r[key + '_accuracy'] == pytest.approx(co.accuracy_from_counts(tp, tn, n), abs=1e-15)


def test_g10_preparation(golden):
    g = golden('g10_prep')
    u, pred = g['u'], g['prediction']
    r = co.rescale_uncertainties(u, u.min(), u.max())
    assert np.array_equal(r, g['rescaled_subject'])
    assert np.array_equal(co.rescale_uncertainties(u, 0.1, 3.5), g['rescaled_global'])
    fg = co.uncertainty_to_foreground_probabilities(r, pred)
    assert np.array_equal(fg, g['foreground'])
    assert np.array_equal(co.add_background_probability(fg), g['with_background'])
    for entry, idp, idu in (('probabilities', 'run', 'run'), ('confidence', 'run_rescale', 'run_rescale'),
                            ('sigma', 'run_rescale', 'run_rescale')):
        src = g['prob_prep_in_' + entry]
        out = co.probability_preparation(entry, {entry: src.copy(), 'prediction': pred.copy()})
        assert np.array_equal(out, g['prob_prep_out_' + entry])
        assert str(g['prob_prep_id_' + entry]) == idp
        out_u = co.uncertainty_preparation(entry, {entry: src.copy(), 'prediction': pred.copy()})
        assert np.array_equal(out_u, g['unc_prep_out_' + entry])
        assert str(g['unc_prep_id_' + entry]) == idu
    assert bool(g['raises_range']) and bool(g['raises_shape']) and bool(g['raises_nonbinary'])
    assert bool(g['raises_entropy_classes'])
    with pytest.raises(ValueError):
        co.add_background_probability(np.array([0.5, 1.5]))
    with pytest.raises(ValueError):
        co.uncertainty_to_foreground_probabilities(np.zeros((2, 2)), np.zeros((2, 3)))
    with pytest.raises(ValueError):
        co.uncertainty_to_foreground_probabilities(np.zeros((2, 2)), np.full((2, 2), 2))
    with pytest.raises(ValueError):
        co.normalised_entropy(np.zeros((2, 2, 3)))


def test_g11_fullsize_digest(golden):
    """Full-width model (start_filters=32, 8.6M parameters).  The fixture holds only a strided
    sub-sample of the reference's logits; the weights are rebuilt here by replaying the reference's
    constructor draws (oracle.unet_oracle.reference_init_state), so a match also proves that replay."""
    g = golden('g11_fullsize_digest')
    p = golden_params(g)
    st = uo.reference_init_state(int(g['seed']), bn_seed=int(g['seed']) + 1000, **p)
    assert sorted(st.keys()) == sorted(str(k) for k in g['state_keys'])
    assert int(g['n_state_tensors']) == 143
    n_params = sum(v.numel() for k, v in st.items() if 'running' not in k and 'num_batches' not in k)
    assert n_params == int(g['n_params']) == 8646018
    y = uo.unet_forward(st, g['x'], None, **p).numpy().reshape(-1)
    _close(y[::int(g['stride'])], g['logits_strided'], 2e-5)
    assert abs(float(y.mean()) - float(g['logits_mean'])) < 1e-5
    st2 = uo.synthetic_state(20, **p)
    assert sorted(st2.keys()) == sorted(st.keys())
    assert all(tuple(st2[k].shape) == tuple(st[k].shape) for k in st)


def test_g13_postnet_and_features(golden):
    """auxiliary_feat chain: U-Net features (input of conv_cls) and PostNet logits of the reference."""
    from oracle import unet_oracle as uo
    g = golden('g13_postnet')
    params = golden_params(g)
    params.pop('provide_features')
    unet_state = {k[len('unet::'):]: g[k] for k in g if k.startswith('unet::')}
    logits, feats = uo.unet_forward(unet_state, g['x'], None, return_features=True, **params)
    assert np.max(np.abs(logits.numpy() - g['segm_logits'])) < 1e-5
    assert np.max(np.abs(feats.numpy() - g['features'])) < 1e-5
    for tag, nb_convs in (('post', 3), ('post5', 5)):
        state = {k[len(tag) + 2:]: g[k] for k in g if k.startswith(tag + '::')}
        out = uo.postnet_forward(state, g['features'], nb_convs)
        assert np.max(np.abs(out.numpy() - g['logits' if tag == 'post' else 'logits5'])) < 1e-5


def test_g14_residual_blocks(golden):
    """ConvResidualBlock (unet.py:42-60): second unit without ReLU, 1x1 conv of the block input added, nothing behind the sum."""
    g = golden('g14_unet_residual')
    p, st = golden_params(g), golden_state(g)
    _, sites = uo.unet_plan(**p)
    assert [s[0] for s in sites] == list(g['sites'])
    assert any(k.endswith('.residual.weight') for k in st)
    masks = [g['mask_{}'.format(s)] for s in range(len(sites))]
    assert any((m == 0).any() for m in masks)
    _close(uo.unet_forward(st, g['x'], None, **p).numpy(), g['logits_eval'], 5e-6)
    _close(uo.unet_forward(st, g['x'], masks, **p).numpy(), g['logits_mc'], 5e-6)
    gb = golden('g14_unet_residual_b')      # + dropout_center, sigma head, 3 classes, a size 2^depth does not divide
    logits, sigma = uo.unet_forward(golden_state(gb), gb['x'], None, **golden_params(gb))
    _close(logits.numpy(), gb['logits'], 5e-6)
    _close(sigma.numpy(), gb['sigma'], 5e-6)


def test_g15_centre_pad(golden):
    """Sizes that 2^depth does not divide (unet.py:89, 110-116)."""
    g = golden('g15_unet_centre_pad')
    for tag in ('a', 'b', 'c'):
        p = eval(str(g['params_' + tag]), {'__builtins__': {}}, {'dict': dict})
        st = {k[len('sd_{}::'.format(tag)):]: v for k, v in g.items() if k.startswith('sd_{}::'.format(tag))}
        _close(uo.unet_forward(st, g['x_' + tag], None, **p).numpy(), g['logits_' + tag], 5e-6)


def test_g16_postnet_wide_and_mc(golden):
    g = golden('g16_postnet_wide')
    for tag in ('a', 'b', 'c'):
        c, classes, convs = (int(v) for v in g['shape_' + tag])
        st = {k[len('post_{}::'.format(tag)):]: v for k, v in g.items() if k.startswith('post_{}::'.format(tag))}
        _close(uo.postnet_forward(st, g['features_' + tag], nb_convs=convs).numpy(), g['logits_' + tag], 5e-6)
    st = {k[len('post_d::'):]: v for k, v in g.items() if k.startswith('post_d::')}
    _close(uo.postnet_forward(st, g['features_d'], nb_convs=3).numpy(), g['logits_d_eval'], 5e-6)
    masks = [g['mask_d_{}'.format(s)] for s in range(3)]
    assert any((m == 0).any() for m in masks)
    _close(uo.postnet_forward(st, g['features_d'], nb_convs=3, masks=masks).numpy(), g['logits_d_mc'], 5e-6)


def _tagged(g, tag):
    params = eval(str(g['params_' + tag]), {'__builtins__': {}}, {'dict': dict})
    st = {k[len('sd_{}::'.format(tag)):]: v for k, v in g.items() if k.startswith('sd_{}::'.format(tag))}
    return params, st


def test_g17_no_batchnorm_and_no_dropout(golden):
    """bn=False (unet.py:16-17, 128-164) and dropout=None (unet.py:14-15): the switches no shipped config uses."""
    g = golden('g17_unet_no_bn')
    p, st = _tagged(g, 'a')
    assert p['bn'] is False and not any('.bn.' in k for k in st)
    _, sites = uo.unet_plan(**p)
    assert [s[0] for s in sites] == list(g['sites_a'])
    masks = [g['mask_a_{}'.format(s)] for s in range(len(sites))]
    assert any((m == 0).any() for m in masks)
    _close(uo.unet_forward(st, g['x_a'], None, **p).numpy(), g['logits_a_eval'], 5e-6)
    _close(uo.unet_forward(st, g['x_a'], masks, **p).numpy(), g['logits_a_mc'], 5e-6)
    p, st = _tagged(g, 'b')
    assert p['dropout'] is None and uo.unet_plan(**p)[1] == []
    logits, sigma = uo.unet_forward(st, g['x_b'], None, **p)
    _close(logits.numpy(), g['logits_b'], 5e-6)
    _close(sigma.numpy(), g['sigma_b'], 5e-6)
    p, st = _tagged(g, 'c')
    assert uo.unet_plan(**p)[1] == [] and any('.bn.' in k for k in st)
    _close(uo.unet_forward(st, g['x_c'], None, **p).numpy(), g['logits_c'], 5e-6)


def test_g18_unet_stress(golden):
    """Numerics stress fixture: BatchNorm affines x 2.5, classifier x 0.5 on the full-width U-Net -- interior activations of 1e2..1e3,
    logits of +-20 under Dropout2d(0.3) -- through the reference (an eval pass and three MC passes under its own masks).  The weights are
    rebuilt by replaying the constructor's draws and applying the same scaling rule (oracle.unet_oracle.stress_state); the bound on
    the logits is RELATIVE to their range."""
    g = golden('g18_unet_stress')
    p = golden_params(g)
    st = uo.stress_state(uo.reference_init_state(int(g['seed']), bn_seed=int(g['seed']) + 1000, **p),
                         float(g['bn_gain']), float(g['head_gain']))
    _, sites = uo.unet_plan(**p)
    assert [s[0] for s in sites] == list(g['sites'])
    stride = int(g['stride'])
    y, feats = uo.unet_forward(st, g['x'], None, return_features=True, **p)
    _close(y.numpy().reshape(-1)[::stride], g['logits_eval_strided'], 2e-6 * float(g['logits_eval_absmax']))
    assert abs(float(feats.abs().max()) - float(g['features_eval_absmax'])) < 1e-4 * float(g['features_eval_absmax'])
    for t in range(3):
        masks = [g['mask{}_{}'.format(t, s)] for s in range(len(sites))]
        assert any((m == 0).any() for m in masks)
        y, feats = uo.unet_forward(st, g['x'], masks, return_features=True, **p)
        scale = float(g['logits_mc{}_absmax'.format(t)])
        assert scale >= 10.0 and float(g['features_mc{}_absmax'.format(t)]) >= 100.0       # what the fixture is for
        _close(y.numpy().reshape(-1)[::stride], g['logits_mc{}_strided'.format(t)], 2e-6 * scale)
        assert abs(float(feats.abs().max()) - float(g['features_mc{}_absmax'.format(t)])) < 1e-4 * float(g['features_mc{}_absmax'.format(t)])


def test_confusion_dice_accuracy_against_sklearn_g19(golden):
    """a17: pymia's ConfusionMatrix / DiceCoefficient / Accuracy are absent, so the restatement is pinned against an independent
    third party -- fixture g19 holds scikit-learn's confusion_matrix / f1_score / accuracy_score for random, degenerate and 2D label pairs.
    The one case no third party decides is 0 / 0 (no foreground anywhere): pymia 0.2.1 returns Dice 1 there, which is sklearn's
    zero_division=1 variant."""
    from oracle import calib_oracle as co
    g = golden('g19_confusion_third_party')
    for name in g['names']:
        name = str(name)
        pred, tgt = g[name + '::prediction'], g[name + '::target']
        tp, tn, fp, fn, n = co.confusion_counts(pred, tgt)
        assert [tp, tn, fp, fn, n] == list(g[name + '::counts_tp_tn_fp_fn_n']), name
        dice = co.dice_from_counts(tp, fp, fn)
        assert abs(dice - float(g[name + '::f1_zero_division_1'])) < 1e-15, name
        if 2 * tp + fp + fn > 0:
            assert abs(dice - float(g[name + '::f1_zero_division_0'])) < 1e-15, name
        else:
            assert dice == 1.0 and float(g[name + '::f1_zero_division_0']) == 0.0          # the convention case
        assert abs(co.accuracy_from_counts(tp, tn, n) - float(g[name + '::accuracy'])) < 1e-15, name


def test_mask_oracle_against_the_published_philox_vectors():
    """oracle/mask_oracle.py restates the generator behind rcu_dropout_masks: Philox4x32-10 (Salmon et al., SC'11) pinned by the three
    known-answer vectors of the Random123 distribution (kat_vectors: `philox4x32 10`), then the mask definition's own properties -- values in
    {0, 1 / keep}, the Bernoulli(keep) share, inactive / p = 1 sites, the group layout as the row-wise interleave of the passes' own masks."""
    from oracle import mask_oracle as mo
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for counter, key, want in kats:
        got = mo.philox4x32_10(np.array(counter, dtype=np.uint32), key)
        assert tuple(int(v) for v in got) == want
    channels, keep = [8, 8, 16, 5], [0.7, -1.0, 0.7, 0.0]
    one = mo.pass_mask(12345, 3, channels, keep)
    assert one.dtype == np.float32 and one.size == 3 * sum(channels)
    assert set(np.unique(one[:24])) <= {np.float32(0.0), np.float32(1.0) / np.float32(0.7)}
    assert np.all(one[24:48] == 1.0) and np.all(one[96:] == 0.0)
    big = mo.pass_mask(7, 64, [256], [0.95])
    assert abs(float((big > 0).mean()) - 0.95) < 4 * (0.95 * 0.05 / big.size) ** 0.5
    assert not np.array_equal(mo.pass_mask(7, 4, [32], [0.5]), mo.pass_mask(7 + 2 ** 32, 4, [32], [0.5]))      # both key words count
    # a sample's factors are a function of its GLOBAL index, not of the batch it arrives in: 8 samples at once = 3 + 5 samples, per site
    whole, head, tail = mo.pass_mask(12345, 8, channels, keep), mo.pass_mask(12345, 3, channels, keep), mo.pass_mask(12345, 5, channels, keep, first_sample=3)
    at_w = at_h = at_t = 0
    for c in channels:
        assert np.array_equal(whole[at_w:at_w + 8 * c], np.concatenate([head[at_h:at_h + 3 * c], tail[at_t:at_t + 5 * c]]))
        at_w, at_h, at_t = at_w + 8 * c, at_h + 3 * c, at_t + 5 * c
    assert np.array_equal(head, one)
    far = mo.pass_mask(7, 2, [32], [0.5], first_sample=2 ** 40)          # the counter's second word counts
    assert not np.array_equal(far, mo.pass_mask(7, 2, [32], [0.5], first_sample=2 ** 40 + 2 ** 36)) and not np.array_equal(far, mo.pass_mask(7, 2, [32], [0.5]))
    seeds = [3, 4, 5]
    grouped = mo.group_masks(seeds, 3, channels, keep)
    singles = [mo.pass_mask(s, 3, channels, keep) for s in seeds]
    at, out = 0, 0
    for c in channels:
        for m in singles:
            assert np.array_equal(grouped[out:out + 3 * c], m[at:at + 3 * c])
            out += 3 * c
        at += 3 * c


def test_uncertain_voxel_table_holds_under_the_local_numpy():
    """ADVICE r05.  csrc/rcu_ue_table.inc (fixture g20) records for every float32 p whether the reference's ToEntropy([1 - p, p]) exceeds each of the
    script's 11 thresholds -- as numpy's float32 ``log`` gave it in the container that built the fixture (``numpy_version`` / ``cpu_features`` are
    stored with it).  A host whose numpy rounds ``log`` differently in the last ulp (another SIMD path) would disagree on the few values right at a
    boundary; this re-evaluates every probe value of the fixture with the oracle's numpy restatement on THIS host.  (bench.py repeats the
    check on the GPU box's host against the timed output: ``parity.ue_counts_equal``.)"""
    from conftest import load_golden
    from oracle import calib_oracle as co
    g = load_golden('g20_ue_boundaries')
    p = g['probe_bits'].view(np.float32)
    u = co.normalised_entropy(np.stack([1 - p, p], -1))
    member = np.stack([u > float(t) for t in g['thresholds']])
    disagree = int(np.count_nonzero(member != g['probe_member'].astype(bool)))
    assert disagree == 0, ('numpy {} on this CPU rounds log differently from the build that made the table (numpy {}, {}): {} probe values flip; '
                           'rebuild the table (tests/golden/generate_ue_boundaries.py) or evaluate through rcu_normalised_entropy + rcu_unc_counts'
                           .format(np.__version__, g['numpy_version'], g['cpu_features'], disagree))


def test_g21_real_data_shapes(golden):
    """The shapes of the reference's real data -- a 240 x 240 BraTS slice (levels 240 / 120 / 60 / 30 / 15), ISIC's 24 x 32 and 12 x 16 levels, a ragged
    48 x 80 batch under MC-dropout masks -- through the reference itself (tests/golden/generate_golden.py g21): what pins the oracle, and through
    it the padded levels of the GPU path (tests/test_gpu_padded_levels.py), at shapes whose levels are not whole Winograd tiles."""
    g = golden('g21_unet_real_shapes')
    pa, sta = _tagged(g, 'a')
    pb, stb = _tagged(g, 'b')
    _close(uo.unet_forward(sta, g['x_a'], None, **pa).numpy(), g['logits_a'], 5e-6)
    _close(uo.unet_forward(stb, g['x_b'], None, **pb).numpy(), g['logits_b'], 5e-6)
    _close(uo.unet_forward(sta, g['x_c'], None, **pa).numpy(), g['logits_c'], 5e-6)
    masks = [g['mask_c_{}'.format(s)] for s in range(len(g['sites_c']))]
    assert any((m == 0).any() for m in masks)
    _close(uo.unet_forward(sta, g['x_c'], masks, **pa).numpy(), g['logits_mc_c'], 5e-6)
