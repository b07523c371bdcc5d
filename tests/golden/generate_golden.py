#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference lives at /root/reference and never travels to the
GPU box):

    python tests/golden/generate_golden.py

What is committed is data (inputs + the outputs the reference produced for them), never reference
source.  The reference's absent third-party imports (pymia, SimpleITK, h5py, tensorboardX,
matplotlib) are replaced by inert stub modules: none of the code paths exercised here touches
them.  Fixture names follow SURVEY.md section 8(c), G1..G11.
"""
import hashlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get('RCU_REFERENCE_ROOT', '/root/reference')
OUT_DIR = os.path.dirname(os.path.abspath(__file__))

_STUBBED_ROOTS = ('pymia', 'SimpleITK', 'h5py', 'tensorboardX', 'matplotlib')


class _StubModule(types.ModuleType):
    """Module whose every attribute is a fresh empty class (good enough for `class X(stub.Base)`)."""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        cls = type(name, (object,), {})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in _STUBBED_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        if module.__name__ == 'pymia.config.configuration':
            _fill_pymia_configuration(module)


def _fill_pymia_configuration(module):
    """`from pymia.config.configuration import *` needs real base classes (call sites:
    common/configuration/config.py:1, common/trainloop/config.py:7-190)."""

    class Dictable:
        def to_dict(self, **kwargs):
            return dict(vars(self))

        def from_dict(self, d, **kwargs):
            for k, v in d.items():
                setattr(self, k, v)

    class ConfigurationBase(Dictable):
        pass

    def member_to_dict(obj):
        return {k: (v.to_dict() if isinstance(v, Dictable) else v) for k, v in vars(obj).items()}

    def dict_to_member(obj, d):
        for k, v in d.items():
            cur = getattr(obj, k, None)
            if isinstance(cur, Dictable) and isinstance(v, dict):
                cur.from_dict(v)
            else:
                setattr(obj, k, v)

    module.Dictable = Dictable
    module.ConfigurationBase = ConfigurationBase
    module.member_to_dict = member_to_dict
    module.dict_to_member = dict_to_member
    module.__all__ = ['Dictable', 'ConfigurationBase', 'member_to_dict', 'dict_to_member']


def install_reference():
    sys.meta_path.insert(0, _StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if not hasattr(np, 'bool'):  # reference uses np.bool (eval.py:159)
        np.bool = bool


def save(name, **arrays):
    path = os.path.join(OUT_DIR, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote {} ({:.1f} KiB)'.format(path, os.path.getsize(path) / 1024))


def randomise_bn(model, gen):
    """Non-trivial BN running stats / affine so that eval-mode BN is not the identity."""
    import torch.nn as nn
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.2)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
            # a few negative gammas: the folded scale may be negative
            m.weight.data[::5] *= -1.0


def state_to_npz(model, prefix='sd::'):
    return {prefix + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


def dropout_modules(model):
    import torch.nn as nn
    return [(n, m) for n, m in model.named_modules() if isinstance(m, nn.Dropout2d)]


def capture_masks(model, fn):
    """Run fn() with forward hooks on every Dropout2d; returns per-call list of (site name, mask[N,C])
    where mask holds the multiplicative factor {0, 1/(1-p)} the module applied."""
    records = []
    handles = []

    def make_hook(name, p):
        def hook(mod, inp, out):
            x = inp[0]
            if not mod.training:
                mask = torch.ones(x.shape[:2])
            else:
                kept = (out.abs().sum(dim=(2, 3)) > 0) | (x.abs().sum(dim=(2, 3)) == 0)
                mask = kept.float() / (1.0 - p)
            records.append((name, mask.numpy().copy()))
        return hook

    for name, m in dropout_modules(model):
        handles.append(m.register_forward_hook(make_hook(name, m.p)))
    try:
        result = fn()
    finally:
        for h in handles:
            h.remove()
    return result, records


def make_unet(seed, **params):
    import common.model.unet as ref_unet
    torch.manual_seed(seed)
    model = ref_unet.UNet(**params)
    gen = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        randomise_bn(model, gen)
    model.eval()
    return model


def g1_unet_eval():
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=4, dropout=0.05)
    model = make_unet(1, **params)
    gen = torch.Generator().manual_seed(11)
    xa = torch.randn(2, 4, 32, 32, generator=gen)
    xb = torch.randn(1, 4, 48, 32, generator=gen)
    with torch.no_grad():
        ya = model(xa)
        yb = model(xb)
    save('g1_unet_eval', params=np.array(repr(params)), x_a=xa.numpy(), logits_a=ya.numpy(),
         x_b=xb.numpy(), logits_b=yb.numpy(), **state_to_npz(model))
    # a wider model (start_filters=8 -> channels 8..128) on a 16-divisible non-square input
    params8 = dict(nb_classes=2, in_channels=4, depth=4, start_filters=8, dropout=0.05)
    model8 = make_unet(2, **params8)
    xc = torch.randn(1, 4, 32, 48, generator=gen)
    with torch.no_grad():
        yc = model8(xc)
    save('g1_unet_eval_sf8', params=np.array(repr(params8)), x=xc.numpy(), logits=yc.numpy(),
         **state_to_npz(model8))


def g2_unet_mc():
    import common.utils.torchhelper as ref_th
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=4, dropout=0.3)
    model = make_unet(3, **params)
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(2, 4, 32, 32, generator=gen)
    torch.manual_seed(20)
    ref_th.set_dropout_mode(model, True)
    T = 3
    arrays = {}
    site_names = [n for n, _ in dropout_modules(model)]
    with torch.no_grad():
        for t in range(T):
            y, recs = capture_masks(model, lambda: model(x))
            assert [r[0] for r in recs] == site_names
            arrays['logits_{}'.format(t)] = y.numpy()
            for s, (_, mask) in enumerate(recs):
                arrays['mask_{}_{}'.format(t, s)] = mask
    ref_th.set_dropout_mode(model, False)
    save('g2_unet_mc', params=np.array(repr(params)), x=x.numpy(), T=np.array(T),
         sites=np.array(site_names), **arrays, **state_to_npz(model))


def g3_unet_center():
    import common.utils.torchhelper as ref_th
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=4, dropout=0.5, dropout_center=4)
    model = make_unet(4, **params)
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(2, 4, 32, 32, generator=gen)
    site_names = [n for n, _ in dropout_modules(model)]
    torch.manual_seed(21)
    ref_th.set_dropout_mode(model, True)
    with torch.no_grad():
        y, recs = capture_masks(model, lambda: model(x))
    ref_th.set_dropout_mode(model, False)
    with torch.no_grad():
        y_eval = model(x)
    arrays = {'mask_{}'.format(s): m for s, (_, m) in enumerate(recs)}
    save('g3_unet_center', params=np.array(repr(params)), x=x.numpy(), logits=y.numpy(),
         logits_eval=y_eval.numpy(), sites=np.array(site_names), **arrays, **state_to_npz(model))
    # second placement: dropout_center=2 (only the two deepest levels carry dropout)
    params2 = dict(nb_classes=2, in_channels=4, depth=4, start_filters=4, dropout=0.5, dropout_center=2)
    model2 = make_unet(5, **params2)
    site_names2 = [n for n, _ in dropout_modules(model2)]
    save('g3_unet_center2_sites', params=np.array(repr(params2)), sites=np.array(site_names2))


def g4_unet_sigma():
    import common.utils.labelhelper as ref_lh
    import torch.nn.functional as F
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=4, dropout=0.05, sigma_out=True)
    model = make_unet(6, **params)
    gen = torch.Generator().manual_seed(14)
    x = torch.randn(2, 4, 32, 32, generator=gen)
    with torch.no_grad():
        logits, sigma = model(x)
        # AleatoricPredictStep semantics (bin-dl/brats_test_aleatoric.py:63-73)
        sigma_abs = sigma.abs()
        sigma_exp = sigma.exp()
        probs = F.softmax(logits, 1)
    # writer-side selection (bin-dl/brats_test_aleatoric.py:95-97) on channel-last numpy arrays
    probs_np = probs.permute(0, 2, 3, 1).numpy()
    sig_np = sigma_abs.permute(0, 2, 3, 1).numpy()
    prediction = np.argmax(probs_np, axis=-1)
    sigma_pred = sig_np[ref_lh.to_one_hot(prediction, 2).astype(bool)].reshape(prediction.shape)
    save('g4_unet_sigma', params=np.array(repr(params)), x=x.numpy(), logits=logits.numpy(),
         sigma_raw=sigma.numpy(), sigma_abs=sigma_abs.numpy(), sigma_exp=sigma_exp.numpy(),
         probabilities=probs.numpy(), prediction=prediction.astype(np.uint8), sigma_pred=sigma_pred,
         **state_to_npz(model))


def g5_unet_isic():
    params = dict(nb_classes=2, in_channels=3, depth=4, start_filters=4, dropout=0.05)
    model = make_unet(7, **params)
    gen = torch.Generator().manual_seed(15)
    x = torch.rand(1, 3, 32, 48, generator=gen)
    with torch.no_grad():
        y = model(x)
    save('g5_unet_isic', params=np.array(repr(params)), x=x.numpy(), logits=y.numpy(), **state_to_npz(model))


def _summary_via_reference(multi, do_mi, do_var):
    import common.trainloop.context as ref_ctx
    import rechun.dl.customsteps as ref_steps
    bc = ref_ctx.BatchContext({}, 0)
    bc.output['multi_probabilities'] = multi.clone()
    ref_steps.MultiPredictionSummary(do_mi=do_mi, do_var=do_var)(bc, None, None)
    return {k: v.numpy() for k, v in bc.output.items()}


def g6_mc_summary():
    gen = torch.Generator().manual_seed(16)
    arrays = {}
    for case, (T, N, C, H, W) in enumerate([(3, 2, 2, 8, 8), (5, 2, 2, 8, 8), (4, 1, 3, 4, 8)]):
        logits = torch.randn(T, N, C, H, W, generator=gen) * 3
        multi = torch.softmax(logits, 2)
        if C == 2:
            # exact 0/1 probabilities (where(p>0) branch) and an all-equal voxel (zero variance)
            multi[:, 0, 0, 0, 0] = 0.0
            multi[:, 0, 1, 0, 0] = 1.0
            multi[:, 0, 0, 0, 1] = 0.25
            multi[:, 0, 1, 0, 1] = 0.75
            multi[0, 0, 0, 0, 2] = 0.0
            multi[0, 0, 1, 0, 2] = 1.0
        out = _summary_via_reference(multi, True, True)
        arrays['multi_{}'.format(case)] = multi.numpy()
        for k, v in out.items():
            arrays['{}_{}'.format(k, case)] = v
    # KATs from SURVEY 8(c)
    import common.utils.torchhelper as ref_th
    kat = ref_th.entropy(torch.tensor([[.5, .5], [1., 0.], [.9, .1]]), dim=1).numpy()
    arrays['kat_entropy_in'] = np.array([[.5, .5], [1., 0.], [.9, .1]], dtype=np.float32)
    arrays['kat_entropy_out'] = kat
    save('g6_mc_summary', **arrays)


def g7_mc_step_end2end():
    import common.trainloop.context as ref_ctx
    import rechun.dl.customsteps as ref_steps
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=4, dropout=0.3)
    model = make_unet(8, **params)
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(2, 4, 32, 32, generator=gen).double()  # step casts with .float()
    context = ref_ctx.TorchTestContext('cpu')
    context.model = model
    torch.set_grad_enabled(False)
    T = 4
    bc = ref_ctx.BatchContext({'images': x.clone()}, 0)
    torch.manual_seed(20)
    site_names = [n for n, _ in dropout_modules(model)]
    _, recs = capture_masks(model, lambda: ref_steps.McPredictStep(T)(bc, None, context))
    assert len(recs) == (T + 1) * len(site_names)
    multi = bc.output['multi_probabilities'].numpy().copy()
    ref_steps.MultiPredictionSummary(do_mi=True, do_var=True)(bc, None, context)
    arrays = {'out::' + k: v.numpy() for k, v in bc.output.items()}
    arrays['out_keys'] = np.array(list(bc.output.keys()))
    S = len(site_names)
    for t in range(T):  # records [0:S] belong to the weight-scaling pass (all ones)
        for s in range(S):
            arrays['mask_{}_{}'.format(t, s)] = recs[(t + 1) * S + s][1]
    # default flags: only mean + entropy
    bc2 = ref_ctx.BatchContext({}, 0)
    bc2.output['multi_probabilities'] = torch.from_numpy(multi)
    ref_steps.MultiPredictionSummary()(bc2, None, context)
    arrays['default_out_keys'] = np.array(list(bc2.output.keys()))
    # wrong context type: the step means to raise ValueError (customsteps.py:17-18); record what the
    # reference really raises (its message helper fails on the tuple argument, common/utils/messages.py:5)
    try:
        ref_steps.McPredictStep(1)(ref_ctx.BatchContext({'images': x}, 0), None, object())
        raised = 'none'
    except Exception as e:  # noqa: BLE001 - recording the type is the point
        raised = type(e).__name__
    arrays['wrong_context_exception'] = np.array(raised)
    save('g7_mc_step', params=np.array(repr(params)), x=x.numpy(), T=np.array(T), sites=np.array(site_names),
         multi_probabilities=multi, **arrays, **state_to_npz(model))


def g8_ece():
    import common.evalutation.numpyfunctions as ref_np
    import common.evalutation.eval as ref_ev
    rng = np.random.RandomState(18)
    arrays = {}
    # (a) random volume with mask
    p = rng.rand(6, 16, 16).astype(np.float32)
    p[0, 0, :4] = [0.0, 1.0, 0.5, 0.1]
    target = (rng.rand(6, 16, 16) < p * 0.8 + 0.1).astype(np.uint8)
    mask = rng.rand(6, 16, 16) > 0.3
    probs2 = np.stack([1 - p, p], axis=-1)
    for tag, m in (('masked', mask), ('nomask', None)):
        bins = {}
        ece = ref_np.ece_binary(probs2, target, mask=m, out_bins=bins)
        arrays['a_ece_' + tag] = np.array(ece)
        for k, v in bins.items():
            arrays['a_{}_{}'.format(k, tag)] = np.asarray(v)
    arrays.update(a_p=p, a_target=target, a_mask=mask)
    # raw bin ids exactly as _binary_calibration derives them (numpyfunctions.py:53-54)
    edges = np.linspace(0., 1. + 1e-8, 11)
    arrays['a_binids'] = (np.digitize(p.flatten(), edges) - 1).astype(np.int64)
    # (b) boundary vector: every float32 threshold, its predecessor and successor, 0 and 1
    cands = [np.float32(0.0), np.float32(1.0), np.nextafter(np.float32(1.0), np.float32(0.0))]
    for k in range(1, 10):
        e = edges[k]
        c = np.float32(e)
        for _ in range(3):
            c = np.nextafter(c, np.float32(0.0))
        for _ in range(7):
            cands.append(c)
            c = np.nextafter(c, np.float32(2.0))
    b = np.array(cands, dtype=np.float32)
    arrays['b_p'] = b
    arrays['b_binids'] = (np.digitize(b, edges) - 1).astype(np.int64)
    bt = (np.arange(b.size) % 3 == 0).astype(np.uint8)
    bins = {}
    arrays['b_ece'] = np.array(ref_np.ece_binary(np.stack([1 - b, b], -1), bt, out_bins=bins))
    arrays['b_target'] = bt
    for k, v in bins.items():
        arrays['b_' + k] = np.asarray(v)
    # (c) the known-answer test quoted in SURVEY 8(c)
    kp = np.array([.05, .15, .15, .95, .65, .5, 1, 0], dtype=np.float32)
    kt = np.array([0, 0, 1, 1, 1, 0, 1, 0], dtype=np.uint8)
    bins = {}
    arrays['c_ece'] = np.array(ref_np.ece_binary(np.stack([1 - kp, kp], -1), kt, out_bins=bins))
    arrays.update(c_p=kp, c_target=kt)
    for k, v in bins.items():
        arrays['c_' + k] = np.asarray(v)
    # (d) EvaluationStrategy wrapper (eval.py:118-142) incl. return_bins and other bin weightings
    res = {}
    ref_ev.EceBinaryNumpy(with_mask=True, return_bins=True)({'target': target, 'probabilities': probs2,
                                                             'mask': mask}, res)
    arrays['d_keys'] = np.array(sorted(res.keys()))
    arrays['d_ece'] = np.array(res['ece'])
    for w in ('log_proportion', 'power_proportion', 'mean_proportion'):
        arrays['d_ece_' + w] = np.array(ref_np.ece_binary(probs2, target, mask=mask, bin_weighting=w))
    arrays['d_ece_thresrange'] = np.array(ref_np.ece_binary(probs2, target, threshold_range=(0.2, 0.9)))
    # (e) 5-bin variant
    arrays['e_ece_5bins'] = np.array(ref_np.ece_binary(probs2, target, n_bins=5))
    # (f) degenerate: all-masked-out -> empty input
    save('g8_ece', **arrays)


def g9_uncertainty():
    import common.evalutation.numpyfunctions as ref_np
    import rechun.eval.analysis as ref_an
    rng = np.random.RandomState(19)
    p = rng.rand(4, 12, 12).astype(np.float32)
    p[0, 0, :6] = [0.0, 1.0, 0.5, 1e-7, 0.999999, 0.25]
    target = (rng.rand(4, 12, 12) < 0.3).astype(np.uint8)
    prediction = (p > 0.5).astype(np.uint8)
    to_eval = {'probabilities': p.copy(), 'prediction': prediction, 'target': target}
    to_eval = ref_an.AddBackgroundProbabilities()(to_eval)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        to_eval = ref_an.ToEntropy()(to_eval)
    unc = to_eval['uncertainty']
    assert unc.dtype == np.float64
    thresholds = [0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95]
    counts = np.zeros((len(thresholds), 8), dtype=np.int64)
    derived = np.zeros((len(thresholds), 3), dtype=np.float64)
    for i, thr in enumerate(thresholds):
        c = ref_np.uncertainty(prediction.astype(bool), target.astype(bool), unc > thr)
        counts[i] = c
        tp, tn, fp, fn, tpu, tnu, fpu, fnu = c
        derived[i] = [ref_np.error_dice(fp, fn, tpu, tnu, fpu, fnu), ref_np.error_recall(fp, fn, fpu, fnu),
                      ref_np.error_precision(tpu, tnu, fpu, fnu)]
    # masked variant (numpyfunctions.py:87-90)
    mask = rng.rand(4, 12, 12) > 0.5
    masked = np.array(ref_np.uncertainty(prediction.astype(bool), target.astype(bool), unc > 0.5, mask=mask))
    # numpy entropy (numpyfunctions.py:166-168)
    ent = ref_np.entropy(np.stack([1 - p, p], -1))
    save('g9_uncertainty', p=p, target=target, prediction=prediction, probabilities2=to_eval['probabilities'],
         uncertainty=unc, thresholds=np.array(thresholds), counts=counts, derived=derived, mask=mask,
         masked_counts_thr05=masked, entropy_nat=ent,
         undefined_error_metrics=np.array([ref_np.error_dice(0, 0, 0, 0, 0, 0), ref_np.error_recall(0, 0, 0, 0),
                                           ref_np.error_precision(0, 0, 0, 0)]))


def g10_prep():
    import rechun.eval.helper as ref_h
    import rechun.eval.analysis as ref_an
    import rechun.eval.evaldata as ref_ed
    rng = np.random.RandomState(20)
    u = (rng.rand(3, 8, 8).astype(np.float32) * 3.0 + 0.2)
    pred = (rng.rand(3, 8, 8) > 0.6).astype(np.uint8)
    arrays = dict(u=u, prediction=pred)
    mn, mx = float(u.min()), float(u.max())
    r = ref_h.rescale_uncertainties(u, u.min(), u.max())
    arrays['rescaled_subject'] = r
    arrays['rescaled_global'] = ref_h.rescale_uncertainties(u, 0.1, 3.5)
    fg = ref_h.uncertainty_to_foreground_probabilities(r, pred)
    arrays['foreground'] = fg
    arrays['with_background'] = ref_h.add_background_probability(fg)
    # composed recipes per confidence entry (analysis.py:218-274)
    for entry in ('probabilities', 'confidence', 'sigma'):
        ed = types.SimpleNamespace(confidence_entry=entry, id_='run')
        prep, id_ = ref_an.get_probability_preparation(ed, rescale_confidence='subject', rescale_sigma='subject')
        src = rng.rand(3, 8, 8).astype(np.float32) if entry == 'probabilities' else u.copy()
        to_eval = {entry: src.copy(), 'prediction': pred.copy()}
        out = prep(to_eval)
        arrays['prob_prep_in_' + entry] = src
        arrays['prob_prep_out_' + entry] = out['probabilities']
        arrays['prob_prep_id_' + entry] = np.array(id_)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            prep_u, id_u = ref_an.get_uncertainty_preparation(ed, rescale_confidence='subject',
                                                              rescale_sigma='subject')
            out_u = prep_u({entry: src.copy(), 'prediction': pred.copy()})
        arrays['unc_prep_out_' + entry] = out_u['uncertainty']
        arrays['unc_prep_id_' + entry] = np.array(id_u)
    # guards
    def raises(fn):
        try:
            fn()
            return False
        except ValueError:
            return True
    arrays['raises_range'] = np.array(raises(lambda: ref_h.add_background_probability(np.array([0.5, 1.5]))))
    arrays['raises_shape'] = np.array(raises(
        lambda: ref_h.uncertainty_to_foreground_probabilities(np.zeros((2, 2)), np.zeros((2, 3)))))
    arrays['raises_nonbinary'] = np.array(raises(
        lambda: ref_h.uncertainty_to_foreground_probabilities(np.zeros((2, 2)), np.full((2, 2), 2))))
    arrays['raises_entropy_classes'] = np.array(raises(
        lambda: ref_an.ToEntropy()({'probabilities': np.zeros((2, 2, 3))})))
    arrays['minmax'] = np.array([mn, mx])
    save('g10_prep', **arrays)


def g11_fullsize_digest():
    """Full-width model (start_filters=32, 8.6M parameters) on one small slice: only a strided
    sub-sample of the reference logits is committed; the weights are regenerated from the seed."""
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05)
    model = make_unet(20, **params)
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(2, 4, 48, 32, generator=gen)
    with torch.no_grad():
        y = model(x)
    flat = y.numpy().reshape(-1)
    n_params = sum(p.numel() for p in model.parameters())
    keys = list(model.state_dict().keys())
    sha = hashlib.sha256(y.numpy().tobytes()).hexdigest()
    save('g11_fullsize_digest', params=np.array(repr(params)), seed=np.array(20), x=x.numpy(),
         logits_strided=flat[::37].copy(), stride=np.array(37), logits_mean=np.array(flat.mean()),
         logits_absmax=np.array(np.abs(flat).max()), n_params=np.array(n_params), n_state_tensors=np.array(len(keys)),
         state_keys=np.array(keys), sha256_here=np.array(sha))


def g12_eval_csv():
    """CSV text the reference's eval hooks (rechun/eval/hook.py:28-116) write for fixed result dicts."""
    import tempfile
    import rechun.eval.hook as ref_hook
    import common.evalutation.numpyfunctions as ref_np
    rng = np.random.RandomState(22)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        # WriteCsvHook with explicit entries (EceAction) and with entries=None + list-valued results
        h1 = ref_hook.WriteCsvHook(os.path.join(tmp, 'a.csv'), entries=('ece', 'dice', 'tp', 'tn', 'fp', 'fn', 'n'))
        h2 = ref_hook.WriteCsvHook(os.path.join(tmp, 'b.csv'), None)
        h3 = ref_hook.WriteBinsCsvHook(os.path.join(tmp, 'c.csv'))
        h4 = ref_hook.WriteSummaryCsvHook(os.path.join(tmp, 'd.csv'), confidence_entry='sigma')
        hist = {}
        subjects = ['Brats18_A_1', 'Brats18_B_1', 'Brats18_C_1']
        inputs = []
        for i, sub in enumerate(subjects):
            p = rng.rand(4, 8, 8).astype(np.float32)
            if i == 1:
                p = p * 0.4          # leaves the upper bins empty
            t = (rng.rand(4, 8, 8) < p).astype(np.uint8)
            bins = {}
            ece = ref_np.ece_binary(np.stack([1 - p, p], -1), t, out_bins=bins)
            res1 = {'ece': ece, 'dice': 0.5 + 0.1 * i, 'tp': 10 + i, 'tn': 200 - i, 'fp': 3 * i, 'fn': 7, 'n': 256,
                    'extra': 'ignored'}
            res2 = {'tpu': 3 + i, 'values': np.array([0.25, 0.5 * i]), 'flag': bool(i % 2), 'twelve': list(range(12))}
            res3 = dict(bins)
            res3['ece'] = ece
            res3['dice'] = 0.25 * i
            h1.on_subject(dict(res1), sub, 'baseline_mc')
            h2.on_subject(dict(res2), sub, 'baseline_mc')
            h3.on_subject(res3, sub, 'baseline_mc')
            for k, v in (('min', float(p.min())), ('max', float(p.max()))):
                hist.setdefault(k, []).append(v)
            inputs.append((p, t))
        for h in (h1, h2, h3):
            h.on_run_end({}, 'baseline_mc')
        h4.on_run_end(hist, 'aleatoric')
        for name in 'abcd':
            with open(os.path.join(tmp, name + '.csv'), newline='') as f:
                out['csv_' + name] = np.array(f.read())
    for i, (p, t) in enumerate(inputs):
        out['p_{}'.format(i)] = p
        out['t_{}'.format(i)] = t
    out['subjects'] = np.array(subjects)
    out['hist_min'] = np.array(hist['min'])
    out['hist_max'] = np.array(hist['max'])
    import rechun.directories as ref_dirs
    out['names'] = np.array([ref_dirs.ECE_FOREGROUND_NAME, ref_dirs.ECE_NAME, ref_dirs.CALIB_NAME, ref_dirs.UNCERTAINTY_NAME,
                             ref_dirs.MINMAX_NAME, ref_dirs.CALIBRATION_PLACEHOLDER, ref_dirs.UNCERTAINTY_PLACEHOLDER,
                             ref_dirs.ECE_PLACEHOLDER, ref_dirs.MINMAX_PLACEHOLDER])
    save('g12_eval_csv', **out)


def g13_postnet():
    """auxiliary_feat: reference UNet(provide_features=True) -> features -> reference PostNet (postnet.py:6-18),
    the way bin-dl/brats_test_auxiliary_feat.py:67-77 chains them; a second PostNet with non-default depth/classes."""
    import common.model.postnet as ref_postnet
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=4, dropout=0.05, provide_features=True)
    model = make_unet(13, **params)
    gen = torch.Generator().manual_seed(131)
    x = torch.randn(2, 4, 32, 48, generator=gen)
    torch.manual_seed(14)
    post = ref_postnet.PostNet(4, 2)
    randomise_bn(post, gen)
    post.eval()
    post5 = ref_postnet.PostNet(4, 3, nb_convs=5)
    randomise_bn(post5, gen)
    post5.eval()
    with torch.no_grad():
        segm_logits = model(x)
        feats = model.features
        logits = post(feats)
        logits5 = post5(feats)
    arrays = dict(params=np.array(repr(params)), x=x.numpy(), segm_logits=segm_logits.numpy(), features=feats.numpy(),
                  logits=logits.numpy(), logits5=logits5.numpy())
    arrays.update(state_to_npz(model, 'unet::'))
    arrays.update(state_to_npz(post, 'post::'))
    arrays.update(state_to_npz(post5, 'post5::'))
    save('g13_postnet', **arrays)


def g14_unet_residual():
    """ConvResidualBlock (unet.py:42-60, ``residual=True``): eval pass and one MC pass with the masks the reference drew; a second
    model with dropout_center and a size that 2^depth does not divide (residual blocks + centre pad together)."""
    import common.utils.torchhelper as ref_th
    params = dict(nb_classes=2, in_channels=4, depth=3, start_filters=4, dropout=0.3, residual=True)
    model = make_unet(14, **params)
    gen = torch.Generator().manual_seed(141)
    x = torch.randn(2, 4, 32, 24, generator=gen)
    site_names = [n for n, _ in dropout_modules(model)]
    arrays = {}
    with torch.no_grad():
        arrays['logits_eval'] = model(x).numpy()
        torch.manual_seed(22)
        ref_th.set_dropout_mode(model, True)
        y, recs = capture_masks(model, lambda: model(x))
        ref_th.set_dropout_mode(model, False)
    arrays['logits_mc'] = y.numpy()
    for s_, (_, mask) in enumerate(recs):
        arrays['mask_{}'.format(s_)] = mask
    save('g14_unet_residual', params=np.array(repr(params)), x=x.numpy(), sites=np.array(site_names), **arrays,
         **state_to_npz(model))
    params2 = dict(nb_classes=3, in_channels=3, depth=2, start_filters=8, dropout=0.2, dropout_center=2, residual=True, sigma_out=True)
    model2 = make_unet(15, **params2)
    x2 = torch.rand(1, 3, 22, 30, generator=gen)
    with torch.no_grad():
        logits2, sigma2 = model2(x2)
    save('g14_unet_residual_b', params=np.array(repr(params2)), x=x2.numpy(), logits=logits2.numpy(), sigma=sigma2.numpy(),
         **state_to_npz(model2))


def g15_unet_centre_pad():
    """Sizes 2^depth does not divide: the up-convolution is zero-padded around its centre to the skip tensor's size
    (unet.py:110-116; the max-pool floors, unet.py:89)."""
    out = {}
    for tag, (cin, shape) in {'a': (4, (2, 40, 36)), 'b': (3, (1, 50, 30)), 'c': (4, (1, 37, 19))}.items():
        params = dict(nb_classes=2, in_channels=cin, depth=4, start_filters=4, dropout=0.05)
        model = make_unet(16 + ord(tag), **params)
        gen = torch.Generator().manual_seed(150 + ord(tag))
        x = torch.randn(shape[0], cin, shape[1], shape[2], generator=gen)
        with torch.no_grad():
            y = model(x)
        assert y.shape[-2:] == x.shape[-2:]
        out.update({'params_' + tag: np.array(repr(params)), 'x_' + tag: x.numpy(), 'logits_' + tag: y.numpy()})
        out.update(state_to_npz(model, 'sd_{}::'.format(tag)))
    save('g15_unet_centre_pad', **out)


def g21_unet_real_shapes():
    """The shapes the reference's real data has, none of whose deeper levels is a whole number of the build's Winograd tiles (round 6: padded
    levels): a BraTS slice is 240 x 240 (scripts/create_brats18_dataset.py:53-72 never crops; levels 240 / 120 / 60 / 30 / 15), an ISIC image
    192 x 256 (scripts/prepare_isic_data.py:29-30; bottom level 12 x 16 -- here at depth 3 from 96 x 128, the same levels 24 x 32 and 12 x 16),
    and a ragged 48 x 80 batch with MC-dropout masks."""
    ref_th = __import__('common.utils.torchhelper', fromlist=['x'])
    out = {}
    params_a = dict(nb_classes=2, in_channels=4, depth=4, start_filters=4, dropout=0.3)
    params_b = dict(nb_classes=2, in_channels=3, depth=3, start_filters=4, dropout=0.05)
    model_a, model_b = make_unet(137, **params_a), make_unet(138, **params_b)
    out.update({'params_a': np.array(repr(params_a)), 'params_b': np.array(repr(params_b))})
    out.update(state_to_npz(model_a, 'sd_a::'))
    out.update(state_to_npz(model_b, 'sd_b::'))
    gen = torch.Generator().manual_seed(211)
    for tag, model, shape in (('a', model_a, (1, 4, 240, 240)), ('b', model_b, (2, 3, 96, 128)), ('c', model_a, (3, 4, 48, 80))):
        x = torch.randn(*shape, generator=gen)
        with torch.no_grad():
            y = model(x)
        out.update({'x_' + tag: x.numpy(), 'logits_' + tag: y.numpy()})
        if tag == 'c':       # the ragged batch under MC dropout (model a's weights), masks captured
            with torch.no_grad():
                torch.manual_seed(23)
                ref_th.set_dropout_mode(model, True)
                y_mc, recs = capture_masks(model, lambda: model(x))
                ref_th.set_dropout_mode(model, False)
            out['logits_mc_c'] = y_mc.numpy()
            out['sites_c'] = np.array([n for n, _ in recs])
            for s_, (_, mask) in enumerate(recs):
                out['mask_c_{}'.format(s_)] = mask
    save('g21_unet_real_shapes', **out)


def g16_postnet_wide():
    """PostNet (postnet.py:6-18) on more than 32 feature channels (a U-Net with start_filters 48 / 64) and with MC-dropout inside
    (Conv2dBnRelu with a dropout rate, masks captured)."""
    import common.model.postnet as ref_postnet
    import common.utils.torchhelper as ref_th
    gen = torch.Generator().manual_seed(161)
    arrays = {}
    for tag, (c, classes, convs) in {'a': (64, 2, 3), 'b': (48, 3, 2), 'c': (40, 2, 4)}.items():
        torch.manual_seed(30 + ord(tag))
        post = ref_postnet.PostNet(c, classes, nb_convs=convs)
        randomise_bn(post, gen)
        post.eval()
        f = torch.randn(2, c, 12, 20, generator=gen)
        with torch.no_grad():
            arrays['features_' + tag] = f.numpy()
            arrays['logits_' + tag] = post(f).numpy()
        arrays['shape_' + tag] = np.array([c, classes, convs])
        arrays.update(state_to_npz(post, 'post_{}::'.format(tag)))
    torch.manual_seed(40)
    post = ref_postnet.PostNet(32, 2, nb_convs=3, dropout=0.3)
    randomise_bn(post, gen)
    post.eval()
    f = torch.randn(3, 32, 8, 16, generator=gen)
    with torch.no_grad():
        arrays['features_d'] = f.numpy()
        arrays['logits_d_eval'] = post(f).numpy()
        torch.manual_seed(41)
        ref_th.set_dropout_mode(post, True)
        y, recs = capture_masks(post, lambda: post(f))
        ref_th.set_dropout_mode(post, False)
    arrays['logits_d_mc'] = y.numpy()
    for s_, (_, mask) in enumerate(recs):
        arrays['mask_d_{}'.format(s_)] = mask
    arrays.update(state_to_npz(post, 'post_d::'))
    save('g16_postnet_wide', **arrays)


def g17_unet_no_bn():
    """The model-seam switches no shipped config uses: ``bn=False`` (Conv2dBnRelu skips the BatchNorm, unet.py:16-17; UNet passes
    the flag to every block and to both heads, unet.py:128-164) and ``dropout=None`` (no Dropout2d modules at all, unet.py:14-15,
    63-72).  a: bn=False with dropout (eval pass + one MC pass under the masks the reference drew); b: bn=False, dropout=None,
    sigma head; c: bn=True, dropout=None."""
    import common.utils.torchhelper as ref_th
    gen = torch.Generator().manual_seed(171)
    arrays = {}
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=4, dropout=0.3, bn=False)
    model = make_unet(17, **params)
    assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in model.modules())
    x = torch.randn(2, 4, 32, 32, generator=gen)
    site_names = [n for n, _ in dropout_modules(model)]
    with torch.no_grad():
        arrays['logits_a_eval'] = model(x).numpy()
        torch.manual_seed(23)
        ref_th.set_dropout_mode(model, True)
        y, recs = capture_masks(model, lambda: model(x))
        ref_th.set_dropout_mode(model, False)
    arrays['logits_a_mc'] = y.numpy()
    for s_, (_, mask) in enumerate(recs):
        arrays['mask_a_{}'.format(s_)] = mask
    arrays.update(params_a=np.array(repr(params)), x_a=x.numpy(), sites_a=np.array(site_names))
    arrays.update(state_to_npz(model, 'sd_a::'))
    params_b = dict(nb_classes=2, in_channels=3, depth=3, start_filters=8, dropout=None, bn=False, sigma_out=True)
    model_b = make_unet(18, **params_b)
    assert not dropout_modules(model_b)
    xb = torch.rand(1, 3, 24, 40, generator=gen)
    with torch.no_grad():
        lb, sb = model_b(xb)
    arrays.update(params_b=np.array(repr(params_b)), x_b=xb.numpy(), logits_b=lb.numpy(), sigma_b=sb.numpy())
    arrays.update(state_to_npz(model_b, 'sd_b::'))
    params_c = dict(nb_classes=2, in_channels=4, depth=4, start_filters=4, dropout=None)
    model_c = make_unet(19, **params_c)
    xc = torch.randn(2, 4, 32, 48, generator=gen)
    with torch.no_grad():
        arrays.update(params_c=np.array(repr(params_c)), x_c=xc.numpy(), logits_c=model_c(xc).numpy())
    arrays.update(state_to_npz(model_c, 'sd_c::'))
    save('g17_unet_no_bn', **arrays)


def g18_unet_stress():
    """Numerics stress case: the full-width BraTS U-Net (start_filters 32: every F(4x4,3x3) instantiation of the HIP
    path on a 192x128 slice pair) with the BatchNorm affines scaled by 2.5 and the classifier by 0.5 -- interior activations of 1e2..1e3,
    logits of +-20 under Dropout2d(0.3) -- through the REFERENCE module: an eval pass and three MC passes under the masks the reference
    drew.  Like G11 the weights are not committed: the tests rebuild them by replaying the constructor's draws
    (oracle.unet_oracle.reference_init_state) and applying the same scaling rule (oracle.unet_oracle.stress_state); the fixture holds
    the input (rounded to multiples of 1/64: it compresses), the masks and a strided sub-sample of the logits."""
    import common.utils.torchhelper as ref_th
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.3)
    bn_gain, head_gain, stride = 2.5, 0.5, 5
    model = make_unet(28, **params)
    with torch.no_grad():
        for name, p_ in list(model.named_parameters()):
            if name.endswith('.bn.weight') or name.endswith('.bn.bias'):
                p_.mul_(bn_gain)
            if name.startswith('conv_cls.1.'):
                p_.mul_(head_gain)
    gen = torch.Generator().manual_seed(281)
    x = torch.round(torch.randn(2, 4, 192, 128, generator=gen) * 64) / 64
    site_names = [n for n, _ in dropout_modules(model)]
    arrays = dict(params=np.array(repr(params)), seed=np.array(28), bn_gain=np.array(bn_gain), head_gain=np.array(head_gain),
                  stride=np.array(stride), x=x.numpy(), sites=np.array(site_names))
    feats = {}
    hook = model.conv_cls.register_forward_pre_hook(lambda mod, inp: feats.__setitem__('f', inp[0].detach()))
    with torch.no_grad():
        y = model(x)
        arrays['logits_eval_strided'] = y.numpy().reshape(-1)[::stride].copy()
        arrays['logits_eval_absmax'] = np.array(float(y.abs().max()))
        arrays['features_eval_absmax'] = np.array(float(feats['f'].abs().max()))
        torch.manual_seed(29)
        ref_th.set_dropout_mode(model, True)
        for t in range(3):
            y, recs = capture_masks(model, lambda: model(x))
            arrays['logits_mc{}_strided'.format(t)] = y.numpy().reshape(-1)[::stride].copy()
            arrays['logits_mc{}_absmax'.format(t)] = np.array(float(y.abs().max()))
            arrays['features_mc{}_absmax'.format(t)] = np.array(float(feats['f'].abs().max()))
            for s_, (_, mask) in enumerate(recs):
                arrays['mask{}_{}'.format(t, s_)] = mask
        ref_th.set_dropout_mode(model, False)
    hook.remove()
    print('G18: |logits| eval {:.2f}, MC {:.2f} / {:.2f} / {:.2f}; |features| MC {:.1f}'.format(
        float(arrays['logits_eval_absmax']), *[float(arrays['logits_mc{}_absmax'.format(t)]) for t in range(3)],
        float(arrays['features_mc0_absmax'])))
    save('g18_unet_stress', **arrays)


def g19_confusion_third_party():
    """a17: Dice / confusion matrix / accuracy.  The reference gets them from pymia 0.2.1 (common/evalutation/numpyfunctions.py:128-151),
    which is absent here -- so this fixture pins the restatement against an INDEPENDENT third party instead: scikit-learn's
    confusion_matrix / f1_score / accuracy_score on the same label pairs (for two classes the Dice coefficient of the foreground IS the
    binary F1 score).  The one convention neither can decide is 0 / 0 (no foreground in prediction and target): sklearn reports what
    ``zero_division`` says; pymia 0.2.1's DiceCoefficient.calculate returns 1 there (published source) -- both variants are stored."""
    import warnings
    from sklearn.metrics import accuracy_score, confusion_matrix, f1_score
    rng = np.random.RandomState(29)
    cases = {}
    shape = (5, 12, 10)
    cases['random_a'] = ((rng.rand(*shape) < 0.3).astype(np.uint8), (rng.rand(*shape) < 0.35).astype(np.uint8))
    t = (rng.rand(*shape) < 0.2).astype(np.uint8)
    noisy = t.copy()
    flip = rng.rand(*shape) < 0.05
    noisy[flip] = 1 - noisy[flip]
    cases['random_overlapping'] = (noisy, t)
    cases['identical'] = (t.copy(), t.copy())
    cases['all_background'] = (np.zeros(shape, np.uint8), np.zeros(shape, np.uint8))
    cases['all_foreground'] = (np.ones(shape, np.uint8), np.ones(shape, np.uint8))
    cases['prediction_empty'] = (np.zeros(shape, np.uint8), t.copy())
    cases['target_empty'] = (t.copy(), np.zeros(shape, np.uint8))
    a = np.zeros(shape, np.uint8)
    b = np.zeros(shape, np.uint8)
    a[:2] = 1
    b[3:] = 1
    cases['empty_intersection'] = (a, b)
    cases['inverted'] = (1 - t, t.copy())
    cases['isic_2d'] = ((rng.rand(64, 48) < 0.4).astype(np.uint8), (rng.rand(64, 48) < 0.4).astype(np.uint8))
    arrays = {'names': np.array(sorted(cases))}
    for name in sorted(cases):
        pred, tgt = cases[name]
        tn, fp, fn, tp = confusion_matrix(tgt.ravel(), pred.ravel(), labels=[0, 1]).ravel()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            f1_zero = f1_score(tgt.ravel(), pred.ravel(), zero_division=0.0)
            f1_one = f1_score(tgt.ravel(), pred.ravel(), zero_division=1.0)
        arrays[name + '::prediction'] = pred
        arrays[name + '::target'] = tgt
        arrays[name + '::counts_tp_tn_fp_fn_n'] = np.array([tp, tn, fp, fn, pred.size], dtype=np.int64)
        arrays[name + '::f1_zero_division_0'] = np.array(f1_zero, dtype=np.float64)
        arrays[name + '::f1_zero_division_1'] = np.array(f1_one, dtype=np.float64)
        arrays[name + '::accuracy'] = np.array(accuracy_score(tgt.ravel(), pred.ravel()), dtype=np.float64)
    import sklearn
    arrays['sklearn_version'] = np.array(sklearn.__version__)
    save('g19_confusion_third_party', **arrays)


def main():
    install_reference()
    torch.set_num_threads(4)
    torch.set_grad_enabled(False)
    for fn in (g1_unet_eval, g2_unet_mc, g3_unet_center, g4_unet_sigma, g5_unet_isic, g6_mc_summary,
               g7_mc_step_end2end, g8_ece, g9_uncertainty, g10_prep, g11_fullsize_digest, g12_eval_csv, g13_postnet, g14_unet_residual, g15_unet_centre_pad,
               g16_postnet_wide, g17_unet_no_bn, g18_unet_stress, g19_confusion_third_party, g21_unet_real_shapes):
        if len(sys.argv) > 1 and fn.__name__ not in sys.argv[1:]:
            continue
        fn()


if __name__ == '__main__':
    main()
