"""Oracle: calibration (ECE) and uncertainty-error metrics in numpy.  TEST INFRASTRUCTURE ONLY.

Follows common/evalutation/numpyfunctions.py, common/evalutation/eval.py:176-226,
rechun/eval/analysis.py:147-215 and rechun/eval/helper.py:7-47 of the reference.
"""
import warnings

import numpy as np

UE_THRESHOLDS = (0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95)  # bin-eval/eval_uncertainty.py:239


# ---------------------------------------------------------------- ECE (numpyfunctions.py:6-83)

def bin_edges(n_bins=10):
    """float64 edges; the upper edge is nudged so that p == 1.0 falls in the last bin (nf.py:53)."""
    return np.linspace(0., 1. + 1e-8, n_bins + 1)


def bin_ids(p, n_bins=10):
    """digitize - 1 (nf.py:54): the number of interior/lower edges <= p, minus one."""
    return np.digitize(p, bin_edges(n_bins)) - 1


def float32_thresholds(n_bins=10):
    """Smallest float32 t_k with float64(t_k) >= edge_k, k = 1..n_bins-1, so that for float32 p
    ``bin = sum_k [p >= t_k]`` equals ``bin_ids(p)`` exactly (SURVEY 8a row a10 lists the n_bins=10
    bit patterns).  This is what the HIP kernel and the C oracle compare against."""
    edges = bin_edges(n_bins)[1:-1]
    thr = edges.astype(np.float32)
    low = thr.astype(np.float64) < edges
    thr[low] = np.nextafter(thr[low], np.float32(np.inf))
    return thr


def select_foreground(probabilities, target, mask=None, threshold_range=None):
    """nf.py:26-43: take the foreground column, apply the mask, apply the open threshold range."""
    p = probabilities
    if p.ndim > target.ndim:
        if p.shape[-1] > 2:
            raise ValueError('can only evaluate the calibration for binary classification')
        p = p[..., 1] if p.shape[-1] == 2 else np.squeeze(p, axis=-1)
    t = target
    if mask is not None:
        p, t = p[mask], t[mask]
    if threshold_range is not None:
        lo, hi = threshold_range
        keep = np.logical_and(p < hi, p > lo)
        p, t = p[keep], t[keep]
    return p.reshape(-1), t.reshape(-1)


def calibration_histogram(p, t, n_bins=10):
    """Raw per-bin (count i64, sum of confidences f64, sum of positives f64) over all n_bins
    (nf.py:61-63 before the non-empty filtering)."""
    ids = bin_ids(p, n_bins)
    count = np.bincount(ids, minlength=n_bins)
    sum_conf = np.bincount(ids, weights=p, minlength=n_bins)
    sum_pos = np.bincount(ids, weights=t, minlength=n_bins)
    return count, sum_conf, sum_pos


def ece_from_histogram(count, sum_conf, sum_pos, bin_weighting='proportion', n_dim=3, out_bins=None):
    """nf.py:65-69 + 14-22 + 72-83 from the three raw histograms."""
    count = np.asarray(count)
    nonzero = count != 0
    cnt = count[nonzero]
    pos_frac = np.asarray(sum_pos, dtype=np.float64)[nonzero] / cnt
    mean_conf = np.asarray(sum_conf, dtype=np.float64)[nonzero] / cnt
    if bin_weighting == 'proportion':
        w = cnt / cnt.sum()
    elif bin_weighting == 'log_proportion':
        w = np.log(cnt) / np.log(cnt).sum()
    elif bin_weighting == 'power_proportion':
        w = cnt ** (1 / n_dim) / (cnt ** (1 / n_dim)).sum()
    elif bin_weighting == 'mean_proportion':
        w = 1 / nonzero.sum()
    else:
        raise ValueError('unknown bin weighting "{}"'.format(bin_weighting))
    if out_bins is not None:
        out_bins['bins_count'] = cnt
        out_bins['bins_avg_confidence'] = mean_conf
        out_bins['bins_positive_fraction'] = pos_frac
        out_bins['bins_non_zero'] = nonzero
    return (np.abs(mean_conf - pos_frac) * w).sum()


def ece_binary(probabilities, target, n_bins=10, threshold_range=None, mask=None, out_bins=None,
               bin_weighting='proportion'):
    p, t = select_foreground(probabilities, target, mask, threshold_range)
    return ece_from_histogram(*calibration_histogram(p, t, n_bins), bin_weighting=bin_weighting,
                              n_dim=target.ndim, out_bins=out_bins)


# ------------------------------------------------ uncertainty-error counts (nf.py:86-125)

def uncertainty_counts(prediction, target, thresholded_uncertainty, mask=None):
    """tp, tn, fp, fn, tpu, tnu, fpu, fnu (nf.py:86-107); inputs boolean arrays."""
    if mask is not None:
        prediction, target = prediction[mask], target[mask]
        thresholded_uncertainty = thresholded_uncertainty[mask]
    prediction = prediction.astype(bool)
    target = target.astype(bool)
    u = thresholded_uncertainty.astype(bool)
    cells = [target & prediction, ~target & ~prediction, ~target & prediction, target & ~prediction]
    base = [int(c.sum()) for c in cells]
    unc = [int((c & u).sum()) for c in cells]
    return tuple(base + unc)


def error_dice(fp, fn, tpu, tnu, fpu, fnu):
    den = fn + fp + fnu + fpu + tnu + tpu
    if (fnu + fpu) == 0 and den == 0:
        return 1.
    return (2 * (fnu + fpu)) / den


def error_recall(fp, fn, fpu, fnu):
    if (fnu + fpu) == 0 and (fn + fp) == 0:
        return 1.
    return (fnu + fpu) / (fn + fp)


def error_precision(tpu, tnu, fpu, fnu):
    den = fnu + fpu + tpu + tnu
    if (fnu + fpu) == 0 and den == 0:
        return 1.
    return (fnu + fpu) / den


# --- pymia 0.2.1 ConfusionMatrix / DiceCoefficient / Accuracy (package absent; restated from the call sites nf.py:128-151 and pymia's
# published algorithm: counts over ==1 / ==0, Dice 2tp/(2tp+fp+fn) -- 1 when there is no foreground at all --, accuracy (tp+tn)/n).
# Pinned against an independent third party: fixture g19 = scikit-learn's confusion_matrix / f1_score / accuracy_score on the same
# label pairs (tests/test_oracle_golden.py); only the 0 / 0 Dice convention rests on pymia's source alone.

def confusion_counts(prediction, target):
    prediction = np.asarray(prediction)
    target = np.asarray(target)
    tp = int(np.sum((prediction == 1) & (target == 1)))
    tn = int(np.sum((prediction == 0) & (target == 0)))
    fp = int(np.sum((prediction == 1) & (target == 0)))
    fn = int(np.sum((prediction == 0) & (target == 1)))
    return tp, tn, fp, fn, int(prediction.size)


def dice_from_counts(tp, fp, fn):
    den = 2 * tp + fp + fn
    return 2 * tp / den if den else 1.0


def accuracy_from_counts(tp, tn, n):
    return (tp + tn) / n if n else 0.0


def correction_metrics(counts):
    """Everything UncertaintyAndCorrectionEvalNumpy (eval.py:182-226) writes, derived from the eight
    counts alone.  Correcting uncertain voxels to background removes tpu from tp and fpu from fp;
    correcting to foreground turns fnu into tp and tnu into fp."""
    tp, tn, fp, fn, tpu, tnu, fpu, fnu = (int(c) for c in counts)
    n = tp + tn + fp + fn
    r = dict(tpu=tpu, tnu=tnu, fpu=fpu, fnu=fnu, tp=tp, tn=tn, fp=fp, fn=fn)
    with np.errstate(divide='ignore', invalid='ignore'):
        ratio = np.float64(tpu) / np.float64(fpu)
        jaccard = np.float64(tp) / np.float64(tp + fp + fn)
    r['dice_benefit'] = bool(ratio < jaccard)
    r['accuracy_benefit'] = bool(ratio < 1)
    r['dice'] = dice_from_counts(tp, fp, fn)
    r['accuracy'] = accuracy_from_counts(tp, tn, n)
    r['corrected_dice'] = dice_from_counts(tp - tpu, fp - fpu, fn + tpu)
    r['corrected_accuracy'] = accuracy_from_counts(tp - tpu, tn + fpu, n)
    r['dice_benefit_correct'] = (r['corrected_dice'] > r['dice']) == r['dice_benefit']
    r['accuracy_benefit_correct'] = (r['corrected_accuracy'] > r['accuracy']) == r['accuracy_benefit']
    r['corrected_add_dice'] = dice_from_counts(tp + fnu, fp + tnu, fn - fnu)
    r['corrected_add_accuracy'] = accuracy_from_counts(tp + fnu, tn - tnu, n)
    return r


# ------------------------------------------------------ entropy / preparation (analysis.py, helper.py)

def numpy_entropy(p, dim=-1, keepdims=False):
    """nf.py:166-168.  For float32 p the products are float32; the [0.0] operand of ``where`` promotes
    the selected values to float64 before the sum."""
    with np.errstate(divide='ignore', invalid='ignore'):
        return -np.where(p > 0, p * np.log(p), [0.0]).sum(axis=dim, keepdims=keepdims)


def check_min_max(arr, min_=0, max_=1, only_warn=False):
    """helper.py:31-47."""
    for bad, txt, val in ((arr.max() > max_, 'larger than {}'.format(max_), arr.max()),
                          (arr.min() < min_, 'smaller than {}'.format(min_), arr.min())):
        if bad:
            text = 'Found value {}: "{}"'.format(txt, val)
            if only_warn:
                warnings.warn(text)
            else:
                raise ValueError(text)


def add_background_probability(p):
    """[1-p, p] stacked on a new last axis, after a [0,1] range check (helper.py:25-28)."""
    check_min_max(p)
    return np.stack([1 - p, p], axis=-1)


def rescale_uncertainties(u, min_, max_, epsilon=1e-5):
    """helper.py:19-22: to [0,1] then shrunk into [eps, 1-eps]."""
    return (u - min_) / (max_ - min_) * (1 - 2 * epsilon) + epsilon


def uncertainty_to_foreground_probabilities(u, prediction):
    """helper.py:7-16: u/2 where predicted background, 1 - u/2 where predicted foreground."""
    if prediction.shape != u.shape:
        raise ValueError('shapes must agree. Found {} and {}'.format(u.shape, prediction.shape))
    check_min_max(u)
    if prediction.max() > 1:
        raise ValueError('Found class larger than 1. Only works for binary problems')
    fg = u * 0.5
    sel = prediction == 1
    fg[sel] = 1 - fg[sel]
    return fg


def normalised_entropy(probabilities2):
    """ToEntropy (analysis.py:196-203): entropy of [1-p, p] divided by log 2, float64 out."""
    if probabilities2.shape[-1] != 2:
        raise ValueError('last dimension of probability array ({}) must be equal to nb_classes (2)'
                         .format(probabilities2.shape))
    return numpy_entropy(probabilities2) / np.log(2)


def probability_preparation(confidence_entry, to_eval, rescale='subject', min_max=None):
    """get_probability_preparation (analysis.py:218-246) for one subject dict; returns the
    [..., 2] probabilities.  rescale in {'subject', 'global', ''}; 'global' needs min_max."""
    if confidence_entry == 'probabilities':
        return add_background_probability(to_eval['probabilities'])
    u = to_eval[confidence_entry]
    if rescale == 'subject':
        u = rescale_uncertainties(u, u.min(), u.max())
    elif rescale == 'global':
        u = rescale_uncertainties(u, min_max[0], min_max[1])
    return add_background_probability(uncertainty_to_foreground_probabilities(u, to_eval['prediction']))


def uncertainty_preparation(confidence_entry, to_eval, rescale='subject', min_max=None):
    """get_uncertainty_preparation (analysis.py:249-274): normalised entropy for probability runs,
    the (rescaled) confidence / sigma map itself otherwise."""
    if confidence_entry == 'probabilities':
        return normalised_entropy(add_background_probability(to_eval['probabilities']))
    u = to_eval[confidence_entry]
    if rescale == 'subject':
        u = rescale_uncertainties(u, u.min(), u.max())
    elif rescale == 'global':
        u = rescale_uncertainties(u, min_max[0], min_max[1])
    return u
