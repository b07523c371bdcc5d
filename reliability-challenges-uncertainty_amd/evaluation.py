"""Metric seam: calibration / uncertainty-error metrics with the reference's protocol, on librcu_hip.

Mirrors
  ece_binary, uncertainty, error_dice/recall/precision, dice, confusion_matrx, accuracy
                                            common/evalutation/numpyfunctions.py:6-151
  EvaluationStrategy family                 common/evalutation/eval.py:9-226
  preparation helpers                       rechun/eval/helper.py:7-47, rechun/eval/analysis.py:147-285
``EvaluationStrategy.__call__(to_evaluate, results)`` takes numpy arrays (or device tensors) in
``to_evaluate`` and writes python / numpy scalars and small arrays into ``results`` under the
reference's keys.  The per-voxel scans run on the GPU: the reliability histogram (bit-exact bin
indices), the 8 confusion x uncertain counts for all thresholds in one pass, the normalised entropy.
What is left on the host is arithmetic on ~30 numbers (ECE from the histogram, Dice from counts).
"""
import abc
import ctypes
import warnings

import numpy as np
import torch

from . import _lib

UE_THRESHOLDS = (0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95)  # bin-eval/eval_uncertainty.py:239


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('rcu_amd.evaluation needs a GPU (librcu_hip); there is no CPU fallback')
    return torch.device('cuda')


def _to_dev(a, dtype):
    # (Measured and NOT done: staging through pinned buffers + asynchronous copies.  Next to a test loop that has run ahead -- every CU held by a
    # persistent conv kernel -- a small operation of the metric seam waits tens of milliseconds whichever way its bytes travel, and the pinned
    # form was the slower one on the same box: tools/host_costs_probe.py, 0.175-0.25 against 0.139 s per subject.  The drop-in scripts
    # therefore take their per-subject Dice counts at batch time, on the compute stream: scripts.ConfusionOnDeviceStep.)
    if isinstance(a, torch.Tensor):
        return a.to(device=_device(), dtype=dtype).contiguous()
    a = np.ascontiguousarray(a)
    if dtype == torch.uint8 and a.dtype.kind in 'iub' and a.dtype.itemsize > 1:
        a = a.astype(np.uint8)       # the cast torch would make on the device (wraps alike), made before the copy: an eighth of the bytes for int64 label maps
    return torch.from_numpy(a).to(device=_device(), dtype=dtype)


# ------------------------------------------------------------------------------------------- ECE
def _foreground(probabilities, target_ndim):
    """numpyfunctions.py:27-33."""
    if probabilities.ndim > target_ndim:
        if probabilities.shape[-1] > 2:
            raise ValueError('can only evaluate the calibration for binary classification')
        if probabilities.shape[-1] == 2:
            return probabilities[..., 1]
        return probabilities.squeeze(-1) if isinstance(probabilities, torch.Tensor) else np.squeeze(probabilities, -1)
    return probabilities


def calibration_histogram(probabilities, target, n_bins=10, mask=None, threshold_range=None, n_volumes=1):
    """Raw reliability histogram(s) on the GPU -> (count int64 [V, n_bins], sum_conf float64, sum_pos int64).
    ``probabilities``: foreground probability (or ``[..., 2]``), float32; ``n_volumes`` > 1 treats the
    leading axis as independent volumes (one launch for a whole test split)."""
    p = _foreground(probabilities, np.ndim(target) if not isinstance(target, torch.Tensor) else target.dim())
    p = _to_dev(p, torch.float32).reshape(n_volumes, -1)
    t = _to_dev(target, torch.uint8).reshape(n_volumes, -1)
    m = None if mask is None else _to_dev(mask, torch.uint8).reshape(n_volumes, -1)
    if threshold_range is not None:  # numpyfunctions.py:39-43: open interval on the confidence
        lo, hi = threshold_range
        keep = ((p < hi) & (p > lo)).to(torch.uint8)
        m = keep if m is None else (m != 0).to(torch.uint8) * keep
    n = p.shape[1]
    lib = _lib.load()
    thr = _lib.ece_thresholds(n_bins)
    result = torch.empty(n_volumes * ctypes.sizeof(_lib.EceResult), device=p.device, dtype=torch.uint8)
    ws = torch.empty(max(lib.rcu_ece_workspace_bytes(n, n_volumes), 8), device=p.device, dtype=torch.uint8)
    _lib.check(lib.rcu_ece_hist(_lib.ptr(p), _lib.ptr(t), _lib.ptr(m), n, n_volumes, thr, n_bins, _lib.ptr(result),
                                _lib.ptr(ws), _lib.current_stream()))
    raw = result.cpu().numpy().view(np.uint64).reshape(n_volumes, 3, _lib.RCU_MAX_BINS)
    count = raw[:, 0, :n_bins].astype(np.int64)
    sum_conf = raw[:, 1, :n_bins].copy().view(np.float64)
    sum_pos = raw[:, 2, :n_bins].astype(np.int64)
    return count, sum_conf, sum_pos


def bin_ids(p, n_bins=10):
    """Bin index per voxel exactly as ``np.digitize(p, linspace(0, 1+1e-8, n_bins+1)) - 1``."""
    p = _to_dev(p, torch.float32).reshape(-1)
    ids = torch.empty(p.numel(), device=p.device, dtype=torch.uint8)
    _lib.check(_lib.load().rcu_ece_bin_ids(_lib.ptr(p), p.numel(), _lib.ece_thresholds(n_bins), n_bins, _lib.ptr(ids),
                                           _lib.current_stream()))
    return ids.cpu().numpy()


def _bin_proportions(bin_weighting, bin_count, non_zero_bins, n_dim):
    # numpyfunctions.py:72-83
    if bin_weighting == 'proportion':
        return bin_count / bin_count.sum()
    if bin_weighting == 'log_proportion':
        return np.log(bin_count) / np.log(bin_count).sum()
    if bin_weighting == 'power_proportion':
        return bin_count ** (1 / n_dim) / (bin_count ** (1 / n_dim)).sum()
    if bin_weighting == 'mean_proportion':
        return 1 / non_zero_bins.sum()
    raise ValueError('unknown bin weighting "{}"'.format(bin_weighting))


def ece_from_histogram(count, sum_conf, sum_pos, n_dim=3, out_bins=None, bin_weighting='proportion'):
    """numpyfunctions.py:65-69 and 14-22 on one volume's raw histogram."""
    nonzero = count != 0
    bin_count = count[nonzero]
    pos_frac = sum_pos[nonzero] / bin_count
    mean_confidence = sum_conf[nonzero] / bin_count
    if out_bins is not None:
        out_bins['bins_count'] = bin_count
        out_bins['bins_avg_confidence'] = mean_confidence
        out_bins['bins_positive_fraction'] = pos_frac
        out_bins['bins_non_zero'] = nonzero
    return (np.abs(mean_confidence - pos_frac) * _bin_proportions(bin_weighting, bin_count, nonzero, n_dim)).sum()


def ece_binary(probabilities, target, n_bins=10, threshold_range: tuple = None, mask=None, out_bins: dict = None,
               bin_weighting='proportion'):
    n_dim = target.dim() if isinstance(target, torch.Tensor) else np.ndim(target)
    count, sum_conf, sum_pos = calibration_histogram(probabilities, target, n_bins, mask, threshold_range)
    return ece_from_histogram(count[0], sum_conf[0], sum_pos[0], n_dim, out_bins, bin_weighting)


# ------------------------------------------------------ everything the evaluation asks of a probability map, from ONE resident copy
class SubjectBatch:
    """``count`` subjects of ``n`` voxels each, resident on the device as [count, n] arrays: foreground probability (float32), prediction,
    target and (optionally) evaluation mask (uint8).  Filled slot by slot from pinned staging buffers (numpy arrays) or device tensors;
    ``metrics`` runs every per-voxel scan of the evaluation script on it in one launch each."""

    def __init__(self, count, n, device=None, with_mask=False):
        self.count, self.n = int(count), int(n)
        self.device = torch.device(device) if device is not None else _device()
        self.with_mask = bool(with_mask)
        shape = (self.count, self.n)
        self.p = torch.empty(shape, device=self.device, dtype=torch.float32)
        self.prediction = torch.empty(shape, device=self.device, dtype=torch.uint8)
        self.target = torch.empty(shape, device=self.device, dtype=torch.uint8)
        self.mask = torch.empty(shape, device=self.device, dtype=torch.uint8) if with_mask else None
        self._pinned = None
        self._staged = set()        # (entry, slot) pairs filled on the host since the last upload
        self.used = 0

    def _staging(self):
        if self._pinned is None:       # one pinned image of the batch: the slots are filled on the host, the batch goes up in four copies
            shape = (self.count, self.n)
            self._pinned = {'p': torch.empty(shape, dtype=torch.float32, pin_memory=True),
                            'prediction': torch.empty(shape, dtype=torch.uint8, pin_memory=True),
                            'target': torch.empty(shape, dtype=torch.uint8, pin_memory=True)}
            if self.with_mask:
                self._pinned['mask'] = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
        return self._pinned

    def put(self, slot, p, prediction, target, mask=None):
        """Subject ``slot`` of the batch: numpy arrays (staged in pinned memory, uploaded by ``upload``) or device tensors (copied in place)."""
        arrays = {'p': p, 'prediction': prediction, 'target': target}
        if self.with_mask:
            if mask is None:
                raise ValueError('this batch was made with a mask')
            arrays['mask'] = mask
        for key, a in arrays.items():
            dst_dev = getattr(self, key)[slot]
            if isinstance(a, torch.Tensor) and a.is_cuda:
                dst_dev.copy_(a.reshape(-1).to(dst_dev.dtype), non_blocking=True)
            else:
                a = np.asarray(a)
                if a.size != self.n:
                    raise ValueError('subject of {} voxels in a batch of {}-voxel slots'.format(a.size, self.n))
                dst = self._staging()[key][slot].numpy()
                np.copyto(dst, a.reshape(-1), casting='unsafe')      # (bool / int64 label maps -> uint8, as torch's cast on the device would)
                self._staged.add((key, slot))
        self.used = max(self.used, slot + 1)

    def upload(self):
        """Host-staged entries -> device (entries that were put as device tensors are there already and stay untouched)."""
        for key in ('p', 'prediction', 'target', 'mask'):
            slots = sorted(s_ for k_, s_ in self._staged if k_ == key)
            if not slots:
                continue
            host, dev = self._pinned[key], getattr(self, key)
            if slots == list(range(slots[0], slots[-1] + 1)):       # the usual case: a run of slots, one copy
                dev[slots[0]:slots[-1] + 1].copy_(host[slots[0]:slots[-1] + 1], non_blocking=True)
            else:
                for s_ in slots:
                    dev[s_].copy_(host[s_], non_blocking=True)
        self._staged = set()

    def metrics(self, n_bins=10, thresholds=UE_THRESHOLDS, want=('minmax', 'ece', 'ue')):
        """-> dict of host arrays over the ``used`` subjects: ``min`` / ``max`` (float32), ``hist`` = (count, sum_conf, sum_pos) of the
        reliability histogram inside the mask, ``counts`` [used, len(thresholds), 8] of the uncertainty-error action on the WHOLE volume
        (bin-eval/eval_uncertainty.py:176-202 uses no mask; tp, tn, fp, fn of it are the confusion matrix of ece_dice).  One launch per
        scan for all subjects, one synchronisation for all results."""
        v = self.used
        lib = _lib.load()
        out, keep = {}, []
        if 'minmax' in want:
            lo, hi = torch.aminmax(self.p[:v], dim=1)
            keep.append(('minmax', torch.stack([lo, hi])))
        if 'ece' in want:
            result = torch.empty(v * ctypes.sizeof(_lib.EceResult), device=self.device, dtype=torch.uint8)
            ws = torch.empty(max(lib.rcu_ece_workspace_bytes(self.n, v), 8), device=self.device, dtype=torch.uint8)
            _lib.check(lib.rcu_ece_hist(_lib.ptr(self.p), _lib.ptr(self.target), _lib.ptr(self.mask), self.n, v, _lib.ece_thresholds(n_bins),
                                        n_bins, _lib.ptr(result), _lib.ptr(ws), _lib.current_stream()))
            keep.append(('ece', result))
        if 'ue' in want:
            thr = (ctypes.c_double * len(thresholds))(*[float(t) for t in thresholds])
            counts = torch.empty((v, len(thresholds), 8), device=self.device, dtype=torch.int64)
            ws2 = torch.empty(lib.rcu_unc_from_p_workspace_bytes(self.n, v), device=self.device, dtype=torch.uint8)
            _lib.check(lib.rcu_unc_counts_from_p(_lib.ptr(self.p), _lib.ptr(self.prediction), _lib.ptr(self.target), None, self.n, v, thr,
                                                 len(thresholds), _lib.ptr(counts), _lib.ptr(ws2), _lib.current_stream()))
            keep.append(('ue', counts))
        host = {k: t.cpu() for k, t in keep}          # (the first .cpu() waits for the stream: the others are ready by then)
        if 'minmax' in host:
            mm = host['minmax'].numpy()
            out['min'], out['max'] = mm[0].copy(), mm[1].copy()
        if 'ece' in host:
            raw = host['ece'].numpy().view(np.uint64).reshape(v, 3, _lib.RCU_MAX_BINS)
            out['hist'] = (raw[:, 0, :n_bins].astype(np.int64), raw[:, 1, :n_bins].copy().view(np.float64), raw[:, 2, :n_bins].astype(np.int64))
        if 'ue' in host:
            out['counts'] = host['ue'].numpy()
        return out


# ---------------------------------------------------------------------- uncertainty-error counts
def _uncertainty_counts_device(prediction, target, uncertainty, thresholds, mask, n_volumes):
    """-> device int64 ``[n_volumes, len(thresholds), 8]`` (see ``uncertainty_counts``); asynchronous on the current stream."""
    is64 = uncertainty.dtype == (torch.float64 if isinstance(uncertainty, torch.Tensor) else np.float64)
    u = _to_dev(uncertainty, torch.float64 if is64 else torch.float32).reshape(n_volumes, -1)
    pr = _to_dev(prediction, torch.uint8).reshape(n_volumes, -1)
    tg = _to_dev(target, torch.uint8).reshape(n_volumes, -1)
    m = None if mask is None else _to_dev(mask, torch.uint8).reshape(n_volumes, -1)
    n = u.shape[1]
    thr = (ctypes.c_double * len(thresholds))(*[float(t) for t in thresholds])
    lib = _lib.load()
    out = torch.empty((n_volumes, len(thresholds), 8), device=u.device, dtype=torch.int64)
    ws = torch.empty(max(lib.rcu_unc_workspace_bytes(n, n_volumes), 8), device=u.device, dtype=torch.uint8)
    _lib.check(lib.rcu_unc_counts(_lib.ptr(u), int(is64), _lib.ptr(pr), _lib.ptr(tg), _lib.ptr(m), n, n_volumes, thr,
                                  len(thresholds), _lib.ptr(out), _lib.ptr(ws), _lib.current_stream()))
    return out


def uncertainty_counts(prediction, target, uncertainty, thresholds=UE_THRESHOLDS, mask=None, n_volumes=1):
    """int64 ``[n_volumes, len(thresholds), 8]`` = tp, tn, fp, fn, tpu, tnu, fpu, fnu with
    uncertain := uncertainty > threshold (numpyfunctions.py:86-107), all thresholds in one GPU pass."""
    return _uncertainty_counts_device(prediction, target, uncertainty, thresholds, mask, n_volumes).cpu().numpy()


def from_p_supported(thresholds):
    """True when the library's table of the reference's uncertain-voxel sets covers these thresholds (strictly ascending, each one of
    bin-eval/eval_uncertainty.py:239's eleven): ``uncertainty_counts_from_p`` then reproduces the reference's counts integer for integer."""
    thresholds = [float(t) for t in thresholds]
    if not 1 <= len(thresholds) <= _lib.RCU_MAX_THRESHOLDS:
        return False
    thr = (ctypes.c_double * len(thresholds))(*thresholds)
    return bool(_lib.load().rcu_unc_from_p_supported(thr, len(thresholds)))


def uncertainty_counts_from_p(prediction, target, foreground_probability, thresholds=UE_THRESHOLDS, mask=None, n_volumes=1):
    """``uncertainty_counts`` for uncertainty = ToEntropy([1 - p, p]) (the 'probabilities' confidence entry, analysis.py:249-252) computed
    from the float32 probability map itself: "uncertain" is looked up in the table of the reference's own float32 sets (include/rcu.h,
    rcu_unc_counts_from_p; fixture g20), so neither an entropy map nor a device log enters -- the counts are the reference's."""
    p = _to_dev(foreground_probability, torch.float32).reshape(n_volumes, -1)
    pr = _to_dev(prediction, torch.uint8).reshape(n_volumes, -1)
    tg = _to_dev(target, torch.uint8).reshape(n_volumes, -1)
    m = None if mask is None else _to_dev(mask, torch.uint8).reshape(n_volumes, -1)
    n = p.shape[1]
    thr = (ctypes.c_double * len(thresholds))(*[float(t) for t in thresholds])
    lib = _lib.load()
    out = torch.empty((n_volumes, len(thresholds), 8), device=p.device, dtype=torch.int64)
    ws = torch.empty(lib.rcu_unc_from_p_workspace_bytes(n, n_volumes), device=p.device, dtype=torch.uint8)
    _lib.check(lib.rcu_unc_counts_from_p(_lib.ptr(p), _lib.ptr(pr), _lib.ptr(tg), _lib.ptr(m), n, n_volumes, thr, len(thresholds),
                                         _lib.ptr(out), _lib.ptr(ws), _lib.current_stream()))
    return out.cpu().numpy()


class EntropyOfProbability:
    """What ``ToEntropy`` leaves under ``uncertainty``: the normalised entropy of ``[1 - p, p]`` as a function of the float32 foreground
    map it holds.  The uncertainty-error strategies hand the MAP to ``uncertainty_counts_from_p`` (exact counts, no entropy volume);
    anything that wants the array (``np.asarray``, arithmetic, indexing) gets the device-computed float64 map, made once."""

    def __init__(self, foreground_probability):
        self.foreground_probability = foreground_probability
        self._array = None

    def materialise(self):
        if self._array is None:
            self._array = normalised_entropy(self.foreground_probability).cpu().numpy()
        return self._array

    def __array__(self, dtype=None, copy=None):
        a = self.materialise()
        return a if dtype is None else a.astype(dtype)

    shape = property(lambda self: tuple(self.foreground_probability.shape))
    dtype = np.dtype(np.float64)

    def __getitem__(self, item):
        return self.materialise()[item]

    def __gt__(self, other):
        return self.materialise() > other

    def min(self):
        return self.materialise().min()

    def max(self):
        return self.materialise().max()


def _counts(prediction, target, uncertainty, thresholds, mask=None):
    """The 8 x len(thresholds) counts of one volume: through the probability table when the uncertainty is ToEntropy's and the
    thresholds are the table's, else by comparing the uncertainty map."""
    if isinstance(uncertainty, EntropyOfProbability):
        if from_p_supported(thresholds):
            return uncertainty_counts_from_p(prediction, target, uncertainty.foreground_probability, thresholds, mask)[0]
        uncertainty = uncertainty.materialise()
    return uncertainty_counts(prediction, target, uncertainty, thresholds, mask)[0]


def uncertainty(prediction, target, thresholded_uncertainty, mask=None):
    """numpyfunctions.py:86-107 for an already thresholded (boolean) map."""
    u = _to_dev(thresholded_uncertainty, torch.uint8).to(torch.float32)
    c = uncertainty_counts(prediction, target, u, thresholds=(0.5,), mask=mask)[0, 0]
    return tuple(int(v) for v in c)


def error_dice(fp, fn, tpu, tnu, fpu, fnu):
    if ((fnu + fpu) == 0) and ((fn + fp + fnu + fpu + tnu + tpu) == 0):
        return 1.
    return (2 * (fnu + fpu)) / (fn + fp + fnu + fpu + tnu + tpu)


def error_recall(fp, fn, fpu, fnu):
    if ((fnu + fpu) == 0) and ((fn + fp) == 0):
        return 1.
    return (fnu + fpu) / (fn + fp)


def error_precision(tpu, tnu, fpu, fnu):
    if ((fnu + fpu) == 0) and ((fnu + fpu + tpu + tnu) == 0):
        return 1.
    return (fnu + fpu) / (fnu + fpu + tpu + tnu)


# pymia 0.2.1 ConfusionMatrix / DiceCoefficient / Accuracy are absent from the reference tree: restated from the call sites
# (numpyfunctions.py:128-151) and pinned against scikit-learn's confusion_matrix / f1_score / accuracy_score (fixture g19); the 0 / 0
# Dice (no foreground in prediction and target) is 1, pymia's convention.
def confusion_matrx(prediction, target):
    c = uncertainty_counts(prediction, target, _zeros_like_map(prediction), thresholds=(0.5,))[0, 0]
    tp, tn, fp, fn = (int(v) for v in c[:4])
    return tp, tn, fp, fn, tp + tn + fp + fn


_zero_maps = {}
_ZERO_MAP_CACHED_ELEMENTS = 1 << 24      # maps up to 64 MB are kept (a BraTS subject: 15.7 MB; a loader batch of 32 slices: 3.1 MB)


def _zeros_like_map(a):
    """An all-zero uncertainty map of a's size (nothing is "uncertain": the first four of the eight counts are the confusion matrix).  Kept
    per size -- read-only to every kernel -- so that a subject's Dice does not start with an allocation and a memset kernel."""
    n = a.numel() if isinstance(a, torch.Tensor) else int(np.prod(np.shape(a)))
    if n > _ZERO_MAP_CACHED_ELEMENTS:
        return torch.zeros(n, device=_device(), dtype=torch.float32)
    z = _zero_maps.get(n)
    if z is None:
        if len(_zero_maps) >= 4:
            _zero_maps.clear()
        z = _zero_maps[n] = torch.zeros(n, device=_device(), dtype=torch.float32)
    return z


def confusion_counts_on_device(prediction, target):
    """Device tensors ``[N, ...]`` uint8 (N slices / images) -> device int64 ``[N, 4]`` = tp, tn, fp, fn per slice, asynchronous on the current
    stream: ``confusion_matrx``'s counts taken where the prediction is made (scripts.ConfusionOnDeviceStep).  Counts are integers: the
    rows of a subject's slices add up to the counts of the assembled subject."""
    n = prediction.shape[0]
    counts = _uncertainty_counts_device(prediction, target, _zeros_like_map(prediction), (0.5,), None, n)
    return counts[:, 0, :4].contiguous()


def dice_from_counts(tp, fp, fn):
    """Dice from ``confusion_matrx``'s integers (0 / 0 = 1: pymia's convention, see above)."""
    return _dice(int(tp), int(fp), int(fn))


def _dice(tp, fp, fn):
    den = 2 * tp + fp + fn
    return 2 * tp / den if den else 1.0


def dice(prediction, target):
    tp, tn, fp, fn, n = confusion_matrx(prediction, target)
    return _dice(tp, fp, fn)


def accuracy(prediction, target):
    tp, tn, fp, fn, n = confusion_matrx(prediction, target)
    return (tp + tn) / n if n else 0.0


def correction_results(counts):
    """All entries UncertaintyAndCorrectionEvalNumpy writes (eval.py:182-226), from the eight counts."""
    tp, tn, fp, fn, tpu, tnu, fpu, fnu = (int(c) for c in counts)
    n = tp + tn + fp + fn
    r = {'tpu': tpu, 'tnu': tnu, 'fpu': fpu, 'fnu': fnu, 'tp': tp, 'tn': tn, 'fp': fp, 'fn': fn}
    with np.errstate(divide='ignore', invalid='ignore'):
        tpu_fpu_ratio = np.float64(tpu) / np.float64(fpu)
        jaccard_index = np.float64(tp) / np.float64(tp + fp + fn)
    r['dice_benefit'] = tpu_fpu_ratio < jaccard_index
    r['accuracy_benefit'] = tpu_fpu_ratio < 1
    r['dice'] = _dice(tp, fp, fn)
    r['accuracy'] = (tp + tn) / n if n else 0.0
    # uncertain voxels set to background: tpu leave tp (become fn), fpu leave fp (become tn)
    r['corrected_dice'] = _dice(tp - tpu, fp - fpu, fn + tpu)
    r['corrected_accuracy'] = (tp - tpu + tn + fpu) / n if n else 0.0
    r['dice_benefit_correct'] = (r['corrected_dice'] > r['dice']) == r['dice_benefit']
    r['accuracy_benefit_correct'] = (r['corrected_accuracy'] > r['accuracy']) == r['accuracy_benefit']
    # uncertain voxels set to foreground: fnu become tp, tnu become fp
    r['corrected_add_dice'] = _dice(tp + fnu, fp + tnu, fn - fnu)
    r['corrected_add_accuracy'] = (tp + fnu + tn - tnu) / n if n else 0.0
    return r


# -------------------------------------------------------------------------------- preparation
def check_min_max(arr, min_=0, max_=1, only_warn=False):
    # rechun/eval/helper.py:31-47
    for bad, txt, val in ((arr.max() > max_, 'larger than {}'.format(max_), arr.max()),
                          (arr.min() < min_, 'smaller than {}'.format(min_), arr.min())):
        if bad:
            message = 'Found value {}: "{}"'.format(txt, val)
            if not only_warn:
                raise ValueError(message)
            warnings.warn(message)


def add_background_probability(probability_np):
    check_min_max(probability_np)
    return np.stack([1 - probability_np, probability_np], axis=-1)


def rescale_uncertainties(uncertainty_np, min_, max_, epsilon=1e-5):
    return (uncertainty_np - min_) / (max_ - min_) * (1 - 2 * epsilon) + epsilon


def uncertainty_to_foreground_probabilities(uncertainty_np, prediction_np):
    if prediction_np.shape != uncertainty_np.shape:
        raise ValueError('shapes must agree. Found {} and {}'.format(uncertainty_np.shape, prediction_np.shape))
    check_min_max(uncertainty_np)
    if prediction_np.max() > 1:
        raise ValueError('Found class larger than 1. Only works for binary problems')
    foreground = uncertainty_np * 0.5
    sel = prediction_np == 1
    foreground[sel] = 1 - foreground[sel]
    return foreground


def normalised_entropy(foreground_probability, as_float64=True):
    """ToEntropy (analysis.py:196-203) of ``[1-p, p]`` on the GPU, from the foreground map alone."""
    p = _to_dev(foreground_probability, torch.float32)
    out = torch.empty(p.shape, device=p.device, dtype=torch.float64 if as_float64 else torch.float32)
    _lib.check(_lib.load().rcu_normalised_entropy(_lib.ptr(p), p.numel(), _lib.ptr(out) if as_float64 else None,
                                                  None if as_float64 else _lib.ptr(out), _lib.current_stream()))
    return out


class PrepareData(abc.ABC):
    @abc.abstractmethod
    def __call__(self, to_eval: dict) -> dict:
        pass


class ComposePreparation(PrepareData):
    def __init__(self, prepare_data_list: list) -> None:
        self.prepare_data_list = prepare_data_list

    def __call__(self, to_eval: dict) -> dict:
        for prepare_data in self.prepare_data_list:
            to_eval = prepare_data(to_eval)
        return to_eval


class AddBackgroundProbabilities(PrepareData):
    def __call__(self, to_eval: dict) -> dict:
        to_eval['probabilities'] = add_background_probability(to_eval['probabilities'])
        return to_eval


class RescaleLinear(PrepareData):
    def __init__(self, entry: str, min_: float, max_: float, epsilon=1e-5) -> None:
        self.entry, self.min, self.max, self.epsilon = entry, min_, max_, epsilon

    def __call__(self, to_eval: dict) -> dict:
        to_eval[self.entry] = rescale_uncertainties(to_eval[self.entry], self.min, self.max, self.epsilon)
        return to_eval


class RescaleSubjectMinMax(PrepareData):
    def __init__(self, entry: str, epsilon=1e-5) -> None:
        self.entry, self.epsilon = entry, epsilon

    def __call__(self, to_eval: dict) -> dict:
        a = to_eval[self.entry]
        to_eval[self.entry] = rescale_uncertainties(a, a.min(), a.max(), self.epsilon)
        return to_eval


class ToForegroundProbabilities(PrepareData):
    def __call__(self, to_eval: dict) -> dict:
        to_eval['probabilities'] = uncertainty_to_foreground_probabilities(to_eval['probabilities'],
                                                                           to_eval['prediction'])
        return to_eval


class ToEntropy(PrepareData):
    def __init__(self, entropy_entry='uncertainty') -> None:
        self.nb_classes = 2
        self.entropy_entry = entropy_entry

    def __call__(self, to_eval: dict) -> dict:
        prob = to_eval['probabilities']
        if prob.shape[-1] != self.nb_classes:
            raise ValueError('last dimension of probability array ({}) must be equal to nb_classes ({})'
                             .format(prob.shape, self.nb_classes))
        # (the reference's check_min_max(..., only_warn=True) can only ever warn here: the entropy of a probability pair lies in
        # [0, 1 + 2e-7]; the map itself is made when somebody asks for it)
        to_eval[self.entropy_entry] = EntropyOfProbability(np.ascontiguousarray(prob[..., 1]))
        return to_eval


class MoveEntry(PrepareData):
    def __init__(self, from_entry: str, to_entry: str) -> None:
        self.from_entry, self.to_entry = from_entry, to_entry

    def __call__(self, to_eval: dict) -> dict:
        to_eval[self.to_entry] = to_eval[self.from_entry]
        return to_eval


def _rescale_prep_and_idstr(confidence_entry, rescale_type, min_max=None):
    # analysis.py:277-285 ('global' reads the min/max CSV; here the pair is passed in)
    if rescale_type == 'global':
        return RescaleLinear(confidence_entry, min_max[0], min_max[1]), '_globalrescale'
    if rescale_type == 'subject':
        return RescaleSubjectMinMax(confidence_entry), '_rescale'
    return None, ''


def get_probability_preparation(confidence_entry, id_, rescale_confidence='subject', rescale_sigma='subject',
                                min_max=None):
    """analysis.py:218-246 -> (preparation, run id with rescale suffix)."""
    if confidence_entry == 'probabilities':
        return ComposePreparation([AddBackgroundProbabilities()]), id_
    rescale = rescale_confidence if confidence_entry == 'confidence' else rescale_sigma
    prepare = []
    prep, suffix = _rescale_prep_and_idstr(confidence_entry, rescale, min_max)
    if prep is not None:
        prepare.append(prep)
    prepare.extend([MoveEntry(confidence_entry, 'probabilities'), ToForegroundProbabilities(),
                    AddBackgroundProbabilities()])
    return ComposePreparation(prepare), id_ + suffix


def get_uncertainty_preparation(confidence_entry, id_, rescale_confidence='', rescale_sigma='global', min_max=None):
    """analysis.py:249-274."""
    if confidence_entry == 'probabilities':
        return ComposePreparation([AddBackgroundProbabilities(), ToEntropy()]), id_
    rescale = rescale_confidence if confidence_entry == 'confidence' else rescale_sigma
    prepare = []
    prep, suffix = _rescale_prep_and_idstr(confidence_entry, rescale, min_max)
    if prep is not None:
        prepare.append(prep)
    prepare.append(MoveEntry(confidence_entry, 'uncertainty'))
    return ComposePreparation(prepare), id_ + suffix


# ----------------------------------------------------------------------- evaluation strategies
class EvaluationStrategy(metaclass=abc.ABCMeta):
    def __init__(self, result_entry=None) -> None:
        self.result_entry = result_entry

    @abc.abstractmethod
    def __call__(self, to_evaluate: dict, results: dict) -> None:
        pass


class ComposeEvaluation(EvaluationStrategy):
    def __init__(self, eval_strategies) -> None:
        super().__init__()
        self.eval_strategies = eval_strategies

    def __call__(self, to_evaluate: dict, results: dict) -> None:
        for eval_ in self.eval_strategies:
            eval_(to_evaluate, results)


class LambdaEvaluation(EvaluationStrategy):
    def __init__(self, lambda_fn, entry_keys: tuple, result_entry) -> None:
        super().__init__(result_entry)
        self.lamda_fn = lambda_fn
        self.entry_keys = entry_keys

    def __call__(self, to_evaluate: dict, results: dict) -> None:
        results[self.result_entry] = self.lamda_fn(*[to_evaluate[k] for k in self.entry_keys])


class DiceNumpy(EvaluationStrategy):
    def __init__(self, result_entry='dice') -> None:
        super().__init__(result_entry)

    def __call__(self, to_evaluate: dict, results: dict) -> None:
        results[self.result_entry] = dice(to_evaluate['prediction'], to_evaluate['target'])


class ConfusionMatrix(EvaluationStrategy):
    def __init__(self, result_entries=('tp', 'tn', 'fp', 'fn', 'n')) -> None:
        super().__init__(result_entries)

    def __call__(self, to_evaluate: dict, results: dict) -> None:
        for key, val in zip(self.result_entry, confusion_matrx(to_evaluate['prediction'], to_evaluate['target'])):
            results[key] = val


class EceBinaryNumpy(EvaluationStrategy):
    def __init__(self, n_bins=10, result_entry='ece', threshold_range: tuple = None, with_mask=False,
                 return_bins=False, bin_weighting='proportion') -> None:
        super().__init__(result_entry)
        self.n_bins = n_bins
        self.threshold_range = threshold_range
        self.with_mask = with_mask
        self.return_bins = return_bins
        self.bin_weighting = bin_weighting

    def __call__(self, to_evaluate: dict, results: dict) -> None:
        mask = to_evaluate['mask'] if self.with_mask else None
        out_bins = results if self.return_bins else None
        results[self.result_entry] = ece_binary(to_evaluate['probabilities'], to_evaluate['target'], self.n_bins,
                                                self.threshold_range, mask, out_bins, self.bin_weighting)


class UncertaintyErrorDiceNumpy(EvaluationStrategy):
    def __init__(self, uncertainty_threshold, result_prefix: str = None, with_mask=False) -> None:
        super().__init__()
        self.uncertainty_threshold = uncertainty_threshold
        self.prefix = '' if result_prefix is None else result_prefix + '_'
        self.with_mask = with_mask

    def __call__(self, to_evaluate: dict, results: dict):
        mask = ~to_evaluate['target_boarder'] if self.with_mask else None
        c = _counts(to_evaluate['prediction'], to_evaluate['target'], to_evaluate['uncertainty'], (self.uncertainty_threshold,), mask)[0]
        tp, tn, fp, fn, tpu, tnu, fpu, fnu = (int(v) for v in c)
        results['{}precision'.format(self.prefix)] = error_precision(tpu, tnu, fpu, fnu)
        results['{}recall'.format(self.prefix)] = error_recall(fp, fn, fpu, fnu)
        results['{}dice'.format(self.prefix)] = error_dice(fp, fn, tpu, tnu, fpu, fnu)


class UncertaintyAndCorrectionEvalNumpy(EvaluationStrategy):
    def __init__(self, uncertainty_threshold) -> None:
        super().__init__()
        self.uncertainty_threshold = uncertainty_threshold

    def __call__(self, to_evaluate: dict, results: dict) -> None:
        c = _counts(to_evaluate['prediction'], to_evaluate['target'], to_evaluate['uncertainty'], (self.uncertainty_threshold,))[0]
        results.update(correction_results(c))


class UncertaintyAndCorrectionSweep(EvaluationStrategy):
    """All thresholds of the 'bnf_ue' action (eval_uncertainty.py:176-202, 239) in ONE pass over the
    volume; ``results[threshold]`` holds what UncertaintyAndCorrectionEvalNumpy would write for it."""

    def __init__(self, thresholds=UE_THRESHOLDS) -> None:
        super().__init__()
        self.thresholds = tuple(thresholds)

    def __call__(self, to_evaluate: dict, results: dict) -> None:
        c = _counts(to_evaluate['prediction'], to_evaluate['target'], to_evaluate['uncertainty'], self.thresholds)
        for i, thr in enumerate(self.thresholds):
            results[thr] = correction_results(c[i])
