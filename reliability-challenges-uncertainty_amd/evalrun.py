"""Evaluation-script surface: what ``bin-eval/eval_uncertainty.py`` of the reference does, on the GPU.

Mirrors
  CSV hooks            rechun/eval/hook.py:10-116 (WriteCsvHook, WriteBinsCsvHook, WriteSummaryCsvHook)
  file / run registry  rechun/eval/evaldata.py:8-103, common/data/collector.py:120-174, rechun/directories.py:56-71
  loader               rechun/eval/analysis.py:15-125 (probabilities / target>0 / prediction / T2 brain mask, cached)
  actions + driver     bin-eval/eval_uncertainty.py:13-244 (minmax, ece_dice, calib, bnf_ue)
so that the CSV files ``bin-analysis/*`` consumes keep their names, columns and row order.  The volumes
are read with rcu_amd.nifti, the per-voxel work (histograms, counts, entropy) runs through
rcu_amd.evaluation on the GPU; the ``bnf_ue`` action evaluates its 11 thresholds in ONE pass per subject
and fans the result out to the 11 per-threshold CSV files the reference writes.
"""
import abc
import csv
import glob
import os
import time

import numpy as np

from . import evaluation as ev
from . import nifti

# rechun/directories.py:56-71
ECE_FOREGROUND_NAME = 'ece_foreground'
ECE_NAME = 'ece'
CALIB_NAME = 'calibration'
UNCERTAINTY_NAME = 'uncertainty'
MINMAX_NAME = 'minmax'
CALIBRATION_PLACEHOLDER = 'eval_calibration_{}.csv'
UNCERTAINTY_PLACEHOLDER = 'eval_uncertainty_{}_th{}.csv'
ECE_PLACEHOLDER = 'eval_ece_{}.csv'
MINMAX_PLACEHOLDER = 'eval_summary_minmax_{}.csv'

CONFIDENCE_ENTRY = {'baseline': 'probabilities', 'baseline_mc': 'probabilities', 'center': 'probabilities',
                    'center_mc': 'probabilities', 'ensemble': 'probabilities', 'auxiliary_feat': 'confidence',
                    'auxiliary_segm': 'confidence', 'aleatoric': 'sigma'}   # evaldata.py:21-47


# ------------------------------------------------------------------------------------- CSV hooks
class EvalHook:
    def on_run_start(self, run_id: str):
        pass

    def on_subject(self, results: dict, subject_name: str, run_id: str):
        pass

    def on_run_end(self, results_history: dict, run_id: str):
        pass


class ReducedComposeEvalHook(EvalHook):
    """Calls only the methods a member hook really overrides (common/trainloop/hooks.py:116-133)."""

    def __init__(self, hooks: list) -> None:
        for name in ('on_run_start', 'on_subject', 'on_run_end'):
            fns = [getattr(h, name) for h in hooks if getattr(type(h), name) is not getattr(EvalHook, name)]
            setattr(self, name, self._chain(fns))

    @staticmethod
    def _chain(fns):
        def call(*args, **kwargs):
            for fn in fns:
                fn(*args, **kwargs)
        return call


class WriteCsvHook(EvalHook):
    """One row per subject: ``test_id, subject_name, <entries>``; list-valued results are unfolded to
    ``key_<i>`` columns with zero-padded indices (hook.py:28-72)."""

    def __init__(self, file_path: str, entries=None) -> None:
        self.file_path = file_path
        self.rows = []
        self.entries = None if entries is None else list(entries)
        self.header = None

    @staticmethod
    def _unfold_results(results):
        flat = {}
        for key, value in results.items():
            if isinstance(value, np.ndarray):
                value = value.tolist()
            if isinstance(value, (list, tuple)):
                digits = len(str(len(value)))
                for i, v in enumerate(value):
                    flat['{}_{:0{}d}'.format(key, i, digits)] = v
            else:
                flat[key] = value
        return flat

    def on_subject(self, results: dict, subject_name: str, run_id: str):
        flat = self._unfold_results(results)
        if self.entries is None:
            self.entries = list(flat.keys())
        if self.header is None:
            self.header = ['test_id', 'subject_name'] + self.entries
        self.rows.append([run_id, subject_name] + [flat[e] for e in self.entries])

    def on_run_end(self, results_history: dict, run_id: str):
        with open(self.file_path, 'w', newline='') as f:
            writer = csv.writer(f)
            writer.writerow(self.header)
            writer.writerows(self.rows)


class WriteBinsCsvHook(WriteCsvHook):
    """Re-expands the non-empty-bin arrays to all bins before unfolding (hook.py:75-93)."""

    def on_subject(self, results: dict, subject_name: str, run_id: str):
        non_zero = results['bins_non_zero']
        for key in ('bins_count', 'bins_avg_confidence', 'bins_positive_fraction'):
            full = np.zeros_like(non_zero, dtype=results[key].dtype)
            full[non_zero] = results[key]
            results[key] = full
        super().on_subject(results, subject_name, run_id)


class WriteSummaryCsvHook(EvalHook):
    """``confidence_entry, min, max`` over the whole run (hook.py:96-116)."""

    def __init__(self, file_path: str, entries=('min', 'max'), summary_fn=(np.min, np.max),
                 confidence_entry='probabilities') -> None:
        if len(entries) != len(summary_fn):
            raise ValueError('entries and summary_fn must be of same length')
        self.file_path = file_path
        self.entries = list(entries)
        self.summary_fn = list(summary_fn)
        self.confidence_entry = confidence_entry

    def on_run_end(self, results_history: dict, run_id: str):
        with open(self.file_path, 'w', newline='') as f:
            writer = csv.writer(f)
            writer.writerow(['confidence_entry'] + self.entries)
            writer.writerow([self.confidence_entry] + [fn(results_history[e]) for e, fn in
                                                       zip(self.entries, self.summary_fn)])


def read_min_max(min_max_file: str):
    """rechun/eval/helper.py:50-55."""
    with open(min_max_file, 'r') as f:
        reader = csv.reader(f)
        next(reader)
        _, min_, max_ = next(reader)
    return float(min_), float(max_)


# ------------------------------------------------------------------------------ files and runs
class SubjectFiles:
    """subject id + {category: {entry: path}} (the part of pymia's SubjectFile the path uses)."""

    def __init__(self, subject, **categories):
        self.subject = subject
        self.categories = {k: dict(v) for k, v in categories.items()}


def collect_predictions(prediction_path, post_fixes, categories):
    """``**/<subject>_<postfix>.nii.gz`` under a prediction directory (collector.py:120-161)."""
    by_id = {}
    for pf in post_fixes:
        tail = '_{}.nii.gz'.format(pf)
        for path in glob.glob(os.path.join(prediction_path, '**', '*' + tail), recursive=True):
            by_id.setdefault(os.path.basename(path)[:-len(tail)], {})[pf] = path
    out = []
    for subject, files in by_id.items():
        if set(files) != set(post_fixes):
            raise AssertionError('id "{}" has not all required entries "({})"'.format(subject, list(post_fixes)))
        cats = {}
        for pf, cat in zip(post_fixes, categories):
            cats.setdefault(cat, {})[pf] = files[pf]
        out.append(SubjectFiles(subject, **cats))
    return out


def collect_brats_ground_truth(root_dir):
    """``**/<subject>/<subject>_{flair,t1,t2,t1ce,seg}.nii.gz`` (collector.py:17-71); subject = directory name."""
    out = []
    for flair in sorted(glob.glob(os.path.join(root_dir, '**', '*_flair.nii.gz'), recursive=True)):
        stem = flair[:-len('_flair.nii.gz')]
        images = {'flair': flair, 't1': stem + '_t1.nii.gz', 't2': stem + '_t2.nii.gz', 't1c': stem + '_t1ce.nii.gz'}
        labels = {'gt': stem + '_seg.nii.gz'} if os.path.exists(stem + '_seg.nii.gz') else {}
        out.append(SubjectFiles(os.path.basename(os.path.dirname(flair)), images=images, labels=labels))
    return out


def collect_isic_ground_truth(root_dir_with_prefix):
    """``<prefix>_Data/<id>.jpg`` + ``<prefix>_Part1_GroundTruth/<id>_segmentation.png`` (collector.py:75-120);
    subject = the first 12 characters of the file name."""
    by_id = {}
    for path in glob.glob(root_dir_with_prefix + '_Data/*') + glob.glob(root_dir_with_prefix + '_Part1_GroundTruth/*'):
        name = os.path.basename(path)
        if name.endswith('_segmentation.png'):
            by_id.setdefault(name[:12], {})['gt'] = path
        elif name.endswith('.jpg'):
            by_id.setdefault(name[:12], {})['image'] = path
    out = []
    for id_, files in sorted(by_id.items()):
        if 'gt' in files and 'image' in files:
            out.append(SubjectFiles(id_, images={'image': files['image']}, labels={'gt': files['gt']}))
    return out


def read_label_image(path, dtype=np.uint8):
    """NIfTI, or the png masks of ISIC (the reference reads both through ``sitk.ReadImage(path, sitkUInt8)``)."""
    if str(path).endswith(('.png', '.jpg')):
        from PIL import Image
        return np.array(Image.open(path).convert('L')).astype(dtype)
    return nifti.read(path, dtype)[0]


def combine(files_from, files_to):
    """collector.py:164-174: add the categories of ``files_from`` to the same subject in ``files_to``."""
    by_id = {sf.subject: sf for sf in files_from}
    for sf in files_to:
        for cat, entries in by_id[sf.subject].categories.items():
            sf.categories.setdefault(cat, {}).update(entries)
    return files_to


class EvalData:
    def __init__(self, id_, eval_path, confidence_entry='probabilities', subject_files=None) -> None:
        self.id_ = id_
        self.eval_path = eval_path
        self.confidence_entry = confidence_entry
        self.subject_files = subject_files if subject_files is not None else []


def get_eval_data(run_id, prediction_dir, ground_truth_files, expected_subjects=None):
    """One run: collect ``*_prediction`` + ``*_<confidence entry>`` files and join them with the ground truth."""
    entry = EvalData(run_id, prediction_dir, CONFIDENCE_ENTRY.get(run_id, 'probabilities'))
    preds = collect_predictions(prediction_dir, ['prediction', entry.confidence_entry], ['labels', 'misc'])
    preds = combine(ground_truth_files, preds)
    if expected_subjects is not None:
        assert set(expected_subjects) == set(sf.subject for sf in preds)
    entry.subject_files = sorted(preds, key=lambda sf: sf.subject)
    return entry


# -------------------------------------------------------------------------------------- loader
class Loader:
    """Per-subject cached reads (analysis.py:15-125)."""

    class Params:
        def __init__(self, misc_entry='probabilities', need_target=True, need_prediction=True, need_t2_mask=False):
            self.misc_entry = misc_entry
            self.need_target = need_target
            self.need_prediction = need_prediction
            self.need_t2_mask = need_t2_mask

    def __init__(self) -> None:
        self.cached = {}
        self.cached_subject = None

    def _get(self, key, fn):
        if key not in self.cached:
            self.cached[key] = fn()
        return self.cached[key].copy()

    def get_data(self, sf: SubjectFiles, params):
        if sf.subject != self.cached_subject:
            self.cached.clear()
            self.cached_subject = sf.subject
        to_eval = {params.misc_entry: self._get(params.misc_entry,
                                                lambda: nifti.read(sf.categories['misc'][params.misc_entry])[0])}
        if params.need_target:   # labels 0..4 are binarised (analysis.py:88-89)
            to_eval['target'] = self._get('target', lambda: (read_label_image(sf.categories['labels']['gt']) > 0)
                                          .astype(np.uint8))
        if params.need_prediction:
            to_eval['prediction'] = self._get('prediction', lambda: nifti.read(sf.categories['labels']['prediction'],
                                                                               np.uint8)[0])
        if params.need_t2_mask:
            to_eval['mask'] = self._get('mask', lambda: nifti.read(sf.categories['images']['t2'])[0] > 0)
        return to_eval


# ------------------------------------------------------------------------------------- actions
class EvalCase:
    def __init__(self, metric, hook, id_='') -> None:
        self.result_history = {}
        self.metric = metric
        self.hook = hook
        self.id_ = id_

    def record(self, results, subject_name, id_):
        self.hook.on_subject(results, subject_name, id_)
        for k, v in results.items():
            self.result_history.setdefault(k, []).append(v)

    def do_eval(self, to_eval, subject_name, id_):
        results = {}
        self.metric(to_eval, results)
        self.record(results, subject_name, id_)


class EvalAction(abc.ABC):
    def __init__(self) -> None:
        self.load_params = None
        self.prepare = None
        self.eval_cases = []
        self.id_ = ''

    @abc.abstractmethod
    def setup_eval(self, eval_data: EvalData):
        pass

    def start_eval(self):
        print(self.id_ + ', '.join(c.id_ for c in self.eval_cases if c.id_ != ''))
        for case in self.eval_cases:
            case.hook.on_run_start(self.id_)

    def eval_subject(self, sf, loader):
        to_eval = loader.get_data(sf, self.load_params)
        if self.prepare:
            to_eval = self.prepare(to_eval)
        for case in self.eval_cases:
            case.do_eval(to_eval, sf.subject, self.id_)

    def finish_eval(self):
        for case in self.eval_cases:
            case.hook.on_run_end(case.result_history, self.id_)


def _minmax_for(min_max_dir, run_id, rescale):
    if rescale != 'global':
        return None
    return read_min_max(os.path.join(min_max_dir, MINMAX_PLACEHOLDER.format(run_id)))


class SaveMinMaxAction(EvalAction):
    def __init__(self, min_max_dir: str) -> None:
        super().__init__()
        self.min_max_dir = min_max_dir
        os.makedirs(min_max_dir, exist_ok=True)

    def setup_eval(self, eval_data):
        self.id_ = eval_data.id_
        self.prepare = ev.MoveEntry(eval_data.confidence_entry, 'probabilities')
        self.load_params = Loader.Params(eval_data.confidence_entry)
        metric = ev.ComposeEvaluation([ev.LambdaEvaluation(lambda x: x.min(), ('probabilities',), 'min'),
                                       ev.LambdaEvaluation(lambda x: x.max(), ('probabilities',), 'max')])
        hook = WriteSummaryCsvHook(os.path.join(self.min_max_dir, MINMAX_PLACEHOLDER.format(self.id_)),
                                   confidence_entry=eval_data.confidence_entry)
        self.eval_cases = [EvalCase(metric, hook)]


class EceAction(EvalAction):
    def __init__(self, base_dir, details, rescale_confidence='subject', rescale_sigma='subject', min_max_dir=None):
        super().__init__()
        self.rescale_confidence, self.rescale_sigma, self.min_max_dir = rescale_confidence, rescale_sigma, min_max_dir
        self.need_t2_mask = details == 'foreground'
        self.out_dir = os.path.join(base_dir, ECE_FOREGROUND_NAME if self.need_t2_mask else ECE_NAME)
        os.makedirs(self.out_dir, exist_ok=True)

    def setup_eval(self, eval_data):
        rescale = self.rescale_confidence if eval_data.confidence_entry == 'confidence' else self.rescale_sigma
        mm = None if eval_data.confidence_entry == 'probabilities' else _minmax_for(self.min_max_dir, eval_data.id_, rescale)
        self.prepare, self.id_ = ev.get_probability_preparation(eval_data.confidence_entry, eval_data.id_,
                                                                self.rescale_confidence, self.rescale_sigma, mm)
        self.load_params = Loader.Params(eval_data.confidence_entry, need_t2_mask=self.need_t2_mask)
        metric = ev.ComposeEvaluation([ev.EceBinaryNumpy(threshold_range=None, with_mask=self.need_t2_mask),
                                       ev.DiceNumpy(), ev.ConfusionMatrix()])
        hook = WriteCsvHook(os.path.join(self.out_dir, ECE_PLACEHOLDER.format(self.id_)),
                            entries=('ece', 'dice', 'tp', 'tn', 'fp', 'fn', 'n'))
        self.eval_cases = [EvalCase(metric, hook)]


class EceCalibrationAction(EvalAction):
    def __init__(self, base_dir, details='', rescale_confidence='subject', rescale_sigma='subject', min_max_dir=None):
        super().__init__()
        self.need_mask = details == 'foreground'
        self.rescale_confidence, self.rescale_sigma, self.min_max_dir = rescale_confidence, rescale_sigma, min_max_dir
        self.out_dir = os.path.join(base_dir, CALIB_NAME)
        os.makedirs(self.out_dir, exist_ok=True)

    def setup_eval(self, eval_data):
        rescale = self.rescale_confidence if eval_data.confidence_entry == 'confidence' else self.rescale_sigma
        mm = None if eval_data.confidence_entry == 'probabilities' else _minmax_for(self.min_max_dir, eval_data.id_, rescale)
        self.prepare, self.id_ = ev.get_probability_preparation(eval_data.confidence_entry, eval_data.id_,
                                                                self.rescale_confidence, self.rescale_sigma, mm)
        self.load_params = Loader.Params(eval_data.confidence_entry, need_t2_mask=self.need_mask)
        metric = ev.ComposeEvaluation([ev.EceBinaryNumpy(threshold_range=None, return_bins=True,
                                                         with_mask=self.need_mask), ev.DiceNumpy()])
        hook = WriteBinsCsvHook(os.path.join(self.out_dir, CALIBRATION_PLACEHOLDER.format(self.id_)))
        self.eval_cases = [EvalCase(metric, hook)]


class CorrectionAction(EvalAction):
    """11 CSV files (one per threshold) from one GPU pass per subject."""

    def __init__(self, thresholds, base_dir, rescale_confidence='', rescale_sigma='global', min_max_dir=None):
        super().__init__()
        self.thresholds = list(thresholds)
        self.rescale_confidence, self.rescale_sigma, self.min_max_dir = rescale_confidence, rescale_sigma, min_max_dir
        self.out_dir = os.path.join(base_dir, UNCERTAINTY_NAME)
        os.makedirs(self.out_dir, exist_ok=True)

    def setup_eval(self, eval_data):
        rescale = self.rescale_confidence if eval_data.confidence_entry == 'confidence' else self.rescale_sigma
        mm = None if eval_data.confidence_entry == 'probabilities' else _minmax_for(self.min_max_dir, eval_data.id_, rescale)
        self.prepare, self.id_ = ev.get_uncertainty_preparation(eval_data.confidence_entry, eval_data.id_,
                                                                self.rescale_confidence, self.rescale_sigma, mm)
        self.load_params = Loader.Params(eval_data.confidence_entry)
        self.sweep = ev.UncertaintyAndCorrectionSweep(self.thresholds)
        self.eval_cases = []
        for thr in self.thresholds:
            thr_str = '{:.2f}'.format(thr).replace('.', '')
            hook = WriteCsvHook(os.path.join(self.out_dir, UNCERTAINTY_PLACEHOLDER.format(self.id_, thr_str)), None)
            self.eval_cases.append(EvalCase(None, hook))

    def eval_subject(self, sf, loader):
        to_eval = loader.get_data(sf, self.load_params)
        if self.prepare:
            to_eval = self.prepare(to_eval)
        results = {}
        self.sweep(to_eval, results)
        for thr, case in zip(self.thresholds, self.eval_cases):
            case.record(results[thr], sf.subject, self.id_)


ECE_TYPES = {EceAction, EceCalibrationAction}


def get_actions(action_names, min_max_dir, base_dir, ece_details):
    """bin-eval/eval_uncertainty.py:226-244."""
    actions = []
    for name in action_names:
        if name == 'minmax':
            actions.append(SaveMinMaxAction(min_max_dir))
        elif name == 'ece_dice':
            actions.append(EceAction(base_dir, ece_details, 'subject', 'global', min_max_dir))
        elif name == 'calib':
            actions.append(EceCalibrationAction(base_dir, ece_details, 'subject', 'global', min_max_dir))
        elif name == 'bnf_ue':
            actions.append(CorrectionAction(ev.UE_THRESHOLDS, base_dir, 'subject', 'global', min_max_dir))
    return actions


# ------------------------------------------------------------------------ the fused subject loop
class _ReadAhead:
    """The files of the coming subjects, read by a few threads while the current ones are evaluated (zlib releases the GIL: the .nii.gz
    streams really inflate side by side).  One task per FILE -- a subject's two to four files inflate side by side too, and the float32
    maps do not queue behind a neighbour's label images.  ``get(i)`` -> the ``to_eval`` dict of subject i (blocks until its files are in)."""

    def __init__(self, subject_files, params, depth, threads=None):
        import concurrent.futures
        if threads is None:       # gunzip is what the fused loop waits for (tools/eval_throughput.py): as many streams as the host can spare, at most 8
            threads = min(8, max(2, (os.cpu_count() or 4) // 2))
        self.subject_files, self.params, self.depth = subject_files, params, max(1, int(depth))
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=threads, thread_name_prefix='rcu-eval-read')
        self.futures = {}
        self.next = 0

    @staticmethod
    def _timed(fn, *args):
        t0 = time.perf_counter()
        return fn(*args), time.perf_counter() - t0

    def _tasks(self, sf):
        cats, p = sf.categories, self.params
        tasks = {p.misc_entry: (lambda path: nifti.read(path)[0], cats['misc'][p.misc_entry])}
        if p.need_target:
            tasks['target'] = (lambda path: (read_label_image(path) > 0).astype(np.uint8), cats['labels']['gt'])       # analysis.py:88-89
        if p.need_prediction:
            tasks['prediction'] = (lambda path: nifti.read(path, np.uint8)[0], cats['labels']['prediction'])
        if p.need_t2_mask:
            tasks['mask'] = (lambda path: nifti.read(path)[0] > 0, cats['images']['t2'])
        return tasks

    def _fill(self, upto):
        while self.next < min(upto, len(self.subject_files)):
            self.futures[self.next] = {key: self.pool.submit(self._timed, fn, path)
                                       for key, (fn, path) in self._tasks(self.subject_files[self.next]).items()}
            self.next += 1

    def get(self, i):
        self._fill(i + 1 + self.depth)
        entry = self.futures.pop(i)
        if isinstance(entry, _Done):
            return entry.result()
        out, read_s = {}, 0.0
        for key, future in entry.items():
            out[key], seconds = future.result()
            read_s += seconds
        out['_read_s'] = read_s
        return out

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)


class _LoaderAhead:
    """``Loader`` objects for the coming subjects of the reference-ordered loop, their caches filled by the reader threads: the union of what
    the run's actions will ask ``Loader.get_data`` for (one task per file, as ``_ReadAhead``)."""

    def __init__(self, subject_files, params_list, depth):
        # (the actions of a run share its confidence entry; an action that wants another one reads it itself: a cache miss in Loader)
        union = Loader.Params(params_list[0].misc_entry, need_target=False, need_prediction=False, need_t2_mask=False)
        for p in params_list:
            union.need_target |= bool(p.need_target)
            union.need_prediction |= bool(p.need_prediction)
            union.need_t2_mask |= bool(p.need_t2_mask)
        self.reader = _ReadAhead(subject_files, union, depth)
        self.subject_files = subject_files

    def get(self, i):
        sf = self.subject_files[i]
        loader = Loader()
        loader.cached_subject = sf.subject
        data = self.reader.get(i)
        data.pop('_read_s', None)
        loader.cached.update(data)
        return loader

    def close(self):
        self.reader.close()


def _fusable(entry, actions):
    """The fused loop covers the runs whose confidence entry IS the probability map (baseline, baseline_mc, center, center_mc, ensemble:
    evaldata.py:21-47) -- no rescaling, no uncertainty-to-probability conversion -- and the four actions of the script."""
    masks = {bool(getattr(a, 'need_t2_mask', False) or getattr(a, 'need_mask', False)) for a in actions if type(a) in ECE_TYPES}
    return (entry.confidence_entry == 'probabilities' and len(masks) <= 1 and
            all(type(a) in (SaveMinMaxAction, EceAction, EceCalibrationAction, CorrectionAction) for a in actions) and
            all(ev.from_p_supported(a.thresholds) for a in actions if isinstance(a, CorrectionAction)))


def metrics_wanted(actions):
    """(`want` of evaluation.SubjectBatch.metrics, thresholds of the uncertainty-error counts, whether the ECE actions use a mask) for a
    list of actions on a probability-map run."""
    by_type = {type(a): a for a in actions}
    want = (['ece'] if (ECE_TYPES & set(by_type)) else []) + ['minmax'] + \
           (['ue'] if (CorrectionAction in by_type or (ECE_TYPES & set(by_type))) else [])
    ue = by_type.get(CorrectionAction)
    want_mask = any(getattr(a, 'need_t2_mask', False) or getattr(a, 'need_mask', False) for a in actions)
    return want, (tuple(ue.thresholds) if ue is not None else (0.5,)), want_mask


def record_subject(actions, subject, res, slot, n_dim):
    """Fan the metrics of subject ``slot`` of a `SubjectBatch.metrics` result out to the actions' CSV hooks -- the rows (keys, key order,
    value types) the per-action strategies of the reference-ordered loop produce."""
    mn, mx = res['min'][slot], res['max'][slot]
    # helper.add_background_probability's range check (rechun/eval/helper.py:8-12, 31-47), on the device's min / max
    if any(not isinstance(a, SaveMinMaxAction) for a in actions):
        if mx > 1:
            raise ValueError('Found value larger than 1: "{}"'.format(mx))
        if mn < 0:
            raise ValueError('Found value smaller than 0: "{}"'.format(mn))
    counts = res['counts'][slot] if 'counts' in res else None
    for action in actions:
        if isinstance(action, SaveMinMaxAction):
            action.eval_cases[0].record({'min': mn, 'max': mx}, subject, action.id_)
        elif isinstance(action, (EceAction, EceCalibrationAction)):
            hist = [h[slot] for h in res['hist']]
            tp, tn, fp, fn = (int(v) for v in counts[0][:4])
            results = {}
            if isinstance(action, EceCalibrationAction):      # key order of EceBinaryNumpy(return_bins=True) + DiceNumpy
                ece = ev.ece_from_histogram(*hist, n_dim=n_dim, out_bins=results)
                results['ece'] = ece
                results['dice'] = ev._dice(tp, fp, fn)
            else:
                results['ece'] = ev.ece_from_histogram(*hist, n_dim=n_dim)
                results['dice'] = ev._dice(tp, fp, fn)
                results.update(tp=tp, tn=tn, fp=fp, fn=fn, n=tp + tn + fp + fn)
            action.eval_cases[0].record(results, subject, action.id_)
        elif isinstance(action, CorrectionAction):
            for t, case in enumerate(action.eval_cases):
                case.record(ev.correction_results(counts[t]), subject, action.id_)


def _evaluate_fused(entry, actions, batch_subjects, timing):
    """All actions of a 'probabilities' run from ONE upload per subject and ONE launch per scan and batch of subjects: files read ahead by
    threads, subjects of equal size staged side by side in pinned memory, `evaluation.SubjectBatch.metrics` -- reliability histogram inside
    the mask (ece_dice and calib share it), the uncertainty-error counts of all thresholds from the probability map (their tp / tn / fp /
    fn are ece_dice's confusion matrix), min / max -- and the results fanned out to the actions' CSV hooks in subject order.  The rows are
    those of the per-action loop, byte for byte (tests/test_gpu_parity.py)."""
    want, thresholds, want_mask = metrics_wanted(actions)
    params = Loader.Params('probabilities', need_target=True, need_prediction=True, need_t2_mask=want_mask)
    files = entry.subject_files
    reader = _ReadAhead(files, params, depth=2 * batch_subjects)
    batches = {}          # voxels per subject -> SubjectBatch (datasets have one size; a mixed one gets a batch object per size)
    try:
        i = 0
        while i < len(files):
            t_start = time.perf_counter()
            first = reader.get(i)
            n_vox, n_dim = first['probabilities'].size, first['target'].ndim
            group = [(i, first)]
            while len(group) < batch_subjects and i + len(group) < len(files):
                nxt = reader.get(i + len(group))
                if nxt['probabilities'].size != n_vox:
                    reader.futures[i + len(group)] = _Done(nxt)       # another size: it opens the next batch
                    break
                group.append((i + len(group), nxt))
            t_read = time.perf_counter()
            batch = batches.get((n_vox, want_mask))
            if batch is None or batch.count < len(group):
                batch = batches[(n_vox, want_mask)] = ev.SubjectBatch(max(batch_subjects, len(group)), n_vox, with_mask=want_mask)
            batch.used = 0
            for slot, (_, d) in enumerate(group):
                batch.put(slot, d['probabilities'], d['prediction'], d['target'], d.get('mask'))
            t_stage = time.perf_counter()
            batch.upload()
            res = batch.metrics(thresholds=thresholds, want=want)
            t_gpu = time.perf_counter()
            for slot, (k, d) in enumerate(group):
                record_subject(actions, files[k].subject, res, slot, n_dim)
            t_end = time.perf_counter()
            per = (t_end - t_start) / len(group)
            for k, d in group:
                print('[{}/{}] {} ({}s)'.format(k + 1, len(files), files[k].subject, per))
            if timing is not None:
                timing['subjects'] += len(group)
                timing['batches'] += 1
                timing['wait_for_files_s'] += t_read - t_start
                timing['read_thread_s'] += sum(d['_read_s'] for _, d in group)
                timing['stage_s'] += t_stage - t_read
                timing['upload_and_kernels_s'] += t_gpu - t_stage
                timing['csv_rows_s'] += t_end - t_gpu
            i += len(group)
    finally:
        reader.close()


class _Done:
    def __init__(self, value):
        self.value = value

    def result(self):
        return self.value


def evaluate_runs(eval_data_list, action_names, base_dir, ece_details='', fused=True, batch_subjects=8, timing=None):
    """The subject loop of bin-eval/eval_uncertainty.py:13-50 for already collected runs.
    ``fused`` (default): runs whose confidence entry is the probability map go through ``_evaluate_fused`` -- one upload per subject shared by
    all actions, ``batch_subjects`` subjects per launch, files read ahead; the other runs (confidence / sigma entries: host-side
    rescaling recipes) and ``fused=False`` take the reference's subject-by-subject, action-by-action order.
    ``timing``: a dict that receives where the fused loop's time went (tools/eval_throughput.py)."""
    actions = get_actions(action_names, os.path.join(base_dir, MINMAX_NAME), base_dir, ece_details)
    for entry in eval_data_list:
        for action in actions:
            action.setup_eval(entry)
        for action in actions:
            action.start_eval()
        if fused and entry.subject_files and _fusable(entry, actions):
            if timing is not None:
                for key in ('subjects', 'batches', 'wait_for_files_s', 'read_thread_s', 'stage_s', 'upload_and_kernels_s', 'csv_rows_s'):
                    timing.setdefault(key, 0)
            _evaluate_fused(entry, actions, max(1, int(batch_subjects)), timing)
            for action in actions:
                action.finish_eval()
            continue
        # the reference's subject-by-subject, action-by-action order (eval_uncertainty.py:36-46); the files of the coming subjects are read by
        # the threads of _ReadAhead meanwhile, into the caches of the subjects' Loaders (what an action asks for first is there already)
        wanted = [a.load_params for a in actions if a.load_params is not None]
        ahead = _LoaderAhead(entry.subject_files, wanted, depth=4) if (wanted and len(entry.subject_files) > 1) else None
        try:
            for i, sf in enumerate(entry.subject_files):
                print('[{}/{}] {}'.format(i + 1, len(entry.subject_files), sf.subject), end=' ', flush=True)
                loader = ahead.get(i) if ahead is not None else Loader()
                start = time.time()
                for action in actions:
                    action.eval_subject(sf, loader)
                print('({}s)'.format(time.time() - start))
        finally:
            if ahead is not None:
                ahead.close()
        for action in actions:
            action.finish_eval()
