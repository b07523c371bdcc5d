"""Directory conventions of the reference (rechun/directories.py), same names.

The reference asks its user to edit the module ("dirs and path required to be set", directories.py:4-28); here every one of
those settings can also come from the environment -- ``RCU_<NAME>`` (e.g. ``RCU_BRATS_ORIG_DATA_DIR``, ``RCU_BRATS_BASELINE_MC_PREDICT``) --
so that the unchanged command line of the reference's scripts (``bin-eval/eval_uncertainty.py --ds --ids --act``,
eval_uncertainty.py:248-251) works without touching source files.  The derived directories (directories.py:31-60) follow from
them exactly as in the reference; ``RCU_PROJECT_DIR`` moves the project root (default: the repository root)."""
import os


def _setting(name, default=''):
    return os.environ.get('RCU_' + name, default)


PROJECT_DIR = _setting('PROJECT_DIR', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# dirs and paths required to be set (directories.py:7-28)
BRATS_ORIG_DATA_DIR = _setting('BRATS_ORIG_DATA_DIR')
ISIC_ORIG_DATA_DIR = _setting('ISIC_ORIG_DATA_DIR')

RUN_IDS = ('baseline', 'baseline_mc', 'center', 'center_mc', 'ensemble', 'auxiliary_feat', 'auxiliary_segm', 'aleatoric')
_SETTING_OF_RUN = {'baseline': 'BASELINE', 'baseline_mc': 'BASELINE_MC', 'center': 'CENTER', 'center_mc': 'CENTER_MC',
                   'ensemble': 'ENSEMBLE', 'auxiliary_feat': 'AUX_FEAT', 'auxiliary_segm': 'AUX_SEGM', 'aleatoric': 'ALEATORIC'}
for _ds in ('ISIC', 'BRATS'):
    for _run, _key in _SETTING_OF_RUN.items():
        # e.g. BRATS_BASELINE_MC_PREDICT = '<timestamp>_brats_baseline_mc'; unset: the run id itself (out/predictions/brats/baseline_mc)
        globals()['{}_{}_PREDICT'.format(_ds, _key)] = _setting('{}_{}_PREDICT'.format(_ds, _key), _run)
BRATS_CV_PREDICT = _setting('BRATS_CV_PREDICT')

# important directories (directories.py:31-60)
CONFIG_DIR = os.path.join(PROJECT_DIR, 'config')
SPLITS_DIR = os.path.join(CONFIG_DIR, 'splits')
DATASET_DIR = os.path.join(PROJECT_DIR, 'in', 'datasets')

ISIC_PREPROCESSED_DIR = _setting('ISIC_PREPROCESSED_DIR', os.path.join(DATASET_DIR, 'isic_small'))
ISIC_PREPROCESSED_TRAIN_DATA_DIR = os.path.join(ISIC_PREPROCESSED_DIR, 'ISIC-2017_Training')
ISIC_PREPROCESSED_TEST_DATA_DIR = os.path.join(ISIC_PREPROCESSED_DIR, 'ISIC-2017_Test_v2')

ISIC_ORIG_TRAIN_DATA_DIR = os.path.join(ISIC_ORIG_DATA_DIR, 'ISIC-2017_Training')
ISIC_ORIG_VALID_DATA_DIR = os.path.join(ISIC_ORIG_DATA_DIR, 'ISIC-2017_Validation')
ISIC_ORIG_TEST_DATA_DIR = os.path.join(ISIC_ORIG_DATA_DIR, 'ISIC-2017_Test_v2')

PREDICT_DIR = _setting('PREDICT_DIR', os.path.join(PROJECT_DIR, 'out', 'predictions'))
ISIC_PREDICT_DIR = os.path.join(PREDICT_DIR, 'isic')
BRATS_PREDICT_DIR = os.path.join(PREDICT_DIR, 'brats')

EVAL_DIR = _setting('EVAL_DIR', os.path.join(PROJECT_DIR, 'out', 'eval'))
ISIC_EVAL_DIR = os.path.join(EVAL_DIR, 'isic')
BRATS_EVAL_DIR = os.path.join(EVAL_DIR, 'brats')

PLOT_DIR = os.path.join(PROJECT_DIR, 'out', 'plots')
ISIC_PLOT_DIR = os.path.join(PLOT_DIR, 'isic')
BRATS_PLOT_DIR = os.path.join(PLOT_DIR, 'brats')

# definitions used in evaluation & analysis (directories.py:63-75)
ECE_FOREGROUND_NAME = 'ece_foreground'
ECE_NAME = 'ece'
CALIB_NAME = 'calibration'
UNCERTAINTY_NAME = 'uncertainty'
MINMAX_NAME = 'minmax'

CALIBRATION_PLACEHOLDER = 'eval_calibration_{}.csv'
UNCERTAINTY_PLACEHOLDER = 'eval_uncertainty_{}_th{}.csv'
ECE_PLACEHOLDER = 'eval_ece_{}.csv'
MINMAX_PLACEHOLDER = 'eval_summary_minmax_{}.csv'


def prediction_dir(dataset, run_id):
    """Prediction directory of a run, as rechun/eval/evaldata.py:21-46 composes it: <DS>_PREDICT_DIR / <DS>_<RUN>_PREDICT."""
    ds = dataset.upper()
    return os.path.join(globals()['{}_PREDICT_DIR'.format(ds)], globals()['{}_{}_PREDICT'.format(ds, _SETTING_OF_RUN[run_id])])


def ground_truth_dir(dataset):
    """Where the evaluation reads the ground truth from (rechun/eval/evaldata.py:55, 82-83): the original BraTS training tree, the
    preprocessed ISIC test set."""
    return BRATS_ORIG_DATA_DIR if dataset == 'brats' else ISIC_PREPROCESSED_TEST_DATA_DIR


def eval_dir(dataset):
    return BRATS_EVAL_DIR if dataset == 'brats' else ISIC_EVAL_DIR
