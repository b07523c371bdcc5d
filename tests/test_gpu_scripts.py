"""End-to-end runs of the drop-in test scripts on the GPU: YAML config -> model dir + checkpoint ->
volume dataset -> Test loop -> NIfTI + metrics.csv, then the evaluation driver on what was written."""
import csv
import glob
import json
import os

import numpy as np
import pytest
import torch

from test_script_surface_cpu import BRATS_MC_YAML

pytestmark = pytest.mark.gpu
PARAMS = dict(nb_classes=2, in_channels=4, depth=4, start_filters=32, dropout=0.05)


def _setup(tmp_path, mc=None, sigma=False, seeds=(20,), shape=(32, 32)):
    from oracle import unet_oracle as uo
    from rcu_amd import data as data_mod
    from rcu_amd import management as mgt
    from rcu_amd import nifti
    rng = np.random.RandomState(3)
    params = dict(PARAMS, sigma_out=True) if sigma else dict(PARAMS)
    vols = {}
    for i in range(2):
        name = 'Brats18_T{}_1'.format(i)
        images = rng.randn(6 + i, shape[0], shape[1], 4).astype(np.float32)
        labels = (rng.rand(6 + i, shape[0], shape[1]) < 0.3).astype(np.uint8)
        props = nifti.ImageProperties((shape[1], shape[0], 6 + i), (1.0, -2.0, 3.0), (1.0, 1.0, 2.5))
        data_mod.write_volume(str(tmp_path / 'ds'), name, images, labels, props)
        vols[name] = (images, labels, props)
    states, dirs = [], []
    for seed in seeds:
        st = uo.synthetic_state(seed, **params)
        mf = mgt.ModelFiles(str(tmp_path / 'train_{}'.format(seed)), 'm{}'.format(seed))
        mgt.save_model(mf, 'unet', params, st, epoch=2)
        states.append(st)
        dirs.append(mf.model_dir)
    split = str(tmp_path / 'split.json')
    with open(split, 'w') as f:
        json.dump({'train': [], 'valid': [], 'test': list(vols)}, f)
    text = BRATS_MC_YAML.format(test_dir=str(tmp_path / 'out'), model_dir=dirs[0], split=split, dataset=str(tmp_path / 'ds'))
    text = text.replace('batch_size: 32', 'batch_size: 4')
    if mc is None:
        text = text.replace('  others:\n    mc: 20\n', '  others: {}\n')
    else:
        text = text.replace('mc: 20', 'mc: {}'.format(mc))
    if len(dirs) > 1:
        extra = ''.join('    - {}\n'.format(d) for d in dirs[1:])
        text = text.replace('  others: {}\n', '  others:\n    model_dir:\n{}    test_at: best\n'.format(extra))
    cfg_path = str(tmp_path / 'test_cfg.yaml')
    with open(cfg_path, 'w') as f:
        f.write(text)
    return cfg_path, vols, states, params


def _outputs(context, vols):
    from rcu_amd import nifti
    out = {}
    for name in vols:
        p, props = nifti.read(os.path.join(context.test_dir, name + '_probabilities.nii.gz'))
        pred, _ = nifti.read(os.path.join(context.test_dir, name + '_prediction.nii.gz'))
        out[name] = (p, pred, props)
    return out


def test_brats_default_deterministic_and_mc(tmp_path):
    from oracle import unet_oracle as uo
    from oracle import calib_oracle as co
    from rcu_amd import scripts
    cfg_path, vols, states, params = _setup(tmp_path)
    ctx = scripts.test_default('brats', cfg_path, None)
    assert os.path.basename(ctx.test_dir).endswith('_brats_test_baseline_mc')
    for f in ('config.yaml', 'log.txt', 'metrics.csv', 'split.json'):
        assert os.path.exists(os.path.join(ctx.test_dir, f)), f
    outs = _outputs(ctx, vols)
    rows = list(csv.DictReader(open(os.path.join(ctx.test_dir, 'metrics.csv'))))
    assert [r['subject'] for r in rows] == sorted(vols)
    for name, (images, labels, props) in vols.items():
        x = torch.from_numpy(images).permute(0, 3, 1, 2)
        ref = torch.softmax(uo.unet_forward(states[0], x, None, **params), 1)[:, 1].numpy()
        p, pred, rprops = outs[name]
        assert p.dtype == np.float32 and pred.dtype == np.uint8 and rprops == props
        assert np.max(np.abs(p - ref)) < 1e-4
        agree = np.mean(pred == (ref > 0.5))
        assert agree > 0.999
        tp, tn, fp, fn, n = co.confusion_counts(pred, labels)
        row = [r for r in rows if r['subject'] == name][0]
        assert abs(float(row['dice']) - co.dice_from_counts(tp, fp, fn)) < 1e-12
    # MC-dropout config: same script, others.mc set; probabilities stay close to the deterministic pass
    cfg_mc, vols2, _, _ = _setup(tmp_path / 'mc', mc=6)
    ctx2 = scripts.test_default('brats', cfg_mc, None)
    outs2 = _outputs(ctx2, vols2)
    for name in vols2:
        assert np.all((outs2[name][0] >= 0) & (outs2[name][0] <= 1))
        assert 0 < np.max(np.abs(outs2[name][0] - outs[name][0])) < 0.5


def test_brats_ensemble_and_aleatoric_then_eval(tmp_path):
    from oracle import summary_oracle as so
    from oracle import unet_oracle as uo
    from rcu_amd import nifti, scripts
    cfg_path, vols, states, params = _setup(tmp_path / 'ens', seeds=(20, 21, 22))
    ctx = scripts.test_ensemble('brats', cfg_path)
    outs = _outputs(ctx, vols)
    for name, (images, _, _) in vols.items():
        x = torch.from_numpy(images).permute(0, 3, 1, 2)
        multi = so.ensemble_probabilities([lambda xx, m, st=st: uo.unet_forward(st, xx, m, **params) for st in states], x)
        ref = multi.mean(0)[:, 1].numpy()
        assert np.max(np.abs(outs[name][0] - ref)) < 1e-4
    cfg_al, vols_al, st_al, params_al = _setup(tmp_path / 'al', sigma=True)
    ctx_al = scripts.test_aleatoric('brats', cfg_al)
    for name, (images, _, _) in vols_al.items():
        x = torch.from_numpy(images).permute(0, 3, 1, 2)
        logits, sigma = uo.unet_forward(st_al[0], x, None, **params_al)
        pred = logits.argmax(1)
        ref_sigma = torch.gather(sigma.abs(), 1, pred[:, None])[:, 0].numpy()
        got = nifti.read(os.path.join(ctx_al.test_dir, name + '_sigma.nii.gz'))[0]
        close = np.abs(got - ref_sigma) < 1e-4
        assert close.mean() > 0.999      # voxels whose two classes tie within fp32 noise may pick the other sigma
    # evaluation driver on the ensemble run: needs a BraTS-style ground-truth tree
    gt = tmp_path / 'gt' / 'HGG'
    for name, (images, labels, props) in vols.items():
        (gt / name).mkdir(parents=True)
        for mod, arr in (('flair', images[..., 0]), ('t1', images[..., 1]), ('t2', images[..., 2]), ('t1ce', images[..., 3]),
                         ('seg', labels * 4)):
            nifti.write(str(gt / name / '{}_{}.nii.gz'.format(name, mod)), arr, props)
    scripts.eval_uncertainty('brats', {'ensemble': ctx.test_dir}, str(tmp_path / 'gt'), str(tmp_path / 'eval'),
                             expected_subjects=list(vols))
    assert len(glob.glob(str(tmp_path / 'eval' / 'uncertainty' / 'eval_uncertainty_ensemble_th*.csv'))) == 11
    rows = list(csv.DictReader(open(str(tmp_path / 'eval' / 'ece_foreground' / 'eval_ece_ensemble.csv'))))
    assert [r['subject_name'] for r in rows] == sorted(vols) and all(0 <= float(r['ece']) <= 1 for r in rows)
    # subjects of two sizes (6 and 7 slices): the fused loop batches equal sizes only -- still the bytes of the subject-by-subject loop
    scripts.eval_uncertainty('brats', {'ensemble': ctx.test_dir}, str(tmp_path / 'gt'), str(tmp_path / 'eval_plain'), fused=False)
    a = {os.path.relpath(f, str(tmp_path / 'eval')): open(f, 'rb').read() for f in glob.glob(str(tmp_path / 'eval' / '**' / '*.csv'), recursive=True)}
    b = {os.path.relpath(f, str(tmp_path / 'eval_plain')): open(f, 'rb').read()
         for f in glob.glob(str(tmp_path / 'eval_plain' / '**' / '*.csv'), recursive=True)}
    assert a == b and len(a) == 14


def _files(ctx):
    return {os.path.basename(f): open(f, 'rb').read() for f in sorted(glob.glob(os.path.join(ctx.test_dir, '*.nii.gz')))}


def _with_others(cfg_path, suffix, test_dir=None, **others):
    """A copy of the YAML file with keys added to (None: removed from) ``config.others`` -- the rcu_amd loop options ride there;
    ``test_dir``: another output root (two runs within one second share the time-stamped directory name otherwise)."""
    import yaml
    with open(cfg_path) as f:
        doc = yaml.safe_load(f)
    # (every variant gets an output root of its own: two runs within one second would share the time-stamped directory otherwise)
    doc['config']['test_dir'] = test_dir if test_dir is not None else '{}_{}'.format(doc['config']['test_dir'], suffix)
    cur = dict(doc['config'].get('others') or {})
    for k, v in others.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = v
    doc['config']['others'] = cur
    out = cfg_path.replace('.yaml', '_{}.yaml'.format(suffix))
    with open(out, 'w') as f:
        yaml.safe_dump(doc, f)
    return out


def test_dice_counts_taken_at_batch_time_are_the_subject_level_ones(tmp_path, monkeypatch):
    """scripts.ConfusionOnDeviceStep: the subjects' Dice counts are taken slice by slice on the compute stream behind the predict steps and
    travel with the batches (the default for the BraTS scripts) -- metrics.csv and the files must be those of the subject-level evaluation
    (``others.device_confusion: false``: EvalSubjectStep's own arg-max + evaluation.confusion_matrx, the reference's order), on batches that
    straddle subjects (6 and 7 slices in batches of 4), coalesced into one step, and in the reference-ordered serial loop; and the default
    run must not call the subject-level GPU path at all (its three synchronous GPU operations are what the step exists to avoid)."""
    from rcu_amd import evaluation as ev
    from rcu_amd import loops, scripts
    cfg, vols, _, _ = _setup(tmp_path, mc=3)

    def run(suffix, **others):
        ctx = scripts.test_default('brats', _with_others(cfg, suffix, **others), None)
        return _files(ctx), open(os.path.join(ctx.test_dir, 'metrics.csv'), 'rb').read()

    calls = []
    inner = ev.confusion_matrx
    monkeypatch.setattr(ev, 'confusion_matrx', lambda *a, **k: (calls.append(1), inner(*a, **k))[1])
    files_dev, csv_dev = run('dev')
    assert not calls
    files_sub, csv_sub = run('sub', device_confusion=False)
    assert len(calls) == len(vols)
    assert csv_dev == csv_sub and b'dice' in csv_dev.splitlines()[0] and len(csv_dev.splitlines()) == 1 + len(vols)
    assert sorted(files_dev) == sorted(files_sub) and all(files_dev[k] == files_sub[k] for k in files_dev)
    del calls[:]
    _, csv_serial = run('serial', pipelined=False)
    assert csv_serial == csv_dev and not calls
    # the loader's batches as they are (coalescing is the default since round 6): its own subject-level run
    _, csv_plain = run('plain', coalesce_pixels=0)
    _, csv_plain_sub = run('plain_sub', coalesce_pixels=0, device_confusion=False)
    assert csv_plain == csv_plain_sub
    # the Dice values themselves: from the written prediction files against the dataset's labels
    from rcu_amd import nifti
    rows = {line.split(b',')[0].decode(): line.split(b',') for line in csv_dev.splitlines()[1:]}
    col = csv_dev.splitlines()[0].split(b',').index(b'dice')
    ctx = scripts.test_default('brats', _with_others(cfg, 'again'), None)
    for name, (_, labels, _) in vols.items():
        pred = nifti.read(os.path.join(ctx.test_dir, name + '_prediction.nii.gz'))[0].astype(bool)
        tgt = labels.astype(bool)
        den = pred.sum() + tgt.sum()
        want = 2.0 * (pred & tgt).sum() / den if den else 1.0
        assert abs(float(rows[name][col]) - want) < 1e-12, name


def test_pipelined_coalesced_loop_writes_the_files_of_the_serial_loop(tmp_path, monkeypatch):
    """The test loop's pipeline (loader thread, batches coalesced up to a volume, outputs downloaded on a side stream, NIfTI files
    written by a pool of threads) must not change a byte of what the reference-ordered serial loop writes: deterministic config --
    the pipelined loop with coalescing (the default since round 6) against ``others.pipelined: false`` without coalescing
    (``coalesce_pixels: 0``), whole .nii.gz files compared; MC-dropout config -- the pipelined loop against the serial one under the same
    seed, and the coalesced run against the loader's own batches: the seeded masks are keyed by the slice, not by the batch, so both draw
    the same MC samples."""
    from rcu_amd import loops, scripts
    cfg_det, vols, _, _ = _setup(tmp_path / 'det')
    fast = _files(scripts.test_default('brats', _with_others(cfg_det, 'fast', coalesce_pixels=loops.Test.COALESCE_PIXELS), None))
    slow = _files(scripts.test_default('brats', _with_others(cfg_det, 'slow', pipelined=False, coalesce_pixels=0), None))
    assert sorted(fast) == sorted(slow) and len(fast) == 2 * len(vols)
    for name in fast:
        assert fast[name] == slow[name], name
    cfg_mc, vols_mc, _, _ = _setup(tmp_path / 'mc', mc=4)
    fast = _files(scripts.test_default('brats', cfg_mc, None))
    slow = _files(scripts.test_default('brats', _with_others(cfg_mc, 'slow', pipelined=False), None))
    assert len(fast) == 2 * len(vols_mc)
    for name in fast:
        assert fast[name] == slow[name], name
    # the MC files do not depend on the stream lanes either (exact statistics, masks a function of (seed, slice, pass)) ...
    one_lane = _files(scripts.test_default('brats', _with_others(cfg_mc, 'lane1', stream_lanes=1), None))
    for name in fast:
        assert fast[name] == one_lane[name], name
    # ... nor, as far as the MC sample goes, on how the slices are batched: the loader's batches of 4 (coalesce_pixels: 0) draw the masks the
    # coalesced default draws (rounds 1-5: the draw was keyed by the batch -- another sample); what may differ is float32 summation order
    # (a plan sized for another batch may pick other kernels), far below the parity tolerance
    from rcu_amd import nifti
    ctx_plain = scripts.test_default('brats', _with_others(cfg_mc, 'plain', coalesce_pixels=0), None)
    ctx_fast = scripts.test_default('brats', _with_others(cfg_mc, 'fast2'), None)
    for name in vols_mc:
        a = nifti.read(os.path.join(ctx_plain.test_dir, name + '_probabilities.nii.gz'))[0]
        b = nifti.read(os.path.join(ctx_fast.test_dir, name + '_probabilities.nii.gz'))[0]
        assert float(np.max(np.abs(a - b))) < 2e-6, name
    # thirteen batches of one slice: the loop runs Test.MAX_INFLIGHT batches ahead of the one it finishes (download slots and staging
    # buffers in rotation, subjects completed while later batches are enqueued) -- still the serial loop's bytes
    cfg_one, vols_one, _, _ = _setup(tmp_path / 'one', mc=3)
    with open(cfg_one) as f:
        text = f.read()
    assert 'batch_size: 4' in text
    with open(cfg_one, 'w') as f:
        f.write(text.replace('batch_size: 4', 'batch_size: 1'))
    cfg_one = _with_others(cfg_one, 'uncoalesced', test_dir=str(tmp_path / 'one' / 'out'), coalesce_pixels=0)     # the loader's thirteen batches as they are
    inflight = []
    inner = loops.Test._finish_batch

    def counting(self, batch_context, *args):
        inflight.append(batch_context.batch_index)
        return inner(self, batch_context, *args)

    monkeypatch.setattr(loops.Test, '_finish_batch', counting)
    fast = _files(scripts.test_default('brats', cfg_one, None))
    assert inflight == sorted(inflight) and len(inflight) == 13          # finished in order, every batch once
    slow = _files(scripts.test_default('brats', _with_others(cfg_one, 'slow', pipelined=False), None))
    assert len(fast) == 2 * len(vols_one)
    for name in fast:
        assert fast[name] == slow[name], name


ISIC_MC_YAML = """
config:
  test_name: isic_test_baseline_mc
  test_dir: {test_dir}
  model_dir: {model_dir}
  split: ''
  seed: 20
  test_at: best
  others:
    mc: 2
  test_data:
    batch_size: 1
    dataset: {dataset}
    num_workers: 1
    shuffle: false
    transform:
    - rescale:
        entries:
        - images
        - labels
        lower: 0
        upper: 1
    - permute:
        entries:
        - images
        - labels
        permutation:
        - 2
        - 0
        - 1
    - squeeze
meta:
  type: test-config
  version: 0
"""


def test_isic_default_script_mc2(tmp_path):
    """BASELINE.json configs[0]: ISIC baseline_mc, 1 x 3 x 256 x 256, T = 2, through the isic_test_default surface."""
    from PIL import Image
    from oracle import unet_oracle as uo
    from rcu_amd import management as mgt
    from rcu_amd import nifti, scripts
    params = dict(nb_classes=2, in_channels=3, depth=4, start_filters=32, dropout=0.05)
    prefix = tmp_path / 'isic_small' / 'ISIC-2017_Test_v2'
    img_dir, lab_dir = str(prefix) + '_Data', str(prefix) + '_Part1_GroundTruth'
    os.makedirs(img_dir)
    os.makedirs(lab_dir)
    rng = np.random.RandomState(4)
    ids = ['ISIC_0000010', 'ISIC_0000011']
    for id_ in ids:
        Image.fromarray(rng.randint(0, 255, (256, 256, 3)).astype(np.uint8)).save(os.path.join(img_dir, id_ + '.jpg'))
        Image.fromarray(((rng.rand(256, 256) > 0.6) * 255).astype(np.uint8)).save(os.path.join(lab_dir, id_ + '_segmentation.png'))
    st = uo.synthetic_state(20, **params)
    mf = mgt.ModelFiles(str(tmp_path / 'train'), 'isic')
    mgt.save_model(mf, 'unet', params, st)
    cfg_path = str(tmp_path / 'test_isic_baseline_mc.yaml')
    with open(cfg_path, 'w') as f:
        f.write(ISIC_MC_YAML.format(test_dir=str(tmp_path / 'out'), model_dir=mf.model_dir, dataset=str(prefix)))
    ctx = scripts.test_default('isic', cfg_path, None)
    rows = list(csv.DictReader(open(os.path.join(ctx.test_dir, 'metrics.csv'))))
    assert [r['subject'] for r in rows] == ids and all(0 <= float(r['dice']) <= 1 for r in rows)
    # the oracle's T = 2 passes under the masks the step drew: a function of (YAML seed 20, batch index, pass) -- rebuilt here from the step
    from oracle import calib_oracle as co
    from oracle import summary_oracle as so
    from rcu_amd import data as data_mod
    from rcu_amd import steps
    from rcu_amd.model import UNet
    dataset = data_mod.IsicDataset(str(prefix), data_mod.Compose([data_mod.IntensityRescale(0, 1, entries=('images', 'labels')),
                                                                  data_mod.Permute((2, 0, 1), entries=('images', 'labels')), data_mod.Squeeze()]))
    model = UNet(**params)
    model.load_state_dict(st)
    model = model.cuda()
    step = steps.McPredictStep(2, seed=20)
    sites = model.dropout_sites()
    for k, id_ in enumerate(ids):
        sample = dataset[k]
        assert sample['ids'] == id_
        x = torch.from_numpy(np.ascontiguousarray(sample['images']))[None]
        steps.set_dropout_mode(model, True)
        flat = [step._seeded_masks(model, x.cuda(), k, j) for j in (1, 2)]          # batch_size 1: batch k is image k
        steps.set_dropout_mode(model, False)
        mask_sets = [[m.view(1, -1).cpu() for m in torch.split(f, [c for _, c in sites])] for f in flat]
        ws, multi = so.mc_probabilities(lambda xx, mk: uo.unet_forward(st, xx, mk, **params), x, mask_sets)
        ref = so.multi_prediction_summary(multi)['probabilities'][0, 1].numpy()
        p = nifti.read(os.path.join(ctx.test_dir, id_ + '_probabilities.nii.gz'))[0]
        pred = nifti.read(os.path.join(ctx.test_dir, id_ + '_prediction.nii.gz'))[0]
        assert p.shape == (256, 256) and pred.shape == (256, 256) and p.dtype == np.float32
        assert np.max(np.abs(p - ref)) < 1e-4                                        # north_star tolerance; measured ~1e-7
        assert np.mean(pred == (ref > 0.5)) > 0.9999
        assert np.array_equal(pred, (p > 0.5).astype(np.uint8)) or np.mean(pred == (p > 0.5)) > 0.9999
        tp, tn, fp, fn, n = co.confusion_counts(pred, (np.squeeze(sample['labels']) > 0.5).astype(np.uint8))
        assert abs(float(rows[k]['dice']) - co.dice_from_counts(tp, fp, fn)) < 1e-12
        assert os.path.islink(os.path.join(ctx.test_dir, id_ + '.jpg'))
        assert os.path.islink(os.path.join(ctx.test_dir, id_ + '_segmentation.png'))


@pytest.mark.parametrize('coalesce', ['0', str(160 * 192 * 128)], ids=['batch-per-image', 'coalesced'])
def test_isic_many_batches_keep_their_own_labels(tmp_path, monkeypatch, coalesce):
    """Ten ISIC subjects, batch_size 1, float labels (the shipped configs rescale ``labels``): uncoalesced (``coalesce_pixels: 0``) the pipelined loop
    runs ten batches -- more than the loader's staging ring holds -- and every subject's Dice must be the Dice of ITS prediction
    against ITS label image (labels are kept on the host by PrepareSubjectStep until the batch is finished: a staging buffer reused
    too early would hand a later batch's labels to an earlier subject).  Coalesced (the default since round 6) the ten images run as
    one batch and give the same files."""
    from PIL import Image
    from oracle import calib_oracle as co
    from oracle import unet_oracle as uo
    from rcu_amd import management as mgt
    from rcu_amd import nifti, scripts
    params = dict(nb_classes=2, in_channels=3, depth=4, start_filters=8, dropout=0.05)
    prefix = tmp_path / 'isic_many' / 'ISIC-2017_Test_v2'
    img_dir, lab_dir = str(prefix) + '_Data', str(prefix) + '_Part1_GroundTruth'
    os.makedirs(img_dir)
    os.makedirs(lab_dir)
    rng = np.random.RandomState(6)
    ids = ['ISIC_00001{:02d}'.format(i) for i in range(10)]
    labels = {}
    for i, id_ in enumerate(ids):
        Image.fromarray(rng.randint(0, 255, (64, 64, 3)).astype(np.uint8)).save(os.path.join(img_dir, id_ + '.jpg'))
        lab = np.zeros((64, 64), np.uint8)
        lab[2 * i:2 * i + 8 + 5 * i, 3 * i:3 * i + 30] = 255               # a different rectangle (and area) per subject
        labels[id_] = lab
        Image.fromarray(lab).save(os.path.join(lab_dir, id_ + '_segmentation.png'))
    st = uo.synthetic_state(23, **params)
    mf = mgt.ModelFiles(str(tmp_path / 'train'), 'isic')
    mgt.save_model(mf, 'unet', params, st)
    cfg_path = str(tmp_path / 'test_isic_baseline_mc.yaml')
    with open(cfg_path, 'w') as f:
        f.write(ISIC_MC_YAML.format(test_dir=str(tmp_path / 'out'), model_dir=mf.model_dir, dataset=str(prefix)))
    cfg_path = _with_others(cfg_path, 'coalesce{}'.format(coalesce), coalesce_pixels=int(coalesce))
    ctx = scripts.test_default('isic', cfg_path, None)
    rows = {r['subject']: r for r in csv.DictReader(open(os.path.join(ctx.test_dir, 'metrics.csv')))}
    assert sorted(rows) == ids
    dices = []
    for id_ in ids:
        pred = nifti.read(os.path.join(ctx.test_dir, id_ + '_prediction.nii.gz'))[0]
        tp, tn, fp, fn, n = co.confusion_counts(pred, (labels[id_] > 127).astype(np.uint8))
        dices.append(co.dice_from_counts(tp, fp, fn))
        assert abs(float(rows[id_]['dice']) - dices[-1]) < 1e-12, id_
    assert len(set(round(d, 9) for d in dices)) > 5          # the subjects are told apart by their labels
    # The Dice counts (and the arg-max maps) above came with the batches (scripts.ConfusionOnDeviceStep, the default): the subject-level
    # evaluation -- three synchronous GPU operations per image on the loop's main thread -- gives the same metrics.csv and the same files
    from rcu_amd import evaluation as ev
    calls = []
    inner = ev.confusion_matrx
    monkeypatch.setattr(ev, 'confusion_matrx', lambda *a, **k: (calls.append(1), inner(*a, **k))[1])
    again = scripts.test_default('isic', _with_others(cfg_path, 'again'), None)
    assert not calls
    subject_level = scripts.test_default('isic', _with_others(cfg_path, 'subject_level', device_confusion=False), None)
    assert len(calls) == len(ids)
    for run in (again, subject_level):
        assert open(os.path.join(run.test_dir, 'metrics.csv'), 'rb').read() == open(os.path.join(ctx.test_dir, 'metrics.csv'), 'rb').read()
        assert _files(run) == _files(ctx)


def _confidence(ctx, name):
    from rcu_amd import nifti
    return (nifti.read(os.path.join(ctx.test_dir, name + '_confidence.nii.gz'))[0],
            nifti.read(os.path.join(ctx.test_dir, name + '_prediction.nii.gz'))[0])


def test_brats_auxiliary_feat_and_segm_scripts(tmp_path):
    """bin-dl/brats_test_auxiliary_feat.py (PostNet on the segmentation U-Net's features) and
    bin-dl/brats_test_auxiliary_segm.py (5-channel U-Net on images + segmentation) against the oracle chain."""
    from oracle import unet_oracle as uo
    from rcu_amd import data as data_mod
    from rcu_amd import management as mgt
    from rcu_amd import scripts
    cfg_path, vols, states, params = _setup(tmp_path)
    # ---- auxiliary_feat: model_dir = PostNet, others.model_dir = the segmentation network
    post_state = uo.postnet_synthetic_state(31, 32, 2)
    mf = mgt.ModelFiles(str(tmp_path / 'train_post'), 'post')
    mgt.save_model(mf, 'postnet', dict(in_channels=32, nb_classes=2), post_state, epoch=3)
    text = open(cfg_path).read()
    seg_dir = [ln.split('model_dir: ')[1] for ln in text.splitlines() if ln.startswith('  model_dir: ')][0]
    text_feat = text.replace('model_dir: ' + seg_dir, 'model_dir: ' + mf.model_dir) \
                    .replace('  others: {}\n', '  others:\n    model_dir: {}\n    test_at: best\n'.format(seg_dir)) \
                    .replace('brats_test_baseline_mc', 'brats_test_axuiliary_feat')
    cfg_feat = str(tmp_path / 'test_feat.yaml')
    open(cfg_feat, 'w').write(text_feat)
    ctx = scripts.test_auxiliary_feat('brats', cfg_feat)
    rows = {r['subject']: r for r in csv.DictReader(open(os.path.join(ctx.test_dir, 'metrics.csv')))}
    for name, (images, labels, props) in vols.items():
        x = torch.from_numpy(images).permute(0, 3, 1, 2)
        logits, feats = uo.unet_forward(states[0], x, None, return_features=True, **params)
        ref_conf = torch.softmax(uo.postnet_forward(post_state, feats), 1)[:, 1].numpy()
        conf, pred = _confidence(ctx, name)
        assert conf.dtype == np.float32 and pred.dtype == np.uint8
        assert np.max(np.abs(conf - ref_conf)) < 1e-4
        assert np.mean(pred == logits.argmax(1).numpy()) > 0.999
        assert 0 <= float(rows[name]['dice']) <= 1
    # ---- auxiliary_segm: dataset with labels [D,H,W,2] = (ground truth, segmentation to be judged), 5-channel U-Net
    params5 = dict(params, in_channels=5)
    st5 = uo.synthetic_state(41, **params5)
    mf5 = mgt.ModelFiles(str(tmp_path / 'train_seg5'), 'seg5')
    mgt.save_model(mf5, 'unet', params5, st5, epoch=1)
    rng = np.random.RandomState(9)
    vols5 = {}
    for name, (images, labels, props) in vols.items():
        judged = np.where(rng.rand(*labels.shape) < 0.1, 1 - labels, labels).astype(np.uint8)
        lab2 = np.stack([labels, judged], -1)
        data_mod.write_volume(str(tmp_path / 'ds_wpred'), name, images, lab2, props)
        vols5[name] = (images, lab2, props)
    text_segm = text.replace('model_dir: ' + seg_dir, 'model_dir: ' + mf5.model_dir) \
                    .replace(str(tmp_path / 'ds'), str(tmp_path / 'ds_wpred')) \
                    .replace('    - data\n', '    - data:\n        categories:\n        - images\n        - labels\n') \
                    .replace('        entries:\n        - images\n        permutation', '        permutation') \
                    .replace('brats_test_baseline_mc', 'brats_test_auxiliary_segm')
    assert 'categories:\n        - images\n        - labels' in text_segm
    cfg_segm = str(tmp_path / 'test_segm.yaml')
    open(cfg_segm, 'w').write(text_segm)
    ctx5 = scripts.test_auxiliary_segm('brats', cfg_segm)
    rows5 = {r['subject']: r for r in csv.DictReader(open(os.path.join(ctx5.test_dir, 'metrics.csv')))}
    from oracle import calib_oracle as co
    for name, (images, lab2, props) in vols5.items():
        x = torch.cat([torch.from_numpy(images).permute(0, 3, 1, 2), torch.from_numpy(lab2[..., 1:2]).permute(0, 3, 1, 2).float()], 1)
        ref = torch.softmax(uo.unet_forward(st5, x, None, **params5), 1).numpy()
        conf, pred = _confidence(ctx5, name)
        assert np.max(np.abs(conf - ref[:, 1])) < 1e-4
        assert np.array_equal(pred, lab2[..., 1])                  # the judged segmentation is passed through
        tp, tn, fp, fn, n = co.confusion_counts((conf > 0.5).astype(np.uint8), (lab2[..., 1] != lab2[..., 0]).astype(np.uint8))
        assert abs(float(rows5[name]['dice']) - co.dice_from_counts(tp, fp, fn)) < 0.02   # ties at 0.5 aside


def test_isic_auxiliary_scripts(tmp_path):
    """isic_test_auxiliary_feat.py and isic_test_auxiliary_segm.py; the latter reads the `<id>_prediction.nii.gz`
    files of an earlier isic_test_default run through others.prediction_dir (customdatasets.py:104-109)."""
    from PIL import Image
    from oracle import unet_oracle as uo
    from rcu_amd import management as mgt
    from rcu_amd import nifti, scripts
    params = dict(nb_classes=2, in_channels=3, depth=4, start_filters=32, dropout=0.05)
    prefix = tmp_path / 'isic_small' / 'ISIC-2017_Test_v2'
    img_dir, lab_dir = str(prefix) + '_Data', str(prefix) + '_Part1_GroundTruth'
    os.makedirs(img_dir)
    os.makedirs(lab_dir)
    rng = np.random.RandomState(6)
    ids = ['ISIC_0000020', 'ISIC_0000021']
    for id_ in ids:
        Image.fromarray(rng.randint(0, 255, (64, 96, 3)).astype(np.uint8)).save(os.path.join(img_dir, id_ + '.jpg'))
        Image.fromarray(((rng.rand(64, 96) > 0.6) * 255).astype(np.uint8)).save(os.path.join(lab_dir, id_ + '_segmentation.png'))
    st = uo.synthetic_state(20, **params)
    mf = mgt.ModelFiles(str(tmp_path / 'train'), 'isic')
    mgt.save_model(mf, 'unet', params, st)
    base = ISIC_MC_YAML.replace('  others:\n    mc: 2\n', '  others: {{}}\n')
    cfg0 = str(tmp_path / 'test_isic_baseline.yaml')
    open(cfg0, 'w').write(base.format(test_dir=str(tmp_path / 'out'), model_dir=mf.model_dir, dataset=str(prefix)))
    ctx0 = scripts.test_default('isic', cfg0, None)
    # ---- auxiliary_feat
    post_state = uo.postnet_synthetic_state(33, 32, 2)
    mfp = mgt.ModelFiles(str(tmp_path / 'train_post'), 'post')
    mgt.save_model(mfp, 'postnet', dict(in_channels=32, nb_classes=2), post_state)
    text = base.replace('  others: {{}}\n', '  others:\n    model_dir: {seg_dir}\n    test_at: best\n') \
               .replace('isic_test_baseline_mc', 'isic_test_auxiliary_feat')
    cfg1 = str(tmp_path / 'test_isic_auxiliary_feat.yaml')
    open(cfg1, 'w').write(text.format(test_dir=str(tmp_path / 'out'), model_dir=mfp.model_dir, dataset=str(prefix),
                                      seg_dir=mf.model_dir))
    ctx1 = scripts.test_auxiliary_feat('isic', cfg1)
    for id_ in ids:
        img = np.asarray(Image.open(os.path.join(img_dir, id_ + '.jpg'))).astype(np.float32)
        x = torch.from_numpy((img - img.min()) / (img.max() - img.min())).permute(2, 0, 1)[None]
        logits, feats = uo.unet_forward(st, x, None, return_features=True, **params)
        ref = torch.softmax(uo.postnet_forward(post_state, feats), 1)[0, 1].numpy()
        conf, pred = _confidence(ctx1, id_)
        assert np.max(np.abs(conf - ref)) < 1e-4
        assert np.mean(pred == logits.argmax(1)[0].numpy()) > 0.999
        assert os.path.islink(os.path.join(ctx1.test_dir, id_ + '_segmentation.png'))
    # ---- auxiliary_segm on the predictions of the first run
    params4 = dict(params, in_channels=4)
    st4 = uo.synthetic_state(44, **params4)
    mf4 = mgt.ModelFiles(str(tmp_path / 'train_aux'), 'aux')
    mgt.save_model(mf4, 'unet', params4, st4)
    text = base.replace('  others: {{}}\n', '  others:\n    prediction_dir: {pred_dir}\n') \
               .replace('isic_test_baseline_mc', 'isic_test_auxiliary_segm')
    cfg2 = str(tmp_path / 'test_isic_auxiliary_segm.yaml')
    open(cfg2, 'w').write(text.format(test_dir=str(tmp_path / 'out'), model_dir=mf4.model_dir, dataset=str(prefix),
                                      pred_dir=ctx0.test_dir))
    ctx2 = scripts.test_auxiliary_segm('isic', cfg2)
    rows = {r['subject']: r for r in csv.DictReader(open(os.path.join(ctx2.test_dir, 'metrics.csv')))}
    for id_ in ids:
        img = np.asarray(Image.open(os.path.join(img_dir, id_ + '.jpg'))).astype(np.float32)
        judged = nifti.read(os.path.join(ctx0.test_dir, id_ + '_prediction.nii.gz'))[0]
        x = torch.cat([torch.from_numpy((img - img.min()) / (img.max() - img.min())).permute(2, 0, 1),
                       torch.from_numpy(judged.astype(np.float32))[None]])[None]
        ref = torch.softmax(uo.unet_forward(st4, x, None, **params4), 1)[0, 1].numpy()
        conf, pred = _confidence(ctx2, id_)
        assert np.max(np.abs(conf - ref)) < 1e-4
        assert np.array_equal(pred, judged)
        assert 0 <= float(rows[id_]['dice']) <= 1
        for link in (id_ + '_segmentation.png', id_ + '.jpg'):
            assert os.path.islink(os.path.join(ctx2.test_dir, link))
    with pytest.raises(ValueError):
        scripts.test_auxiliary_segm('isic', cfg0)      # others.prediction_dir is required
    # ---- evaluation driver on the ISIC outputs: png ground truth, all pixels (no mask), confidence entries
    from oracle import calib_oracle as co
    runs = {'baseline': ctx0.test_dir, 'auxiliary_feat': ctx1.test_dir, 'auxiliary_segm': ctx2.test_dir}
    scripts.eval_uncertainty('isic', runs, str(prefix), str(tmp_path / 'eval'), expected_subjects=ids)
    rows = list(csv.DictReader(open(str(tmp_path / 'eval' / 'ece' / 'eval_ece_baseline.csv'))))
    assert [r['subject_name'] for r in rows] == ids
    for r in rows:
        p = nifti.read(os.path.join(ctx0.test_dir, r['subject_name'] + '_probabilities.nii.gz'))[0]
        gt = (np.array(Image.open(os.path.join(lab_dir, r['subject_name'] + '_segmentation.png')).convert('L')) > 0)
        ref = co.ece_binary(np.stack([1 - p, p], -1), gt.astype(np.uint8))
        assert abs(float(r['ece']) - ref) < 1e-9
    for run in ('auxiliary_feat', 'auxiliary_segm'):
        files = glob.glob(str(tmp_path / 'eval' / 'ece' / 'eval_ece_{}*.csv'.format(run)))
        assert files, run
        assert all(0 <= float(r['ece']) <= 1 for f in files for r in csv.DictReader(open(f)))
    assert len(glob.glob(str(tmp_path / 'eval' / 'uncertainty' / 'eval_uncertainty_baseline_th*.csv'))) == 11


def _gt_tree(root, vols):
    from rcu_amd import nifti
    for name, (images, labels, props) in vols.items():
        (root / 'HGG' / name).mkdir(parents=True)
        for mod, arr in (('flair', images[..., 0]), ('t1', images[..., 1]), ('t2', images[..., 2]), ('t1ce', images[..., 3]), ('seg', labels * 4)):
            nifti.write(str(root / 'HGG' / name / '{}_{}.nii.gz'.format(name, mod)), arr, props)
    return str(root)


@pytest.mark.parametrize('with_gt', [True, False], ids=['gt-tree', 'dataset-labels'])
def test_device_metrics_hook_writes_the_rows_of_the_evaluation_script(tmp_path, with_gt):
    """``others.device_metrics`` (opt-in): ECE / calibration bins / uncertainty-error counts / min-max of every subject computed while its maps
    are still in HBM must give, byte for byte, the CSV files bin-eval/eval_uncertainty.py writes from the .nii.gz files the SAME run wrote
    (the float32 round trip through NIfTI is lossless).  With the ground-truth tree the hook reads target and T2 mask from where the
    evaluation reads them; without it it takes the dataset's labels and no mask -- which is the evaluation with ece_details ''."""
    from rcu_amd import scripts
    cfg_path, vols, _, _ = _setup(tmp_path, mc=5)
    gt = _gt_tree(tmp_path / 'gt', vols)
    spec = dict(run_id='baseline_mc', gt_dir=gt) if with_gt else dict(run_id='baseline_mc')
    ctx = scripts.test_default('brats', _with_others(cfg_path, 'dm', device_metrics=spec), None)

    def all_csv(root):
        return {os.path.relpath(f, root): open(f, 'rb').read() for f in sorted(glob.glob(os.path.join(root, '**', '*.csv'), recursive=True))}

    on_device = all_csv(os.path.join(ctx.test_dir, 'eval'))
    assert len(on_device) == 14
    if with_gt:
        scripts.eval_uncertainty('brats', {'baseline_mc': ctx.test_dir}, gt, str(tmp_path / 'eval'), expected_subjects=list(vols))
        from_files = all_csv(str(tmp_path / 'eval'))
    else:      # the evaluation driver without a mask (what the script does for ISIC): same loader, ece_details ''
        from rcu_amd import evalrun
        gts = evalrun.collect_brats_ground_truth(gt)
        entry = evalrun.get_eval_data('baseline_mc', ctx.test_dir, gts, expected_subjects=list(vols))
        evalrun.evaluate_runs([entry], ['minmax', 'ece_dice', 'calib', 'bnf_ue'], str(tmp_path / 'eval'), '')
        from_files = all_csv(str(tmp_path / 'eval'))
    assert sorted(on_device) == sorted(from_files)
    for name in on_device:
        assert on_device[name] == from_files[name], name
    # a subset of the actions, given as a plain list
    ctx2 = scripts.test_default('brats', _with_others(cfg_path, 'dm2', test_dir=str(tmp_path / 'out2'), device_metrics=['bnf_ue']), None)
    assert len(all_csv(os.path.join(ctx2.test_dir, 'eval'))) == 11


def test_device_metrics_hook_on_isic_subjects(tmp_path):
    """``others.device_metrics`` where every image is a subject (ISIC: no evaluation mask, labels from the png files): the CSV files of
    bin-eval/eval_uncertainty.py --ds isic on the files of the same run, byte for byte."""
    from PIL import Image
    from oracle import unet_oracle as uo
    from rcu_amd import management as mgt
    from rcu_amd import scripts
    params = dict(nb_classes=2, in_channels=3, depth=4, start_filters=8, dropout=0.2)
    prefix = tmp_path / 'isic' / 'ISIC-2017_Test_v2'
    img_dir, lab_dir = str(prefix) + '_Data', str(prefix) + '_Part1_GroundTruth'
    os.makedirs(img_dir)
    os.makedirs(lab_dir)
    rng = np.random.RandomState(9)
    ids = ['ISIC_00003{:02d}'.format(i) for i in range(4)]
    for i, id_ in enumerate(ids):
        Image.fromarray(rng.randint(0, 255, (64, 48, 3)).astype(np.uint8)).save(os.path.join(img_dir, id_ + '.jpg'))
        lab = np.zeros((64, 48), np.uint8)
        lab[6 * i:6 * i + 24, 4:30] = 255
        Image.fromarray(lab).save(os.path.join(lab_dir, id_ + '_segmentation.png'))
    mf = mgt.ModelFiles(str(tmp_path / 'train'), 'isic')
    mgt.save_model(mf, 'unet', params, uo.synthetic_state(28, **params))
    cfg_path = str(tmp_path / 'test_isic_baseline_mc.yaml')
    with open(cfg_path, 'w') as f:
        f.write(ISIC_MC_YAML.format(test_dir=str(tmp_path / 'out'), model_dir=mf.model_dir, dataset=str(prefix)).replace('mc: 2', 'mc: 4'))
    ctx = scripts.test_default('isic', _with_others(cfg_path, 'dm', device_metrics=dict(run_id='baseline_mc', gt_dir=str(prefix))), None)

    def all_csv(root):
        return {os.path.relpath(f, root): open(f, 'rb').read() for f in sorted(glob.glob(os.path.join(root, '**', '*.csv'), recursive=True))}

    on_device = all_csv(os.path.join(ctx.test_dir, 'eval'))
    scripts.eval_uncertainty('isic', {'baseline_mc': ctx.test_dir}, str(prefix), str(tmp_path / 'eval'), expected_subjects=ids)
    from_files = all_csv(str(tmp_path / 'eval'))
    assert sorted(on_device) == sorted(from_files) and len(on_device) == 14
    for name in on_device:
        assert on_device[name] == from_files[name], name


@pytest.mark.timeout(1200)
def test_the_same_yaml_writes_the_same_files_for_any_batch_size(tmp_path):
    """VERDICT r05 next #3.  The reference's ``batch_size`` is a loader setting (config/test_brats_baseline_mc.yaml:11); a slice's MC sample must
    not depend on it.  The seeded Dropout2d masks are keyed by (seed, pass, GLOBAL slice index, site, channel) (include/rcu.h rcu_dropout_masks;
    rounds 1-5: by the batch and the position in it), the statistics are exact sums, and the loop coalesces loader batches up to one
    benchmark-sized volume by default -- so ``batch_size`` 8, 32 and 160 run the same launches and write byte-identical ``.nii.gz`` and
    ``metrics.csv`` on full-size slices (two subjects of 160 slices of 192 x 128: 320 slices = 40 / 10 / 2 loader batches).  The loader's own
    batches (``coalesce_pixels: 0``) draw the same samples too: their maps agree to float32 summation order (a plan sized for another
    batch may pick another kernel for a level)."""
    from oracle import unet_oracle as uo
    from rcu_amd import data as data_mod
    from rcu_amd import management as mgt
    from rcu_amd import nifti, scripts
    rng = np.random.RandomState(11)
    names = []
    for i in range(2):
        name = 'Brats18_B{}_1'.format(i)
        images = rng.randn(160, 192, 128, 4).astype(np.float32)
        labels = (rng.rand(160, 192, 128) < 0.3).astype(np.uint8)
        data_mod.write_volume(str(tmp_path / 'ds'), name, images, labels, nifti.ImageProperties((192, 128, 160), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))
        names.append(name)
    mf = mgt.ModelFiles(str(tmp_path / 'train'), 'm')
    mgt.save_model(mf, 'unet', PARAMS, uo.synthetic_state(20, **PARAMS), epoch=2)
    split = str(tmp_path / 'split.json')
    with open(split, 'w') as f:
        json.dump({'train': [], 'valid': [], 'test': names}, f)
    base = BRATS_MC_YAML.format(test_dir=str(tmp_path / 'out'), model_dir=mf.model_dir, split=split, dataset=str(tmp_path / 'ds')).replace('mc: 20', 'mc: 3')

    def run(tag, batch_size, **others):
        path = str(tmp_path / 'cfg_{}.yaml'.format(tag))
        with open(path, 'w') as f:
            f.write(base.replace('batch_size: 32', 'batch_size: {}'.format(batch_size)))
        ctx = scripts.test_default('brats', _with_others(path, tag, **others), None)
        return ctx, _files(ctx), open(os.path.join(ctx.test_dir, 'metrics.csv'), 'rb').read()

    _, files32, csv32 = run('b32', 32)
    assert len(files32) == 4
    for tag, bs in (('b8', 8), ('b160', 160)):
        _, files, csv_bytes = run(tag, bs)
        assert csv_bytes == csv32, tag
        assert sorted(files) == sorted(files32) and all(files[k] == files32[k] for k in files), tag
    # an explicit budget of two volumes (what an 8-GPU run sets): 320-slice steps -- other launches, the same samples
    ctx_two, _, _ = run('b32x2', 32, coalesce_pixels=2 * 160 * 192 * 128)
    ctx_plain, _, _ = run('b32plain', 32, coalesce_pixels=0)
    ctx_ref, _, _ = run('b32again', 32)
    for name in names:
        ref = nifti.read(os.path.join(ctx_ref.test_dir, name + '_probabilities.nii.gz'))[0]
        for other in (ctx_two, ctx_plain):
            got = nifti.read(os.path.join(other.test_dir, name + '_probabilities.nii.gz'))[0]
            assert float(np.max(np.abs(got - ref))) < 2e-6, name


@pytest.mark.timeout(1200)
def test_brats_scripts_on_the_references_real_slice_size(tmp_path):
    """The drop-in scripts on slices of the reference's real size, 240 x 240 (scripts/create_brats18_dataset.py:53-72 never crops): the loader, the
    coalesced steps, the MC step's pass groups and the writers on padded levels -- deterministic run and ensemble against the oracle,
    MC run against the oracle's passes under the step's seeded masks (keyed by the slices' global indices), aleatoric run's sigma map."""
    from oracle import summary_oracle as so
    from oracle import unet_oracle as uo
    from rcu_amd import nifti, scripts, steps
    from rcu_amd.model import UNet
    cfg_path, vols, states, params = _setup(tmp_path / 'det', shape=(240, 240))
    ctx = scripts.test_default('brats', cfg_path, None)
    outs = _outputs(ctx, vols)
    for name, (images, labels, props) in vols.items():
        x = torch.from_numpy(images).permute(0, 3, 1, 2)
        ref = torch.softmax(uo.unet_forward(states[0], x, None, **params), 1)[:, 1].numpy()
        p, pred, rprops = outs[name]
        assert p.shape == ref.shape == labels.shape and rprops == props
        assert np.max(np.abs(p - ref)) < 1e-4 and np.mean(pred == (ref > 0.5)) > 0.999
    # MC-dropout: T = 3 passes under the masks the step draws for the run's 13 slices (global indices 0..12: subject 0 has slices 0..5)
    cfg_mc, vols_mc, st_mc, _ = _setup(tmp_path / 'mc', mc=3, shape=(240, 240))
    ctx_mc = scripts.test_default('brats', cfg_mc, None)
    outs_mc = _outputs(ctx_mc, vols_mc)
    model = UNet(**params)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in st_mc[0].items()})
    model = model.cuda()
    step = steps.McPredictStep(3, seed=20)
    sites = model.dropout_sites()
    offset = 0
    for name in sorted(vols_mc):
        images = vols_mc[name][0]
        n = images.shape[0]
        x = torch.from_numpy(images).permute(0, 3, 1, 2).contiguous()
        steps.set_dropout_mode(model, True)
        flat = [step._seeded_masks(model, x.cuda(), offset, j) for j in (1, 2, 3)]
        steps.set_dropout_mode(model, False)
        mask_sets = [[m.view(n, -1).cpu() for m in torch.split(f, [n * c for _, c in sites])] for f in flat]
        _, multi = so.mc_probabilities(lambda xx, mk: uo.unet_forward(st_mc[0], xx, mk, **params), x, mask_sets)
        ref = so.multi_prediction_summary(multi)['probabilities'][:, 1].numpy()
        assert np.max(np.abs(outs_mc[name][0] - ref)) < 1e-4, name
        offset += n
    # ensemble of three members
    cfg_ens, vols_ens, st_ens, _ = _setup(tmp_path / 'ens', seeds=(20, 21, 22), shape=(240, 240))
    outs_ens = _outputs(scripts.test_ensemble('brats', cfg_ens), vols_ens)
    for name, (images, _, _) in vols_ens.items():
        x = torch.from_numpy(images).permute(0, 3, 1, 2)
        multi = so.ensemble_probabilities([lambda xx, m, st=st: uo.unet_forward(st, xx, m, **params) for st in st_ens], x)
        assert np.max(np.abs(outs_ens[name][0] - multi.mean(0)[:, 1].numpy())) < 1e-4
