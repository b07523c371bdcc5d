"""Config / model-directory / loop / dataset surface of the test scripts -- CPU only."""
import os

import numpy as np
import pytest
import torch

from rcu_amd import config as cfg
from rcu_amd import data as data_mod
from rcu_amd import loops
from rcu_amd import management as mgt

# same schema and idioms as the reference's config/test_brats_baseline_mc.yaml (bare-string, {type: params} and
# parameter-less entries, free-form `others`, meta section)
BRATS_MC_YAML = """
config:
  test_name: brats_test_baseline_mc
  test_dir: {test_dir}
  model_dir: {model_dir}
  split: '{split}'
  seed: 20
  test_at: best
  others:
    mc: 20
  test_data:
    batch_size: 32
    dataset: {dataset}
    direct_extractor:
    - names
    - data:
        categories:
        - labels
    - files
    - properties
    - subject
    direct_transform:
    - squeeze:
        entries:
        - labels
    extractor:
    - indexing:
        do_pickle: true
    - shape
    - data
    indexing:
      slice: {{}}
    num_workers: 0
    shuffle: false
    transform:
    - permute:
        entries:
        - images
        permutation:
        - 2
        - 0
        - 1
    - squeeze:
        entries:
        - images
meta:
  type: test-config
  version: 0
"""


def _write_cfg(tmp_path, **kw):
    path = str(tmp_path / 'test_brats_baseline_mc.yaml')
    defaults = dict(test_dir=str(tmp_path / 'out'), model_dir=str(tmp_path / 'train' / 'model_x'), split='',
                    dataset=str(tmp_path / 'ds'))
    defaults.update(kw)
    with open(path, 'w') as f:
        f.write(BRATS_MC_YAML.format(**defaults))
    return path


def test_yaml_config_loads_and_round_trips(tmp_path):
    c = cfg.load(_write_cfg(tmp_path))
    assert c.test_name == 'brats_test_baseline_mc' and c.seed == 20 and c.test_at == 'best'
    assert c.others.mc == 20 and hasattr(c.others, 'mc') and not hasattr(c.others, 'model_dir')
    td = c.test_data
    assert td.batch_size == 32 and td.shuffle is False and td.num_workers == 0
    assert td.indexing.type == 'slice' and td.indexing.params == {}
    assert [p.type for p in td.extractor] == ['indexing', 'shape', 'data'] and td.extractor[0].params == {'do_pickle': True}
    assert [p.type for p in td.direct_extractor] == ['names', 'data', 'files', 'properties', 'subject']
    assert td.transform[0].type == 'permute' and td.transform[0].params['permutation'] == [2, 0, 1]
    out = str(tmp_path / 'copy' / 'config.yaml')
    cfg.save(out, c)
    c2 = cfg.load(out)
    assert c2.to_dict() == c.to_dict()
    assert c2.to_dict()['test_data']['direct_extractor'][0] == 'names'      # parameter-less entries stay bare strings
    cfg.save(str(tmp_path / 'c.json'), c)
    assert cfg.load(str(tmp_path / 'c.json')).to_dict() == c.to_dict()
    with open(str(tmp_path / 'train.yaml'), 'w') as f:
        f.write('config: {}\nmeta: {type: train-config, version: 0}\n')
    with pytest.raises(ValueError):
        cfg.load(str(tmp_path / 'train.yaml'))


def test_model_directory_and_checkpoint_resolution(tmp_path):
    mf = mgt.ModelFiles.from_model_dir(str(tmp_path / 'train' / 'model_190101-120000') + '/')
    assert mf.identifier == '190101-120000' and mf.model_path().endswith('model_190101-120000/model.json')
    assert os.path.basename(mf.build_checkpoint_path(7)) == 'checkpoint_ep007.pth'
    assert os.path.basename(mf.build_checkpoint_path(7, is_best=True)) == 'checkpoint_ep007-best.pth'
    os.makedirs(mf.weight_checkpoint_dir)
    for name in ('checkpoint_ep003.pth', 'checkpoint_ep012.pth', 'checkpoint_ep009-best.pth'):
        open(os.path.join(mf.weight_checkpoint_dir, name), 'w').close()
    d = mf.weight_checkpoint_dir
    assert os.path.basename(mgt.find_checkpoint_file(d, 'best')) == 'checkpoint_ep009-best.pth'
    assert os.path.basename(mgt.find_checkpoint_file(d, 'last')) == 'checkpoint_ep012.pth'
    assert os.path.basename(mgt.find_checkpoint_file(d, 3)) == 'checkpoint_ep003.pth'
    assert mgt.find_checkpoint_file(d, 5) is None
    with pytest.raises(ValueError):
        mgt.find_checkpoint_file(d, 'newest')
    with pytest.raises(AttributeError):
        mgt.find_checkpoint_file(d, 1.5)
    # model.json + checkpoint round trip through the HIP-backed UNet mirror (state_dict keys are the reference's)
    params = dict(nb_classes=2, in_channels=4, depth=4, start_filters=4, dropout=0.05)
    from rcu_amd.model import UNet
    src = UNet(**params)
    mf2 = mgt.ModelFiles(str(tmp_path / 'train2'), 'm')
    ckpt = mgt.save_model(mf2, 'unet', params, {'module.' + k: v for k, v in src.state_dict().items()}, epoch=4)
    assert ckpt == mgt.find_checkpoint_file(mf2.weight_checkpoint_dir, 'best')
    model = mgt.load_model_from_parameters(mf2.model_path())
    rest = mgt.load_checkpoint(ckpt, model)
    assert rest['epoch'] == 4 and 'state_dict' not in rest
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), src.state_dict().values()))
    with pytest.raises(ValueError):
        mgt.load_checkpoint(str(tmp_path / 'missing.pth'), model)


def _make_dataset(root, n_subjects=2, depth=5):
    rng = np.random.RandomState(0)
    vols = {}
    for i in range(n_subjects):
        name = 'Brats18_S{}_1'.format(i)
        images = rng.randn(depth + i, 16, 16, 4).astype(np.float32)
        labels = (rng.rand(depth + i, 16, 16) < 0.2).astype(np.uint8)
        data_mod.write_volume(str(root), name, images, labels)
        vols[name] = (images, labels)
    return vols


def test_volume_dataset_slices_and_assembler(tmp_path):
    vols = _make_dataset(tmp_path / 'ds')
    c = cfg.load(_write_cfg(tmp_path))
    c.test_data.batch_size = 4
    built = data_mod.BuildData(data_mod.BuildVolumeDataset())(c.test_data)
    assert len(built.dataset) == 5 + 6 and built.nb_batches == 3
    asm = loops.SubjectAssembler()
    ready_order = []
    for b, batch in enumerate(built.loader):
        assert batch['images'].shape[1:] == (4, 16, 16) and batch['images'].dtype == torch.float32   # permuted to C,H,W
        out = {'probabilities': batch['images'].permute(0, 2, 3, 1).numpy()}                           # channel-last
        asm.add_batch(out, batch, last_batch=b == built.nb_batches - 1)
        for si in sorted(asm.subjects_ready):
            vol = asm.get_assembled_subject(si)
            info = built.dataset.direct_extract(si)
            ready_order.append(info['subject'])
            assert np.array_equal(vol['probabilities'], vols[info['subject']][0])
            assert np.array_equal(info['labels'], vols[info['subject']][1])
            assert info['properties'].size == (16, 16, vol['probabilities'].shape[0])
    assert ready_order == ['Brats18_S0_1', 'Brats18_S1_1'] and not asm.volumes
    sub = data_mod.VolumeDataset(str(tmp_path / 'ds'), subject_subset=['Brats18_S1_1'])
    assert sub.subjects == ['Brats18_S1_1'] and len(sub) == 6
    with pytest.raises(ValueError):
        data_mod.VolumeDataset(str(tmp_path / 'brats18_test_reduced_norm.h5'))


def test_isic_folder_dataset(tmp_path):
    from PIL import Image
    img_dir, lab_dir = tmp_path / 'ISIC-2017_Test_v2_Data', tmp_path / 'ISIC-2017_Test_v2_Part1_GroundTruth'
    img_dir.mkdir()
    lab_dir.mkdir()
    rng = np.random.RandomState(1)
    for k in (3, 1):
        id_ = 'ISIC_000000{}'.format(k)
        Image.fromarray(rng.randint(0, 255, (32, 48, 3)).astype(np.uint8)).save(str(img_dir / (id_ + '.jpg')))
        Image.fromarray(((rng.rand(32, 48) > 0.5) * 255).astype(np.uint8)).save(str(lab_dir / (id_ + '_segmentation.png')))
    tf = data_mod.get_transform([cfg.Parameter('rescale', entries=['images', 'labels'], lower=0, upper=1),
                                 cfg.Parameter('permute', entries=['images', 'labels'], permutation=[2, 0, 1]),
                                 cfg.Parameter('squeeze')])
    ds = data_mod.IsicDataset(str(tmp_path / 'ISIC-2017_Test_v2'), tf)
    assert ds.ids == ['ISIC_0000001', 'ISIC_0000003']
    s = ds[0]
    assert s['images'].shape == (3, 32, 48) and s['labels'].shape == (32, 48)
    assert 0.0 <= s['images'].min() and s['images'].max() <= 1.0 and set(np.unique(s['labels'])) <= {0.0, 1.0}
    asm = loops.Subject2dAssembler()
    batch = data_mod.CollateDict()([ds[0], ds[1]])
    asm.add_batch({'probabilities': np.zeros((2, 32, 48, 2), np.float32)}, batch)
    assert asm.subjects_ready == {'ISIC_0000001', 'ISIC_0000003'}
    assert asm.get_assembled_subject('ISIC_0000003')['probabilities'].shape == (32, 48, 2)
    with pytest.raises(ValueError):
        data_mod.get_transform(cfg.Parameter('relabel'))


def test_hook_composition_and_metrics_csv(tmp_path):
    calls = []

    class A(loops.TestLoopHook):
        def on_test_start(self, task_context, context):
            calls.append('A.start')

    class B(loops.TestLoopHook):
        def on_test_start(self, task_context, context):
            calls.append('B.start')

        def on_termination(self, context):
            calls.append('B.term')

    h = loops.ReducedComposeTestLoopHook([A(), B()])
    h.on_test_start(None, None)
    h.on_termination(None)
    h.on_startup()
    assert calls == ['A.start', 'B.start', 'B.term']
    ctx = loops.TorchTestContext('cpu')
    ctx.test_dir = str(tmp_path)
    tc = loops.TaskContext(0, None, None)
    tc.history = loops.History()
    w = loops.WriteTestMetricsCsvHook('metrics.csv')
    w.on_test_start(tc, ctx)
    for name, dice in (('s1', 0.5), ('s2', 0.75)):
        sc = loops.SubjectContext(0, {'subject': name})
        sc.metrics = {'dice': dice, 'acc': 1.0}
        tc.history.add(sc.metrics, 'subject_metrics')
        w.on_test_subject_end(sc, tc, ctx)
    w.on_test_end(tc, ctx)
    assert open(str(tmp_path / 'metrics.csv')).read().splitlines() == ['subject,acc,dice', 's1,1.0,0.5', 's2,1.0,0.75']


def test_auxiliary_dataset_variants_and_wrappers(tmp_path):
    """Data side of the auxiliary runs: per-slice labels when the `data` extractor lists them
    (config/test_brats_auxiliary_segm.yaml:25-29), ISIC labels extended by the earlier prediction
    (rechun/dl/customdatasets.py:64-69); every reference test script has a same-named wrapper; the
    model registry knows 'postnet' (common/model/factory.py:12-15)."""
    import subprocess
    import sys
    from PIL import Image
    from rcu_amd import model as model_mod
    from rcu_amd import nifti
    rng = np.random.RandomState(2)
    images = rng.randn(3, 8, 8, 4).astype(np.float32)
    labels = (rng.rand(3, 8, 8, 2) < 0.5).astype(np.uint8)
    data_mod.write_volume(str(tmp_path / 'ds2'), 'S1', images, labels)
    c = cfg.load(_write_cfg(tmp_path))
    c.test_data.dataset = str(tmp_path / 'ds2')
    assert len(data_mod.BuildVolumeDataset()(c.test_data)[0]) and 'labels' not in data_mod.BuildVolumeDataset()(c.test_data)[0]
    c.test_data.extractor = [cfg.Parameter('shape'), cfg.Parameter('data', categories=['images', 'labels'])]
    c.test_data.transform = [cfg.Parameter('permute', permutation=[2, 0, 1])]
    sample = data_mod.BuildVolumeDataset()(c.test_data)[1]
    assert sample['images'].shape == (4, 8, 8) and sample['labels'].shape == (2, 8, 8)
    assert np.array_equal(sample['labels'], labels[1].transpose(2, 0, 1))
    # ISIC + prediction_dir
    prefix = tmp_path / 'ISIC-2017_Test_v2'
    os.makedirs(str(prefix) + '_Data')
    os.makedirs(str(prefix) + '_Part1_GroundTruth')
    os.makedirs(str(tmp_path / 'pred'))
    Image.fromarray(rng.randint(0, 255, (6, 9, 3)).astype(np.uint8)).save(str(prefix) + '_Data/ISIC_0000001.jpg')
    Image.fromarray(((rng.rand(6, 9) > 0.5) * 255).astype(np.uint8)).save(
        str(prefix) + '_Part1_GroundTruth/ISIC_0000001_segmentation.png')
    with pytest.raises(ValueError):
        data_mod.IsicDataset(str(prefix), prediction_dir=str(tmp_path / 'pred'))      # prediction missing
    judged = (rng.rand(6, 9) > 0.5).astype(np.uint8)
    nifti.write(str(tmp_path / 'pred' / 'ISIC_0000001_prediction.nii.gz'), judged)
    ds = data_mod.IsicDataset(str(prefix), prediction_dir=str(tmp_path / 'pred'))
    s = ds[0]
    assert s['labels'].shape == (6, 9, 2) and s['labels'].dtype == np.uint8
    assert np.array_equal(s['labels'][..., 1], judged * 255)
    # wrappers and registry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for ds_name in ('brats', 'isic'):
        for kind in ('default', 'ensemble', 'aleatoric', 'auxiliary_feat', 'auxiliary_segm'):
            path = os.path.join(root, 'bin-dl', '{}_test_{}.py'.format(ds_name, kind))
            assert os.path.exists(path), path
    r = subprocess.run([sys.executable, os.path.join(root, 'bin-dl', 'brats_test_auxiliary_feat.py'), '-h'],
                       capture_output=True, text=True)
    assert r.returncode == 0 and '-config_file' in r.stdout
    assert set(model_mod.model_registry) == {'unet', 'postnet'}
    post = model_mod.PostNet(32, 2)
    assert sorted(k for k in post.state_dict() if not k.endswith('num_batches_tracked'))[:3] == [
        'conv_logits.bias', 'conv_logits.weight', 'convs.0.conv2d_batch_relu.bn.bias']
    with pytest.raises(RuntimeError):
        post(torch.zeros(1, 32, 4, 4))          # CPU tensor: no fallback


def test_directories_mirror_and_eval_command_line(tmp_path, monkeypatch):
    """rcu_amd.directories keeps the reference's names (rechun/directories.py) and takes its "required to be set" entries from the
    environment; bin-eval/eval_uncertainty.py accepts the reference's command line (--ds --ids --act only)."""
    import importlib
    import subprocess
    import sys
    monkeypatch.setenv('RCU_BRATS_ORIG_DATA_DIR', str(tmp_path / 'Brats18' / 'Training'))
    monkeypatch.setenv('RCU_BRATS_BASELINE_MC_PREDICT', '190101-120000_brats_baseline_mc')
    monkeypatch.setenv('RCU_PROJECT_DIR', str(tmp_path))
    from rcu_amd import directories as dirs
    dirs = importlib.reload(dirs)
    try:
        assert dirs.BRATS_ORIG_DATA_DIR == str(tmp_path / 'Brats18' / 'Training')
        assert dirs.prediction_dir('brats', 'baseline_mc') == str(tmp_path / 'out' / 'predictions' / 'brats' / '190101-120000_brats_baseline_mc')
        assert dirs.prediction_dir('isic', 'auxiliary_feat') == str(tmp_path / 'out' / 'predictions' / 'isic' / 'auxiliary_feat')
        assert dirs.ground_truth_dir('isic') == str(tmp_path / 'in' / 'datasets' / 'isic_small' / 'ISIC-2017_Test_v2')
        assert dirs.eval_dir('brats') == str(tmp_path / 'out' / 'eval' / 'brats')
        assert dirs.UNCERTAINTY_PLACEHOLDER.format('baseline', '005') == 'eval_uncertainty_baseline_th005.csv'
        assert dirs.SPLITS_DIR == str(tmp_path / 'config' / 'splits')
    finally:
        monkeypatch.undo()
        importlib.reload(dirs)
    # the reference's command line, with no directory configured: the script names the entry to set and exits non-zero
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith('RCU_')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bin-eval', 'eval_uncertainty.py'), '--ds', 'brats', '--ids', 'baseline_mc',
                        '--act', 'minmax'], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and 'RCU_BRATS_ORIG_DATA_DIR' in r.stderr
    assert "to_evaluate: ['baseline_mc']" in r.stdout and "eval_actions: ['minmax']" in r.stdout


def test_coalesced_batches_and_prefetch_protocol():
    """rcu_amd.loops.coalesced merges collated loader batches up to a pixel budget (tensors concatenated, per-sample lists joined) and
    never across a change of the image shape; prefetch hands every item out with a release callback, stages only the entries it is
    told to, and its loader thread stops when the consumer stops early."""
    import threading
    import torch
    from rcu_amd import loops

    def batch(i, n, h=4, w=4):
        return {'images': torch.full((n, 2, h, w), float(i)), 'labels': torch.full((n, h, w), float(i)),
                'subject_index': [i] * n, 'slice_index': list(range(n))}

    src = [batch(0, 2), batch(1, 2), batch(2, 2), batch(3, 2, 8, 8), batch(4, 3), 'not a dict', batch(5, 1)]
    out = list(loops.coalesced(src, max_pixels=4 * 16))
    assert [b['images'].shape[0] if isinstance(b, dict) else b for b in out] == [4, 2, 2, 3, 'not a dict', 1]
    assert out[0]['subject_index'] == [0, 0, 1, 1] and out[0]['slice_index'] == [0, 1, 0, 1]
    assert torch.equal(out[0]['labels'][:, 0, 0], torch.tensor([0., 0., 1., 1.]))
    assert out[2]['images'].shape[-1] == 8
    assert [b['images'].shape[0] for b in loops.coalesced(src[:3], max_pixels=1)] == [2, 2, 2]      # a budget below one batch: as loaded
    got = []
    for item, release in loops.prefetch(iter(src[:3]), depth=1):
        got.append(int(item['images'][0, 0, 0, 0]))
        release()
    assert got == [0, 1, 2]
    before = threading.active_count()
    gen = loops.prefetch(iter([batch(i, 1) for i in range(50)]), depth=1)
    next(gen)
    gen.close()                      # the consumer stops early: the loader thread must not stay blocked on its queue
    for _ in range(100):
        if threading.active_count() <= before:
            break
        threading.Event().wait(0.02)
    assert threading.active_count() <= before

    def boom():
        yield batch(0, 1)
        raise RuntimeError('loader failed')

    with pytest.raises(RuntimeError, match='loader failed'):
        for _item, release in loops.prefetch(boom()):
            release()


def test_check_min_max_raises_or_warns_like_the_reference():
    """rechun/eval/helper.py:31-47: the maximum is checked first, then the minimum; ValueError unless only_warn (ToEntropy passes
    only_warn=True, rechun/eval/analysis.py:202)."""
    import warnings
    from rcu_amd import evaluation as ev
    ok = np.array([0.0, 0.5, 1.0], dtype=np.float32)
    ev.check_min_max(ok)
    with pytest.raises(ValueError, match='Found value larger than 1: "1.5"'):
        ev.check_min_max(np.array([-1.0, 1.5]))
    with pytest.raises(ValueError, match='Found value smaller than 0: "-0.25"'):
        ev.check_min_max(np.array([-0.25, 1.0]))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        ev.check_min_max(np.array([-1.0, 1.5]), only_warn=True)
    assert [str(x.message) for x in w] == ['Found value larger than 1: "1.5"', 'Found value smaller than 0: "-1.0"']
    with pytest.raises(ValueError, match='larger than 2'):
        ev.check_min_max(np.array([3.0]), min_=1, max_=2)


def test_block_loader_equals_the_per_slice_loader(tmp_path, monkeypatch):
    """VolumeDataset.__getitems__ (a loader batch cut out of the memory-mapped volume as blocks that go through the batched forms of
    the transforms) gives, entry for entry and bit for bit, the batch CollateDict makes of the per-slice samples -- for every transform
    the shipped YAML files use, batches that straddle subjects, and with the labels as a slice category; compressed .npz members fall
    back to np.load."""
    import torch
    from rcu_amd import data as data_mod
    rng = np.random.RandomState(0)
    root = str(tmp_path / 'ds')
    vols = {}
    for i, depth in enumerate((7, 5, 3)):
        x = rng.randn(depth, 12, 10, 4).astype(np.float32)
        lab = (rng.rand(depth, 12, 10) < 0.3).astype(np.uint8)
        vols['s{}'.format(i)] = (x, lab)
        if i == 2:      # a deflated member cannot be mapped
            os.makedirs(root, exist_ok=True)
            np.savez_compressed(os.path.join(root, 's2.npz'), images=x, labels=lab)
        else:
            data_mod.write_volume(root, 's{}'.format(i), x, lab)
    assert isinstance(data_mod._npz_member_mmap(os.path.join(root, 's0.npz'), 'images'), np.memmap)
    assert data_mod._npz_member_mmap(os.path.join(root, 's2.npz'), 'images') is None
    assert np.array_equal(data_mod._npz_member_mmap(os.path.join(root, 's1.npz'), 'images'), vols['s1'][0])

    def transform():
        return data_mod.Compose([data_mod.IntensityRescale(0, 1, entries=('images',)), data_mod.Permute((2, 0, 1), entries=('images',)),
                                 data_mod.UnSqueeze(-1, entries=('labels',)), data_mod.Permute((2, 0, 1), entries=('labels',)),
                                 data_mod.Squeeze(entries=('labels',))])

    assert transform().batchable
    for categories in (('images',), ('images', 'labels')):
        ds = data_mod.VolumeDataset(root, transform(), slice_categories=categories)
        fast = list(torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=data_mod.CollateDict()))
        ds.block_loader = False
        slow = list(torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=data_mod.CollateDict()))
        ds.block_loader = True
        assert len(fast) == len(slow) == 4                       # 15 slices: batches straddle the subjects
        for a, b in zip(fast, slow):
            assert list(a.keys()) == list(b.keys())
            for key in a:
                if torch.is_tensor(a[key]):
                    assert a[key].dtype == b[key].dtype and torch.equal(a[key], b[key]), key
                else:
                    assert list(a[key]) == list(b[key]), key
    # the image batch keeps the FILE's channel-last memory order and is handed over as the channel-first view of it (torch's channels_last
    # format): a memcpy per block on the loader thread, the re-ordering on the GPU behind the upload (steps._images_to_device)
    ds = data_mod.VolumeDataset(root, transform(), slice_categories=('images',))
    first = next(iter(torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=data_mod.CollateDict())))['images']
    assert tuple(first.shape) == (4, 4, 12, 10) and not first.is_contiguous() and first.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(first.contiguous(), torch.from_numpy(np.stack([ds[i]['images'] for i in range(4)])))
    # a transform without a batched form keeps the per-slice path
    class Odd:
        def __call__(self, sample):
            return sample
    ds = data_mod.VolumeDataset(root, data_mod.Compose([Odd()]))
    assert not ds.transform.batchable and isinstance(ds.__getitems__([0, 1]), list)
    # direct_extract serves labels and geometry from the side cache after the image volumes have moved on
    for si in range(3):
        ds[ds.index.index((si, 0))]
    assert np.array_equal(ds.direct_extract(0)['labels'], vols['s0'][1])


def test_npz_member_mapping_checks_sizes_and_collate_honours_entries(tmp_path):
    """the memory map of a stored .npz member skips np.load's CRC pass, so at least the member's size must agree with its
    header and with the file (a truncated file falls back to np.load, which fails loudly); a collate whose ``entries`` exclude 'labels'
    keeps them as per-sample lists also when the dataset handed over a batch it had collated itself."""
    from rcu_amd import data as data_mod
    path = str(tmp_path / 'vol.npz')
    arr = np.arange(4 * 6 * 5 * 3, dtype=np.float32).reshape(4, 6, 5, 3)
    np.savez(path, images=arr, labels=np.zeros((4, 6, 5), np.uint8))
    mapped = data_mod._npz_member_mmap(path, 'images')
    assert isinstance(mapped, np.memmap) and np.array_equal(mapped, arr)
    blob = open(path, 'rb').read()
    cut = str(tmp_path / 'cut.npz')
    with open(cut, 'wb') as f:
        f.write(blob[:len(blob) // 3])
    assert data_mod._npz_member_mmap(cut, 'images') is None
    pre = data_mod.PreCollated(images=torch.zeros(2, 4, 6, 5), labels=torch.ones(2, 6, 5), subject_index=[0, 0])
    both = data_mod.CollateDict()(pre)
    assert torch.is_tensor(both['labels']) and torch.is_tensor(both['images'])
    only_images = data_mod.CollateDict(entries=('images',))(pre)
    assert isinstance(only_images['labels'], list) and len(only_images['labels']) == 2 and torch.is_tensor(only_images['images'])


def test_bench_helpers_plan_fingerprint_and_host_description():
    """bench.py reports the PMC traffic figure only for the plan it was measured on (a fingerprint of the layer table and the samples per launch)
    and describes the host its CPU baseline ran on (VERDICT r03 weak points 8-9)."""
    import bench
    layers = [dict(name='a', kernel='k<1>', cin=4, cout=32, height=192, width=128, grid_height=192, grid_width=128),
              dict(name='b', kernel='k<2>', cin=32, cout=32, height=192, width=128, grid_height=192, grid_width=128)]
    f = bench.plan_fingerprint(layers, 320)
    assert f != bench.plan_fingerprint([layers[0], dict(layers[1], grid_width=160)], 320)      # a padded level is another plan
    assert f == bench.plan_fingerprint([dict(L) for L in layers], 320) and len(f) == 12
    assert f != bench.plan_fingerprint(layers, 160)
    assert f != bench.plan_fingerprint([dict(layers[0], kernel='k<3>'), layers[1]], 320)
    host = bench.host_description()
    assert host['affinity_count'] >= 1 and host['logical_cpus'] >= host['affinity_count'] and host['torch_threads'] >= 1
    assert isinstance(host['affinity'], str) and host['affinity']


def test_pass_groups_are_balanced_over_the_lanes_and_capped_by_the_2gb_tensor_bound():
    """steps.balanced_groups: rounds of one group per lane, so both lanes carry the same passes; steps.pass_group_size: GROUP_PIXELS worth of
    pixels, and no activation tensor beyond the 2 GB the Winograd kernels address (model.UNet.max_group_samples)."""
    from rcu_amd import model as model_mod
    from rcu_amd import steps
    assert steps.balanced_groups(20, 4, 2) == [4, 4, 4, 4, 2, 2]
    assert steps.balanced_groups(20, 2, 2) == [2] * 10
    assert steps.balanced_groups(20, 4, 1) == [4] * 5
    assert steps.balanced_groups(20, 7, 2) == [7, 7, 3, 3]
    assert steps.balanced_groups(3, 2, 2) == [2, 1]
    assert steps.balanced_groups(50, 2, 2) == [2] * 24 + [1, 1]
    assert steps.balanced_groups(0, 4, 2) == []
    for count in range(1, 40):
        for group in (1, 2, 3, 4, 7):
            for lanes in (1, 2, 3):
                sizes = steps.balanced_groups(count, group, lanes)
                assert sum(sizes) == count and max(sizes) <= group and min(sizes) >= 1
                loads = [sum(sizes[k::lanes]) for k in range(lanes)]
                assert max(loads) - min(loads) <= 1, (count, group, lanes, sizes)
    brats = model_mod.UNet(2, 4, 4, 32, 0.05)
    assert brats.max_group_samples(192, 128) == 682
    # the reference's uncropped BraTS slices: level 0 may be allocated up to 256 x 256 (padded levels) -- one 155-slice pass per launch
    assert brats.max_group_samples(240, 240) == ((1 << 31) - 1) // (256 * 256 * 4 * 32) == 255
    assert steps.pass_group_size(brats, 155, 240, 240, steps.McPredictStep.GROUP_PIXELS) == 1
    assert steps.pass_group_size(brats, 160, 192, 128, steps.McPredictStep.GROUP_PIXELS) == 4
    assert steps.pass_group_size(brats, 32, 192, 128, steps.McPredictStep.GROUP_PIXELS) == 20
    assert steps.pass_group_size(brats, 160, 192, 128, 0) == 1
    sigma = model_mod.UNet(2, 4, 4, 32, 0.05, sigma_out=True)
    assert steps.pass_group_size(sigma, 160, 192, 128, steps.McPredictStep.GROUP_PIXELS) == 2
    isic = model_mod.UNet(2, 3, 4, 32, 0.05)
    assert steps.pass_group_size(isic, 32, 256, 256, steps.McPredictStep.GROUP_PIXELS) == 7
    assert steps.pass_group_size(object(), 32, 256, 256, steps.McPredictStep.GROUP_PIXELS) == 7       # a foreign module: the pixel rule alone


def test_the_test_loop_runs_ahead_by_a_pixel_budget():
    """loops._finish_oldest_now: two volume-sized batches stay enqueued behind the one being finished, ten of the shipped batches of 32 slices
    (two volumes' worth of pixels), tiny batches are capped by MAX_INFLIGHT, the plain loop finishes every batch at once."""
    from rcu_amd import loops
    budget, cap = loops.Test.INFLIGHT_PIXELS, loops.Test.MAX_INFLIGHT
    volume, batch32, one = 160 * 192 * 128, 32 * 192 * 128, 32 * 32

    def steady_depth(px):
        inflight, deepest = [], 0
        for _ in range(100):
            inflight.append(px)
            while loops._finish_oldest_now(inflight, True, budget, cap):
                inflight.pop(0)
            deepest = max(deepest, len(inflight))
        return deepest

    assert budget == 2 * volume
    assert steady_depth(volume) == 2            # volume-sized batches: two enqueued while the one before them is finished
    assert steady_depth(batch32) == 10
    assert steady_depth(one) == cap
    assert not loops._finish_oldest_now([], True, budget, cap)
    assert not loops._finish_oldest_now([one], True, budget, cap)
    assert loops._finish_oldest_now([one], False, budget, cap)
    assert loops._finish_oldest_now([volume, volume, volume], True, budget, cap)


def test_round5_host_rules_seeds_worlds_and_loop_options(tmp_path, monkeypatch):
    """Host-side rules that need no GPU: mask seeds are a function of (seed, batch, pass); the padded-channel cap of a pass group; the launcher
    environment -> World; the rcu_amd keys of `others` -> Test options; steps fall back from exact sums beyond 2,048 passes."""
    from rcu_amd import distributed as rdist
    from rcu_amd import model as model_mod
    from rcu_amd import scripts, steps
    assert len({steps.job_seed(20, k, j) for k in range(40) for j in range(1, 41)}) == 1600
    assert steps.job_seed(20, 3, 4) == rdist.job_seed(20, 3, 4) != steps.job_seed(21, 3, 4)
    # ADVICE (round 4): the 2 GB cap of a pass group counts PADDED channels -- start_filters 16 pads to 32, with the sigma twin to 64
    narrow = model_mod.UNet(2, 4, 4, 16, 0.05, sigma_out=True)
    assert narrow.max_group_samples(192, 128) == ((1 << 31) - 1) // (192 * 128 * 4 * 64)
    assert model_mod.UNet(2, 4, 4, 32, 0.05).max_group_samples(192, 128) == 682
    # no launcher: a world of one, no process group
    for key in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(key, raising=False)
    world = rdist.world_from_env('cuda')
    assert (world.rank, world.world, world.is_root, world.device) == (0, 1, True, 'cuda')
    # the YAML file's `others` -> loop options
    context = loops.TorchTestContext('cpu')
    context.config = cfg.TestConfiguration()
    context.config.others = cfg.OtherParameters().from_dict(dict(mc=20, coalesce_pixels=7864320, pipelined=False, max_inflight=3, stream_lanes=1))
    assert scripts._loop_options(context) == dict(coalesce=7864320, pipelined=False, max_inflight=3, loader_timing=False)
    assert scripts._mask_seed(context, world) == context.config.seed == 20
    built = scripts._default_steps(context, world)
    assert type(built[0]) is steps.McPredictStep and (built[0].seed, built[0].lanes, built[0].exact, built[0].mc_steps) == (20, 1, True, 20)
    context.config.others = cfg.OtherParameters().from_dict({})
    # coalescing is the default (round 6: the seeded masks are keyed by the slice, not by the batch); ``coalesce_pixels: 0`` switches it off
    assert scripts._loop_options(context) == dict(coalesce=loops.Test.COALESCE_PIXELS, pipelined=None, max_inflight=None, loader_timing=False)
    assert type(scripts._default_steps(context, world)[0]) is steps.SegmentationPredictStep
    test = loops.Test([], max_inflight=1, inflight_pixels=10, coalesce=5, pipelined=False, loader_timing=True)
    assert (test.max_inflight, test.inflight_pixels, test.coalesce, test.pipelined, test.loader_timing) == (1, 10, 5, False, True)
    assert loops.Test([]).max_inflight == loops.Test.MAX_INFLIGHT
    # exact sums hold 2,048 passes; a step asked for more keeps plain float sums
    assert steps.McPredictStep(20).exact and not steps.McPredictStep(4096).exact and not steps.McPredictStep(20, exact=False).exact
    # the sharded steps need a seed (the masks of a pass must not depend on the rank that runs it)
    with pytest.raises(ValueError):
        rdist.ShardedMcPredictStep(20, rdist.World(1, 2), seed=None)
    # bench.py's volumes-per-step rule
    import bench
    assert [bench.volumes_per_step(w, 21, 4) for w in (1, 2, 4, 8)] == [1, 1, 1, 2]


def test_eval_subject_step_sums_the_counts_that_came_with_the_batches():
    """scripts.EvalSubjectStep with ``subject_data['confusion']`` (scripts.ConfusionOnDeviceStep: tp / tn / fp / fn per slice, taken on the GPU at
    batch time) -- pure host arithmetic: the rows' sum is the subject's confusion matrix, Dice as evaluation.dice computes it (0 / 0 = 1), the
    entry is consumed; a 2-D subject's arg-max map that came along is handed to the writer (and kept as int64 where the script keeps it);
    without the entry the step evaluates the assembled subject itself (needs the GPU: fails loudly here)."""
    import torch
    from rcu_amd import loops, scripts
    rng = np.random.RandomState(1)
    probabilities = rng.rand(5, 6, 4, 2).astype(np.float32)
    counts = np.array([[3, 10, 2, 1], [0, 24, 0, 0], [5, 1, 9, 9], [0, 0, 0, 24], [7, 7, 5, 5]], dtype=np.int64)
    sc = loops.SubjectContext(0, {'probabilities': probabilities, 'confusion': counts.copy(), 'labels': np.zeros((5, 6, 4), np.uint8)})
    scripts.EvalSubjectStep()(sc, None, None)
    tp, fp, fn = counts[:, 0].sum(), counts[:, 2].sum(), counts[:, 3].sum()
    assert sc.metrics == {'dice': 2 * tp / (2 * tp + fp + fn)} and 'confusion' not in sc.subject_data and 'prediction' not in sc.more
    empty = loops.SubjectContext(1, {'probabilities': probabilities, 'confusion': np.array([[0, 24, 0, 0]]), 'labels': None})
    scripts.EvalSubjectStep()(empty, None, None)
    assert empty.metrics == {'dice': 1.0}
    made = (probabilities[0, ..., 1] > probabilities[0, ..., 0]).astype(np.uint8)[..., None]          # [H, W, 1], as the loop delivers it
    two_d = loops.SubjectContext('ISIC_1', {'probabilities': probabilities[0], 'confusion': counts[0], 'prediction': made.copy(), 'labels': None})
    scripts.EvalSubjectStep(squeeze_labels=True, keep_prediction=True)(two_d, None, None)
    assert two_d.metrics == {'dice': 2 * 3 / (2 * 3 + 2 + 1)}
    assert two_d.subject_data['prediction'].dtype == np.int64 and np.array_equal(two_d.subject_data['prediction'], made[..., 0])
    cached, source = two_d.more['prediction']
    assert cached.dtype == np.uint8 and np.array_equal(cached, made[..., 0]) and source is two_d.subject_data['probabilities']
    dropped = loops.SubjectContext('ISIC_2', {'probabilities': probabilities[0], 'confusion': counts[0], 'prediction': made.copy(), 'labels': None})
    scripts.EvalSubjectStep()(dropped, None, None)
    assert 'prediction' not in dropped.subject_data and np.array_equal(dropped.more['prediction'][0], made[..., 0])
    if not torch.cuda.is_available():
        plain = loops.SubjectContext(2, {'probabilities': probabilities, 'labels': np.zeros((5, 6, 4), np.uint8)})
        with pytest.raises(RuntimeError):
            scripts.EvalSubjectStep()(plain, None, None)


def test_the_test_loop_counts_the_slices_it_hands_out_and_the_mask_key_follows():
    """Round 6: the seeded Dropout2d masks of the MC step are keyed by a slice's GLOBAL index.  ``loops.Test`` numbers the run's stream of slices
    (``BatchContext.sample_offset`` = slices handed out before the batch, whatever sizes the loader's batches have); ``steps.first_sample_of``
    falls back to batch_index x n for a hand-built context; ``steps.pass_seed`` does not depend on the batch; the sharded runner takes the offset
    the step hands it (``sample_offsets``) or step x n."""
    import types
    from rcu_amd import distributed as rdist
    from rcu_amd import loops, steps

    class Context(loops.TorchTestContext):
        def __init__(self, batches):
            super().__init__('cpu')
            self.batches = batches

        def setup_directory(self): pass
        def setup_logging(self): pass
        def get_seed(self): return None
        def load_test_data(self, build_test): self.test_data = types.SimpleNamespace(loader=self.batches, nb_batches=len(self.batches), dataset=None)
        def get_test_at(self): return 'best'
        def load_from_checkpoint(self, epoch): self.model = None

        def get_task_context(self):
            tc = loops.TaskContext(0, self.test_data, None)
            tc.history = loops.History()
            return tc

    seen = []

    class Record(steps.BatchStep):
        def __call__(self, batch_context, task_context, context):
            n = batch_context.input['images'].shape[0]
            seen.append((batch_context.batch_index, batch_context.sample_offset, steps.first_sample_of(batch_context, n)))

    sizes = [4, 4, 3, 4, 1]
    batches = [{'images': torch.zeros(n, 4, 8, 8)} for n in sizes]
    loops.Test([Record()], pipelined=False)(Context(batches), None)
    assert seen == [(0, 0, 0), (1, 4, 4), (2, 8, 8), (3, 11, 11), (4, 15, 15)]
    assert steps.first_sample_of(steps.BatchContext({}, 3), 32) == 96 and steps.first_sample_of(steps.BatchContext({}, 3, sample_offset=7), 32) == 7
    assert steps.pass_seed(20, 4) == steps.job_seed(20, 0, 4) == rdist.pass_seed(20, 4) != steps.pass_seed(20, 5)
    runner = rdist.ShardedMcRunner(None, 4, engine=object(), seed=3)
    x = torch.zeros(6, 4, 8, 8)
    assert runner.first_sample(x, 5) == 30
    runner.sample_offsets[5] = 17
    assert runner.first_sample(x, 5) == 17 and runner.first_sample(x, 6) == 36
    assert runner.ws_transport is None      # resolved at the first exchange: point to point on RCCL, inside the reduce on gloo with device tensors
