"""NIfTI-1 codec and evaluation CSV surface -- CPU only."""
import gzip
import os
import struct

import numpy as np
import pytest

from rcu_amd import evalrun, nifti


def test_nifti_round_trip_lossless_and_geometry(tmp_path):
    rng = np.random.RandomState(0)
    props = nifti.ImageProperties((12, 10, 6), origin=(-90.5, 126.0, -72.0), spacing=(1.0, 1.5, 2.0),
                                  direction=(1, 0, 0, 0, 1, 0, 0, 0, 1))
    for dtype in (np.float32, np.uint8, np.int16, np.float64):
        a = (rng.rand(6, 10, 12) * 200).astype(dtype)
        path = str(tmp_path / 'v_{}.nii.gz'.format(np.dtype(dtype).name))
        nifti.write(path, a, props)
        b, p2 = nifti.read(path)
        assert b.dtype == dtype and np.array_equal(a, b)        # lossless (the only property the path needs)
        assert p2 == props
    # rotated direction (90 degrees about z) survives too
    rot = nifti.ImageProperties((12, 10, 6), (1, 2, 3), (0.5, 0.5, 3.0), (0, -1, 0, 1, 0, 0, 0, 0, 1))
    nifti.write(str(tmp_path / 'r.nii'), np.zeros((6, 10, 12), np.uint8), rot)
    assert nifti.read(str(tmp_path / 'r.nii'))[1] == rot
    # 2-D (ISIC) images and the pixel-type cast of sitk.ReadImage(path, sitkUInt8)
    img = rng.rand(16, 24).astype(np.float32)
    nifti.write(str(tmp_path / 'i.nii.gz'), img)
    back, p = nifti.read(str(tmp_path / 'i.nii.gz'))
    assert np.array_equal(back, img) and p.size == (24, 16)
    assert nifti.read(str(tmp_path / 'i.nii.gz'), np.uint8)[0].dtype == np.uint8


def test_nifti_header_follows_the_spec(tmp_path):
    a = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    path = str(tmp_path / 'h.nii.gz')
    nifti.write(path, a, nifti.ImageProperties((4, 3, 2), (10, 20, 30), (2, 3, 4)))
    raw = gzip.open(path, 'rb').read()
    assert struct.unpack_from('<i', raw, 0)[0] == 348 and raw[344:348] == b'n+1\x00'
    assert struct.unpack_from('<8h', raw, 40) == (3, 4, 3, 2, 1, 1, 1, 1)
    assert struct.unpack_from('<2h', raw, 70) == (16, 32)                      # float32, 32 bits
    assert struct.unpack_from('<f', raw, 108)[0] == 352.0
    assert struct.unpack_from('<4f', raw, 76)[1:] == (2.0, 3.0, 4.0)
    # ITK writes LPS geometry into NIfTI's RAS frame: x and y flip sign
    assert struct.unpack_from('<4f', raw, 280) == (-2.0, 0.0, 0.0, -10.0)
    assert struct.unpack_from('<4f', raw, 296) == (0.0, -3.0, 0.0, -20.0)
    assert struct.unpack_from('<4f', raw, 312) == (0.0, 0.0, 4.0, 30.0)
    assert np.array_equal(np.frombuffer(raw, np.float32, 24, 352), a.reshape(-1))   # x fastest
    with pytest.raises(ValueError):
        nifti.write(path, a, nifti.ImageProperties((2, 3, 4)))
    with pytest.raises(ValueError):
        nifti.write(path, a.astype(np.complex64))


def test_nifti_qform_only_header_and_zero_slope(tmp_path):
    """Files whose geometry lives in the qform only (sform_code 0) and files with scl_slope 0 (= "no scaling" in NIfTI-1, whatever
    scl_inter holds): the writer of this package stores both forms, so patch its output."""
    rng = np.random.RandomState(1)
    a = (rng.rand(5, 7, 9) * 100).astype(np.float32)
    for direction in ((1, 0, 0, 0, 1, 0, 0, 0, 1), (0, -1, 0, 1, 0, 0, 0, 0, 1), (1, 0, 0, 0, -1, 0, 0, 0, -1), (-1, 0, 0, 0, -1, 0, 0, 0, 1)):
        props = nifti.ImageProperties((9, 7, 5), origin=(-12.5, 30.0, 4.25), spacing=(0.75, 1.5, 2.0), direction=direction)
        path = str(tmp_path / 'q.nii')
        nifti.write(path, a, props)
        raw = bytearray(open(path, 'rb').read())
        assert struct.unpack_from('<2h', raw, 252) == (1, 1)
        struct.pack_into('<h', raw, 254, 0)                       # sform_code = 0: only the quaternion + offsets are left
        raw[280:328] = bytes(48)
        struct.pack_into('<2f', raw, 112, 0.0, 7.0)               # scl_slope 0, scl_inter 7: must NOT be applied
        open(path, 'wb').write(bytes(raw))
        b, p2 = nifti.read(path)
        assert np.array_equal(a, b)
        assert p2.size == props.size
        assert np.allclose(p2.origin, props.origin, atol=1e-5) and np.allclose(p2.spacing, props.spacing, atol=1e-6)
        assert np.allclose(p2.direction, props.direction, atol=1e-6), (direction, p2.direction)
    # a real scaling is still applied
    struct.pack_into('<2f', raw, 112, 2.0, 1.0)
    open(path, 'wb').write(bytes(raw))
    assert np.allclose(nifti.read(path)[0], a * 2.0 + 1.0)


def test_gzip_streams_inflate_in_one_call_like_the_gzip_module(tmp_path):
    """nifti.read inflates a .gz file member by member in single zlib calls (the evaluation loop's reader threads then run without the GIL):
    the bytes of gzip.open(...).read() for one member, several members and zero padding behind the last; a truncated stream and a wrong
    checksum raise as the gzip module raises; the reader threads of the fused loop deliver the subjects' arrays in order."""
    import gzip
    import zlib
    from rcu_amd import evalrun, nifti
    rng = np.random.default_rng(5)
    payload = rng.integers(0, 4, 300000, dtype=np.uint8).tobytes()
    one = tmp_path / 'one.bin.gz'
    one.write_bytes(gzip.compress(payload, mtime=0))
    many = tmp_path / 'many.bin.gz'
    many.write_bytes(gzip.compress(payload[:1000], mtime=0) + gzip.compress(b'', mtime=0) + gzip.compress(payload[1000:], mtime=0) + b'\0' * 7)
    for path in (one, many):
        assert nifti._file_bytes(str(path)) == gzip.open(str(path), 'rb').read() == payload
    plain = tmp_path / 'plain.bin'
    plain.write_bytes(payload[:99])
    assert nifti._file_bytes(str(plain)) == payload[:99]
    cut = tmp_path / 'cut.bin.gz'
    cut.write_bytes(one.read_bytes()[:-12])
    with pytest.raises(EOFError):
        nifti._file_bytes(str(cut))
    bad = bytearray(one.read_bytes())
    bad[-6] ^= 0x55                                        # inside the CRC-32 of the trailer
    (tmp_path / 'bad.bin.gz').write_bytes(bytes(bad))
    with pytest.raises(zlib.error):
        nifti._file_bytes(str(tmp_path / 'bad.bin.gz'))
    # the reader: one task per file, results in subject order whatever order the threads finish in
    vols = {}
    files = []
    for i in range(5):
        p = rng.random((3, 8, 6)).astype(np.float32)
        pred = (p > 0.5).astype(np.uint8)
        vols[i] = (p, pred)
        nifti.write(str(tmp_path / 'p{}.nii.gz'.format(i)), p)
        nifti.write(str(tmp_path / 'l{}.nii.gz'.format(i)), pred)
        files.append(type('SF', (), {'categories': {'misc': {'probabilities': str(tmp_path / 'p{}.nii.gz'.format(i))},
                                                    'labels': {'prediction': str(tmp_path / 'l{}.nii.gz'.format(i)),
                                                               'gt': str(tmp_path / 'l{}.nii.gz'.format(i))}}})())
    params = evalrun.Loader.Params('probabilities', need_target=True, need_prediction=True, need_t2_mask=False)
    reader = evalrun._ReadAhead(files, params, depth=3, threads=3)
    try:
        for i in range(5):
            got = reader.get(i)
            assert np.array_equal(got['probabilities'], vols[i][0]) and np.array_equal(got['prediction'], vols[i][1])
            assert np.array_equal(got['target'], vols[i][1]) and got['target'].dtype == np.uint8 and got['_read_s'] >= 0.0
    finally:
        reader.close()


def test_write_subject_files(tmp_path):
    rng = np.random.RandomState(1)
    p = rng.rand(4, 8, 8, 2).astype(np.float32)
    sigma = rng.rand(4, 8, 8, 2).astype(np.float32)
    nifti.write_subject(str(tmp_path), 'subj', p, None, sigma)
    nifti.join_all()
    fg = nifti.read(str(tmp_path / 'subj_probabilities.nii.gz'))[0]
    pred = nifti.read(str(tmp_path / 'subj_prediction.nii.gz'))[0]
    sg = nifti.read(str(tmp_path / 'subj_sigma.nii.gz'))[0]
    assert np.array_equal(fg, p[..., 1]) and pred.dtype == np.uint8
    assert np.array_equal(pred, np.argmax(p, -1))
    assert np.array_equal(sg, np.where(pred == 1, sigma[..., 1], sigma[..., 0]))


def test_eval_csv_hooks_match_reference_files(golden, tmp_path):
    from oracle import calib_oracle as co
    g = golden('g12_eval_csv')
    h1 = evalrun.WriteCsvHook(str(tmp_path / 'a.csv'), entries=('ece', 'dice', 'tp', 'tn', 'fp', 'fn', 'n'))
    h2 = evalrun.WriteCsvHook(str(tmp_path / 'b.csv'), None)
    h3 = evalrun.WriteBinsCsvHook(str(tmp_path / 'c.csv'))
    h4 = evalrun.WriteSummaryCsvHook(str(tmp_path / 'd.csv'), confidence_entry='sigma')
    composed = evalrun.ReducedComposeEvalHook([h3])
    for i, sub in enumerate(g['subjects']):
        p, t = g['p_{}'.format(i)], g['t_{}'.format(i)]
        bins = {}
        ece = co.ece_binary(np.stack([1 - p, p], -1), t, out_bins=bins)   # oracle here: the hooks are what is tested
        h1.on_subject({'ece': ece, 'dice': 0.5 + 0.1 * i, 'tp': 10 + i, 'tn': 200 - i, 'fp': 3 * i, 'fn': 7, 'n': 256,
                       'extra': 'ignored'}, str(sub), 'baseline_mc')
        h2.on_subject({'tpu': 3 + i, 'values': np.array([0.25, 0.5 * i]), 'flag': bool(i % 2),
                       'twelve': list(range(12))}, str(sub), 'baseline_mc')
        res3 = dict(bins)
        res3['ece'] = ece
        res3['dice'] = 0.25 * i
        composed.on_subject(res3, str(sub), 'baseline_mc')
    for h in (h1, h2):
        h.on_run_end({}, 'baseline_mc')
    composed.on_run_end({}, 'baseline_mc')
    composed.on_run_start('baseline_mc')     # not overridden by the member: a no-op
    h4.on_run_end({'min': list(g['hist_min']), 'max': list(g['hist_max'])}, 'aleatoric')
    for name in 'abcd':
        with open(str(tmp_path / (name + '.csv')), newline='') as f:
            assert f.read() == str(g['csv_' + name]), name
    assert evalrun.read_min_max(str(tmp_path / 'd.csv')) == (float(g['hist_min'].min()), float(g['hist_max'].max()))
    assert [evalrun.ECE_FOREGROUND_NAME, evalrun.ECE_NAME, evalrun.CALIB_NAME, evalrun.UNCERTAINTY_NAME,
            evalrun.MINMAX_NAME, evalrun.CALIBRATION_PLACEHOLDER, evalrun.UNCERTAINTY_PLACEHOLDER,
            evalrun.ECE_PLACEHOLDER, evalrun.MINMAX_PLACEHOLDER] == list(g['names'])


def test_prediction_collection(tmp_path):
    run = tmp_path / 'run'
    gt = tmp_path / 'gt' / 'HGG'
    for sub in ('Brats18_X_1', 'Brats18_Y_1'):
        (gt / sub).mkdir(parents=True)
        for mod in ('flair', 't1', 't2', 't1ce', 'seg'):
            nifti.write(str(gt / sub / '{}_{}.nii.gz'.format(sub, mod)), np.zeros((2, 4, 4), np.uint8))
        run.mkdir(exist_ok=True)
        for pf in ('prediction', 'probabilities'):
            nifti.write(str(run / '{}_{}.nii.gz'.format(sub, pf)), np.zeros((2, 4, 4), np.uint8))
    gts = evalrun.collect_brats_ground_truth(str(tmp_path / 'gt'))
    entry = evalrun.get_eval_data('baseline_mc', str(run), gts, expected_subjects=['Brats18_X_1', 'Brats18_Y_1'])
    assert [sf.subject for sf in entry.subject_files] == ['Brats18_X_1', 'Brats18_Y_1']
    sf = entry.subject_files[0]
    assert set(sf.categories) == {'labels', 'misc', 'images'} and set(sf.categories['labels']) == {'prediction', 'gt'}
    os.remove(str(run / 'Brats18_Y_1_prediction.nii.gz'))
    with pytest.raises(AssertionError):
        evalrun.collect_predictions(str(run), ['prediction', 'probabilities'], ['labels', 'misc'])


def test_fused_evaluation_fans_the_metrics_out_into_the_reference_rows(tmp_path):
    """The fused subject loop of bin-eval/eval_uncertainty.py (evalrun._evaluate_fused, the device-metrics hook) computes every per-voxel scan
    once per batch and `evalrun.record_subject` turns the raw numbers -- min / max, reliability histogram, 8 x 11 counts -- into the rows
    the per-action strategies of the reference-ordered loop write.  Checked here without a GPU: the raw numbers come from the oracle, the
    CSV files must hold the reference's columns in the reference's order and the oracle's values."""
    import csv
    from oracle import calib_oracle as co
    rng = np.random.RandomState(12)
    subjects = {}
    for name in ('Brats18_B_1', 'Brats18_A_1'):
        p = rng.rand(4, 10, 12).astype(np.float32)
        p[0, 0, :3] = [0.0, 1.0, 0.5]
        subjects[name] = (p, (p > 0.5).astype(np.uint8), (rng.rand(4, 10, 12) < 0.4).astype(np.uint8), rng.rand(4, 10, 12) > 0.3)
    base = str(tmp_path / 'eval')
    actions = evalrun.get_actions(['minmax', 'ece_dice', 'calib', 'bnf_ue'], os.path.join(base, evalrun.MINMAX_NAME), base, 'foreground')
    entry = evalrun.EvalData('baseline_mc', str(tmp_path), 'probabilities')
    for a in actions:
        a.setup_eval(entry)
    assert evalrun._fusable(entry, actions) and not evalrun._fusable(evalrun.EvalData('aleatoric', str(tmp_path), 'sigma'), actions)
    want, thresholds, want_mask = evalrun.metrics_wanted(actions)
    assert want == ['ece', 'minmax', 'ue'] and thresholds == co.UE_THRESHOLDS and want_mask
    assert evalrun.metrics_wanted(actions[:1]) == (['minmax'], (0.5,), False)
    for a in actions:
        a.start_eval()
    for name in sorted(subjects):
        p, pred, tgt, mask = subjects[name]
        hist = co.calibration_histogram(*co.select_foreground(np.stack([1 - p, p], -1), tgt, mask))
        unc = co.normalised_entropy(co.add_background_probability(p))
        counts = np.array([co.uncertainty_counts(pred.astype(bool), tgt.astype(bool), unc > t) for t in co.UE_THRESHOLDS], dtype=np.int64)
        res = {'min': np.array([p.min()]), 'max': np.array([p.max()]), 'hist': tuple(np.asarray(h)[None] for h in hist), 'counts': counts[None]}
        evalrun.record_subject(actions, name, res, 0, tgt.ndim)
    for a in actions:
        a.finish_eval()

    def rows(*parts):
        with open(os.path.join(base, *parts), newline='') as f:
            return list(csv.reader(f))

    ece = rows(evalrun.ECE_FOREGROUND_NAME, evalrun.ECE_PLACEHOLDER.format('baseline_mc'))
    assert ece[0] == ['test_id', 'subject_name', 'ece', 'dice', 'tp', 'tn', 'fp', 'fn', 'n'] and [r[1] for r in ece[1:]] == sorted(subjects)
    for r in ece[1:]:
        p, pred, tgt, mask = subjects[r[1]]
        assert abs(float(r[2]) - co.ece_binary(np.stack([1 - p, p], -1), tgt, mask=mask)) < 1e-15
        tp, tn, fp, fn, n = co.confusion_counts(pred, tgt)
        assert [int(v) for v in r[4:]] == [tp, tn, fp, fn, n] and abs(float(r[3]) - co.dice_from_counts(tp, fp, fn)) < 1e-15
    cal = rows(evalrun.CALIB_NAME, evalrun.CALIBRATION_PLACEHOLDER.format('baseline_mc'))
    assert cal[0][:2] == ['test_id', 'subject_name'] and cal[0][2] == 'bins_count_00' and cal[0][-2:] == ['ece', 'dice'] and len(cal[0]) == 2 + 4 * 10 + 2
    mm = evalrun.read_min_max(os.path.join(base, evalrun.MINMAX_NAME, evalrun.MINMAX_PLACEHOLDER.format('baseline_mc')))
    assert mm == (0.0, 1.0)
    for t in co.UE_THRESHOLDS:
        ue = rows(evalrun.UNCERTAINTY_NAME, evalrun.UNCERTAINTY_PLACEHOLDER.format('baseline_mc', '{:.2f}'.format(t).replace('.', '')))
        assert ue[0][:10] == ['test_id', 'subject_name', 'tpu', 'tnu', 'fpu', 'fnu', 'tp', 'tn', 'fp', 'fn']
        for r in ue[1:]:
            p, pred, tgt, _ = subjects[r[1]]
            unc = co.normalised_entropy(co.add_background_probability(p))
            ref = co.correction_metrics(co.uncertainty_counts(pred.astype(bool), tgt.astype(bool), unc > t))
            got = dict(zip(ue[0], r))
            assert all(int(got[k]) == ref[k] for k in ('tpu', 'tnu', 'fpu', 'fnu', 'tp', 'tn', 'fp', 'fn'))
            assert abs(float(got['corrected_dice']) - ref['corrected_dice']) < 1e-15 and got['dice_benefit'] == str(ref['dice_benefit'])
    # a probability outside [0, 1] is rejected as the reference's preparation rejects it (rechun/eval/helper.py:31-47)
    bad = dict(res, max=np.array([np.float32(1.5)]))
    with pytest.raises(ValueError, match='larger than 1'):
        evalrun.record_subject(actions, 'x', bad, 0, 3)
