"""CPU-side checks of the boundary: the shared library loads and exports every symbol include/rcu.h
declares, argument validation works without a GPU, and the host-side mirrors keep the reference's names."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import rcu_amd.build as b
    b.build()
    from rcu_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported(lib):
    header = open(os.path.join(ROOT, 'include', 'rcu.h')).read()
    declared = set(re.findall(r'\b(rcu_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations found'
    so = lib.load()
    for name in declared:
        assert hasattr(so, name), name
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    assert b'gfx950' in so.rcu_version()


def test_thresholds_match_reference_bit_patterns(lib):
    thr = np.array(list(lib.ece_thresholds(10)), dtype=np.float32)
    assert [hex(v) for v in thr.view(np.uint32)] == ['0x3dcccccd', '0x3e4ccccd', '0x3e99999a', '0x3ecccccd',
                                                     '0x3f000001', '0x3f19999a', '0x3f333334', '0x3f4ccccd',
                                                     '0x3f666667']
    from oracle import calib_oracle as co
    for nb in (2, 5, 10, 15, 32):
        assert np.array_equal(np.array(list(lib.ece_thresholds(nb)), dtype=np.float32)[:nb - 1],
                              co.float32_thresholds(nb))


def test_argument_validation_without_gpu(lib):
    so = lib.load()
    desc = lib.UnetDesc(nb_classes=2, in_channels=4, depth=4, start_filters=32, has_dropout=1, dropout_center=-1,
                        sigma_out=0, bn=1, height=8, width=32, max_batch=1)
    handle = ctypes.c_void_p()
    assert so.rcu_unet_create(ctypes.byref(desc), ctypes.byref(handle)) == -1      # 8 < 2^4: nothing left at the bottom level
    assert b'2^depth' in so.rcu_last_error()
    desc.height, desc.nb_classes = 32, 9
    assert so.rcu_unet_create(ctypes.byref(desc), ctypes.byref(handle)) == -1
    # plan options (rcu_unet_options: what replaced the RCU_CONV_* / RCU_ACT_LAYOUT / RCU_FUSE_HEAD environment switches): defaults, and
    # values outside their sets are refused before anything touches the GPU
    opts = lib.UnetOptions()
    so.rcu_unet_default_options(ctypes.byref(opts))
    assert (opts.conv_winograd, opts.conv_winograd4, opts.conv_first, opts.act_layout, opts.fuse_head, opts.head_winograd4) == (1, 1, 1, 0, 1, 1)
    desc.nb_classes = 2
    for field, bad in (('conv_winograd4', 4), ('act_layout', 5), ('fuse_head', -1), ('head_winograd4', 2)):
        o = lib.UnetOptions()
        so.rcu_unet_default_options(ctypes.byref(o))
        setattr(o, field, bad)
        assert so.rcu_unet_create_with(ctypes.byref(desc), ctypes.byref(o), None, ctypes.byref(handle)) == -1, field
        assert b'rcu_unet_options' in so.rcu_last_error()
    assert so.rcu_unet_set_fuse_head(None, 1) == -1
    # rcu_dropout_masks: null pointers, counts and the site table are checked before anything touches the GPU
    seeds, ch, keep = (ctypes.c_uint64 * 2)(1, 2), (ctypes.c_int32 * 2)(8, 8), (ctypes.c_float * 2)(0.9, 0.9)
    assert so.rcu_dropout_masks(None, 2, 1, 0, ch, keep, 2, ctypes.c_void_p(8), None) == -1
    assert so.rcu_dropout_masks(seeds, 2, 1, 0, ch, keep, 2, None, None) == -1
    assert so.rcu_dropout_masks(seeds, 0, 1, 0, ch, keep, 2, ctypes.c_void_p(8), None) == -1
    assert so.rcu_dropout_masks(seeds, 2, 1, 0, ch, keep, 41, ctypes.c_void_p(8), None) == -1 and b'n_sites' in so.rcu_last_error()
    assert so.rcu_dropout_masks(seeds, 2, 1, 0, ch, (ctypes.c_float * 2)(0.9, 1.5), 2, ctypes.c_void_p(8), None) == -1
    assert so.rcu_dropout_masks(seeds, 2, 1 << 27, 0, ch, keep, 2, ctypes.c_void_p(8), None) == -1 and b'2^31' in so.rcu_last_error()
    assert so.rcu_calib_set_blocks_per_workgroup(-1, 0) == -1 and so.rcu_calib_set_blocks_per_workgroup(0, 0) == 0
    with pytest.raises(lib.RcuError):
        lib.check(so.rcu_mc_finalize(None, 1, 1, 2, 1, 0, None, None, None, None, None))
    assert so.rcu_mc_stats_bytes(160, 192 * 128, 2, 0) == 160 * 192 * 128 * 2 * 4
    assert so.rcu_mc_stats_bytes(160, 192 * 128, 2, 3) == 160 * 192 * 128 * 5 * 8
    assert so.rcu_ece_workspace_bytes(160 * 192 * 128, 1) > 0


def test_model_mirror_keeps_reference_surface():
    from rcu_amd import steps
    from rcu_amd.model import UNet, get_model
    m = get_model('unet', nb_classes=2, in_channels=3, depth=4, start_filters=32, dropout=0.05, sigma_out=True)
    assert isinstance(m, UNet) and len(m.state_dict()) == 152
    assert len(m.dropout_sites()) == 20 and not m.mc_active()
    steps.set_dropout_mode(m, True)
    assert m.mc_active() and all(not mod.training for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d))
    steps.set_dropout_mode(m, False)
    m.load_state_dict({'module.' + k: v for k, v in m.state_dict().items()})       # DataParallel prefix
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32))                                                # CPU tensor: no fallback
    res = UNet(2, 4, depth=2, start_filters=8, residual=True)                       # ConvResidualBlock: "<block>.residual" 1x1 convs
    assert {k for k in res.state_dict() if 'residual' in k} == {
        p + s for p in ('down_convs.0.block.residual', 'down_convs.1.block.residual', 'bottom_convs.residual',
                        'up_convs.0.block.residual', 'up_convs.1.block.residual') for s in ('.weight', '.bias')}
    assert tuple(res.state_dict()['up_convs.0.block.residual.weight'].shape) == (16, 32, 1, 1)
    center = UNet(2, 4, depth=4, start_filters=32, dropout=0.5, dropout_center=4)
    assert len(center.dropout_sites()) == 9
    nodrop = UNet(2, 4, dropout=None)
    assert nodrop.dropout_sites() == []
    # plan options and the workspace donor are validated on the host side too
    m.plan_options = {'conv_winograd5': 1}
    with pytest.raises(ValueError):
        m._options()
    with pytest.raises(ValueError):
        center.share_workspace(nodrop)            # another architecture (no Dropout2d modules: another set of dropout sites)
    twin = UNet(2, 4, depth=4, start_filters=32, dropout=0.5, dropout_center=4)
    twin.share_workspace(center)
    third = UNet(2, 4, depth=4, start_filters=32, dropout=0.5, dropout_center=4)
    third.share_workspace(twin)                   # one level: everybody borrows from the owner
    assert twin._donor is center and third._donor is center
    from rcu_amd.steps import share_member_workspaces
    share_member_workspaces([third, twin, center])          # the same members in another order: the owner stays the owner
    assert center._donor is None and twin._donor is center and third._donor is center
    with pytest.raises(ValueError):
        steps.McPredictStep(2)(steps.BatchContext({'images': torch.zeros(1, 4, 32, 32)}, 0), None, object())


def test_host_side_metric_arithmetic_matches_oracle(golden):
    """ECE-from-histogram and correction metrics are host arithmetic: check them against the oracle on the
    golden histograms without touching the GPU."""
    from oracle import calib_oracle as co
    from rcu_amd import evaluation as ev
    g = golden('g8_ece')
    cnt, sc, sp = co.calibration_histogram(*co.select_foreground(np.stack([1 - g['a_p'], g['a_p']], -1),
                                                                 g['a_target'], g['a_mask']))
    for w in ('proportion', 'log_proportion', 'power_proportion', 'mean_proportion'):
        bins = {}
        e = ev.ece_from_histogram(cnt, sc, sp.astype(np.int64), 3, bins, w)
        assert e == co.ece_from_histogram(cnt, sc, sp, w, 3)
    assert e is not None and np.array_equal(bins['bins_count'], g['a_bins_count_masked'])
    g9 = golden('g9_uncertainty')
    for row in g9['counts']:
        a, b = ev.correction_results(row), co.correction_metrics(row)
        assert set(a) == set(b)
        for k in a:
            assert a[k] == b[k] or (np.isnan(a[k]) and np.isnan(b[k])), k
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            ev.ece_binary(np.zeros((2, 2), np.float32), np.zeros((2, 2), np.uint8))  # needs the GPU: fails loudly


def test_ctypes_structs_have_the_layout_of_the_header(tmp_path):
    """The Python host mirrors include/rcu.h's structs by hand (rcu_amd/_lib.py): compile the header with gcc -- as a C translation unit, the way a
    reference-side binding would include it -- and compare size and every field offset with the ctypes classes."""
    import ctypes
    import subprocess
    from rcu_amd import _lib
    mirrors = {'rcu_unet_desc': _lib.UnetDesc, 'rcu_unet_options': _lib.UnetOptions, 'rcu_layer_info': _lib.LayerInfo, 'rcu_ece_result': _lib.EceResult}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rcu.h"', 'int main(void) {']
    for cname, cls in mirrors.items():
        lines.append('  printf("{0} size %zu\\n", sizeof({0}));'.format(cname))
        for field in cls._fields_:
            lines.append('  printf("{0} {1} %zu\\n", offsetof({0}, {1}));'.format(cname, field[0]))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines) + '\n')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    seen = {}
    for line in subprocess.check_output([str(exe)]).decode().splitlines():
        cname, field, value = line.split()
        seen[(cname, field)] = int(value)
    for cname, cls in mirrors.items():
        assert seen[(cname, 'size')] == ctypes.sizeof(cls), cname
        for field in cls._fields_:
            assert seen[(cname, field[0])] == getattr(cls, field[0]).offset, (cname, field[0])
    assert _lib.RCU_MAX_BINS * 8 * 3 == ctypes.sizeof(_lib.EceResult)


@pytest.mark.timeout(600)
def test_winograd4_kernel_owns_the_accumulator_file(tmp_path):
    """csrc/rcu_wino4.hip names its 256 accumulator registers literally in inline assembly (its header says why).  That is only
    sound while the compiler itself keeps out of the accumulator file and spills nothing: audit the generated code --
    no scratch, no VGPR spill, no v_accvgpr_* outside the asm statements, 256 AGPRs allocated by the kernel descriptor."""
    import subprocess
    csrc = os.path.join(ROOT, 'reliability-challenges-uncertainty_amd', 'csrc')
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-slp-vectorize', '--cuda-device-only', '-S',
           os.path.join(csrc, 'rcu_wino4.hip'), '-o', str(tmp_path / 'w4.s')]
    subprocess.check_call(cmd, cwd=str(tmp_path))
    text = open(str(tmp_path / 'w4.s')).read()
    kernels = re.findall(r'\.amdhsa_kernel (\S*conv_wino4_stream\S*)', text)
    assert len(kernels) >= 2
    inside, stray = False, []
    for line in text.splitlines():
        if ';;#ASMSTART' in line:
            inside = True
        elif ';;#ASMEND' in line:
            inside = False
        elif 'v_accvgpr' in line and not inside:
            stray.append(line.strip())
    assert not stray, stray[:5]
    assert 'scratch_' not in text
    meta = text[text.index('amdhsa.kernels'):]
    for block in meta.split('- .agpr_count:')[1:]:
        if 'conv_wino4_stream' not in block:
            continue
        assert int(block.split()[0]) == 256
        assert int(re.search(r'\.vgpr_spill_count:\s*(\d+)', block).group(1)) == 0
        assert int(re.search(r'\.private_segment_fixed_size:\s*(\d+)', block).group(1)) == 0


def test_uncertain_voxel_table_matches_fixture_g20():
    """The table compiled into the library (csrc/rcu_ue_table.inc) is the committed fixture g20 -- checked through the host-side entry
    points, no GPU: thresholds, support of the script's eleven thresholds and of their subsets, and the membership of every probe value."""
    import ctypes
    import numpy as np
    from conftest import load_golden
    from rcu_amd import _lib
    lib = _lib.load()
    g = load_golden('g20_ue_boundaries')
    thr = [float(t) for t in g['thresholds']]
    assert lib.rcu_unc_from_p_num_thresholds() == len(thr) == 11
    assert [lib.rcu_unc_from_p_threshold(i) for i in range(11)] == thr and lib.rcu_unc_from_p_threshold(11) == -1.0
    arr = (ctypes.c_double * 11)(*thr)
    assert lib.rcu_unc_from_p_supported(arr, 11) == 1
    for sub in ([0.5], [0.05, 0.95], thr[3:8]):
        assert lib.rcu_unc_from_p_supported((ctypes.c_double * len(sub))(*sub), len(sub)) == 1
    for bad in ([0.25], [0.5, 0.3], [0.5, 0.5]):
        assert lib.rcu_unc_from_p_supported((ctypes.c_double * len(bad))(*bad), len(bad)) == 0
    member = g['probe_member'].sum(0)
    got = [lib.rcu_unc_from_p_exceeded(float(v), arr, 11) for v in g['probe_bits'].view(np.float32)]
    assert got == [int(m) for m in member]
    one = (ctypes.c_double * 1)(0.95)
    assert [lib.rcu_unc_from_p_exceeded(float(v), one, 1) for v in g['probe_bits'].view(np.float32)] == [int(m) for m in g['probe_member'][10]]
    # the report stored with the fixture: every float32 in [0, 1] was run through the reference; only the 0.95 threshold has ragged windows
    assert int(g['values_scanned']) == 0x3F800000 + 1
    assert list(g['lo_width']) == [0] * 10 + [3] and list(g['hi_width']) == [0] * 10 + [2]
    with open(os.path.join(ROOT, 'reliability-challenges-uncertainty_amd', 'csrc', 'rcu_ue_table.inc')) as f:
        text = f.read()
    for k in range(11):
        assert '0x{:08x}u, {}u, 0x{:08x}u, {}u'.format(int(g['lo_first'][k]), int(g['lo_width'][k]), int(g['hi_first_false'][k]), int(g['hi_width'][k])) in text


def _plan_rows(h, w, n, cin=4, **options):
    """The planner alone (rcu_unet_plan: no device memory): [(name, kernel, height, width, grid_height, grid_width)] of the shipped BraTS / ISIC architecture."""
    import ctypes
    from rcu_amd import _lib
    lib = _lib.load()
    desc = _lib.UnetDesc(nb_classes=2, in_channels=cin, depth=4, start_filters=32, has_dropout=1, dropout_center=-1, sigma_out=options.pop('sigma_out', 0),
                         bn=1, height=h, width=w, max_batch=n, residual=options.pop('residual', 0), provide_features=options.pop('provide_features', 0))
    opts = _lib.UnetOptions()
    lib.rcu_unet_default_options(ctypes.byref(opts))
    for key, value in options.items():
        setattr(opts, key, value)
    handle = ctypes.c_void_p()
    _lib.check(lib.rcu_unet_plan(ctypes.byref(desc), ctypes.byref(opts), ctypes.byref(handle)))
    try:
        assert lib.rcu_unet_finalize_weights(handle) == -4      # RCU_ERR_STATE: a plan is inspected, never run
        rows = []
        for i in range(lib.rcu_unet_num_layers(handle)):
            info = _lib.LayerInfo()
            _lib.check(lib.rcu_unet_layer_info(handle, i, ctypes.byref(info)))
            rows.append((info.name.decode(), info.kernel.decode(), info.height, info.width, info.grid_height, info.grid_width))
        return rows
    finally:
        lib.rcu_unet_destroy(handle)


def test_planner_pads_the_levels_of_the_references_real_shapes():
    """csrc/rcu_api.hip choose_level_extents (round 6).  The reference's BraTS slices are 240 x 240 (scripts/create_brats18_dataset.py:53-72 never
    crops; levels 240 / 120 / 60 / 30 / 15), ISIC's 192 x 256 ends in a 12 x 16 level (scripts/prepare_isic_data.py:29-30): their levels are
    ALLOCATED with whole-tile extents and every layer runs a Winograd kernel (rounds 1-5: >= 16 of 23 layers of a 240 x 240 slice on the direct
    kernels).  Whole-tile shapes -- the benchmark's 192 x 128 and 256 x 256 -- keep their plans; pad_levels = 0, residual blocks' adding units
    and the feature tap keep real extents where they must."""
    native = _plan_rows(240, 240, 155)
    assert len(native) == 23 and native[0][1].startswith('conv3x3_first')
    assert sum('winograd' in r[1] for r in native) == 22 and not any('igemm' in r[1] for r in native)
    grids = {(r[2], r[3]): (r[4], r[5]) for r in native if 'upconv' not in r[0]}
    assert grids[(120, 120)] == (128, 128) and grids[(60, 60)] == (64, 64) and grids[(30, 30)] == (32, 32) and grids[(15, 15)] == (16, 16)
    assert grids[(240, 240)][0] in (240, 256) and grids[(240, 240)][1] == 256
    up = {(r[2], r[3]): (r[4], r[5]) for r in native if 'upconv' in r[0]}          # an up-convolution's tiles walk the low-resolution level
    assert up[(30, 30)] == (16, 16) and up[(240, 240)] == (128, 128)
    isic = _plan_rows(192, 256, 32, cin=3)
    assert sum('winograd' in r[1] for r in isic) == 22
    isic_grids = {(r[2], r[3]): (r[4], r[5], r[1]) for r in isic if 'upconv' not in r[0]}
    assert isic_grids[(12, 16)][:2] == (16, 16)
    assert isic_grids[(24, 32)] == (24, 32, 'conv3x3_winograd4<S4T8x32,N32,K8>')       # 32 wide, 8 | height: the full-width tile of four slices, unpadded
    assert all((r[2], r[3]) == (r[4], r[5]) for r in isic if r[2] >= 48 and 'upconv' not in r[0])       # the levels with whole tiles are left alone
    # the benchmark shapes: nothing is padded, the round-5 kernels
    bench = _plan_rows(192, 128, 640)
    assert all((r[4], r[5]) == ((r[2] // 2, r[3] // 2) if 'upconv' in r[0] else (r[2], r[3])) for r in bench)
    assert [r[1] for r in bench][1] == 'conv3x3_winograd4<T32x32,N32,K8>' and bench[9][1] == 'conv3x3_winograd4<S8T12x8,N32,K8>'
    assert all((r[2], r[3]) == (r[4], r[5]) for r in _plan_rows(256, 256, 32, cin=3) if 'upconv' not in r[0])
    # A/B switch, and the layers that have no kernel that keeps the padding's zeros
    direct = _plan_rows(240, 240, 155, pad_levels=0)
    assert sum('igemm' in r[1] for r in direct) >= 16 and all((r[4], r[5]) == (r[2], r[3]) for r in direct if 'upconv' not in r[0])
    residual = _plan_rows(240, 240, 8, residual=1)
    assert all((r[4], r[5]) == (r[2], r[3]) for r in residual if 'upconv' not in r[0])      # the adding units run the direct kernels: real extents
    feat = _plan_rows(240, 240, 8, provide_features=1)      # the feature tap gets a compact copy behind the forward: level 0 is padded like any other
    assert [r[1:] for r in feat] == [r[1:] for r in _plan_rows(240, 240, 8)]
    # a batch whose padded level 0 would pass 2 GB keeps to what fits: 640 samples of 240 x 240 x 32 channels are 4.7 GB -- the plan is refused
    # at creation either way (rcu_unet_create: 2^31 elements), the planner itself must not crash
    assert len(_plan_rows(240, 240, 640)) == 23
