"""Two ranks sharing the one GPU of the test box run the pipelined sharded step (ShardedMcRunner.step_async:
async reduce, finalize on a side stream of the root) on the HIP engine and must reproduce the single-rank result.
RCCL refuses two ranks on one device, so the process group is gloo with device tensors; the RCCL launch itself
(one rank per GPU) is what bench.py --gpus N does."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_one_gpu_pipelined_matches_single_rank():
    # a child process: the ranks are spawned from an interpreter that has not touched the GPU.  Three processes share the one GPU
    # of the test box here (pytest's own included); gloo's rendezvous over the loopback has been seen to hang once in that setting,
    # so a hung attempt (the probe dumps its stacks and exits after 240 s) gets one retry -- two hangs in a row fail the test.
    for attempt in range(2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'multirank_single_gpu_probe.py'), 'gloo'],
                           capture_output=True, text=True, timeout=400, cwd=ROOT)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if 'max |pipelined' in ln][-1]
    worst = float(line.rsplit('=', 1)[1])
    assert worst == 0.0      # exact statistics (round 5): the ranks' partial sums merge to the one-rank bits, whatever the order


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize('extra', [['--mc', '4'], ['--mc', '4', '--aleatoric'], ['--ensemble', '3'], ['--mc', '4', '--ws-transport', 'p2p']],
                         ids=['mc', 'aleatoric', 'ensemble', 'mc-ws-p2p'])
def test_bench_two_ranks_through_torch_distributed_run(extra):
    """The driver's N>1 launch line (python -m torch.distributed.run ... bench.py --gpus N) with both ranks on the one
    GPU of the test box over gloo (test-only switches of bench.py): rank 0 prints the one JSON line -- for the MC-dropout
    workload, its sigma-head extension and the ensemble workload."""
    import json
    env = dict(os.environ, RCU_BENCH_SINGLE_DEVICE='1', RCU_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2',
           '--warmup', '1'] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['value'] > 0 and d['scaling'] == 'strong'
    assert d['roofline']['launches'] > 0 and d['cpu_baseline'] is None
    assert d['parity']['bin_ids_equal'] and d['parity']['ece_delta_same_maps'] < 1e-9


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_takes_several_volumes_per_step_and_reports_its_devices():
    """`--volumes-per-step 2` (the rule's choice at 8 ranks: a rank's passes of two consecutive volumes run as one launch, so the launches
    keep the size they have on one GPU): `--steps 3` stays three VOLUMES -- one step of two and one of one --, the forward passes are counted
    in volumes, the parity leg reads the first volume of the last chunk, and the line carries the all-gathered device identities."""
    import json
    env = dict(os.environ, RCU_BENCH_SINGLE_DEVICE='1', RCU_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
           str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--mc', '4', '--volumes-per-step', '2']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert d['steps'] == 3 and d['config']['volumes_per_step'] == 2 and d['config']['samples_per_launch'] == 640
    assert sum(d['forwards_per_rank']) == 3 * 5                       # three volumes x (4 MC passes + the weight-scaling pass)
    assert len(d['devices']) == 2 and {e['rank'] for e in d['devices']} == {0, 1} and all(e['name'] for e in d['devices'])
    assert d['distinct_devices'] == 1                                 # both ranks on the one GPU of the test box
    assert d['parity']['bin_ids_equal'] and d['parity']['ece_delta_same_maps'] < 1e-9 and d['parity']['ue_counts_equal']
    assert d['all_outputs']['value'] > 0
    # the rule itself: one volume per step up to 4 ranks, two at 8 (T + 1 = 21 jobs, groups of 4)
    sys.path.insert(0, ROOT)
    import bench
    assert [bench.volumes_per_step(w, 21, 4) for w in (1, 2, 4, 8, 16)] == [1, 1, 1, 2, 4] and bench.volumes_per_step(8, 10, 1) == 1


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_plain_command_line_starts_its_own_ranks():
    """`python bench.py --gpus 2` -- no launcher, the form the round driver uses for N = 1 -- must start the two ranks
    itself (a fresh child process, before any GPU call), print the one JSON line and exit 0.  Both ranks share the one GPU
    of the test box over gloo (test-only switches); on an 8-GPU node the same line runs one rank per GPU over RCCL."""
    import json
    env = dict(os.environ, RCU_BENCH_SINGLE_DEVICE='1', RCU_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--mc', '5']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['n_ranks_seen'] == 2
    assert len(d['forwards_per_rank']) == 2 and sum(d['forwards_per_rank']) == 2 * 6      # 2 steps x (5 MC + ws pass)
    assert abs(d['forwards_per_rank'][0] - d['forwards_per_rank'][1]) <= 1
    assert d['roofline']['frac'] <= 1.0 and d['roofline']['canonical_frac'] > d['roofline']['frac']
    assert d['parity']['bin_ids_equal'] and d['parity']['ece_delta_same_maps'] < 1e-9


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_rccl_world1_rehearsal_matches_the_plain_step_bit_for_bit():
    """The exchange path of the multi-GPU runner over RCCL itself, as far as ONE GPU allows: a one-rank "nccl" process group, every
    volume through ShardedMcRunner._exchange -- asynchronous reduce, side-stream finalize, record_stream, drain; 'reduce' and 'p2p'
    transports (the latter a no-op when the root owns the weight-scaling pass: asserted in the tool) -- must give the bits of the plain
    world-1 step (tools/rccl_world1_rehearsal.py), and the bench line runs through it (RCU_BENCH_FORCE_PG=1)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rccl_world1_rehearsal.py'), '8', '4', '3'],
                       capture_output=True, text=True, timeout=800, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['backend'] == 'nccl' and d['world'] == 1 and d['bits_equal'] and d['avg_probe_ok']
    for transport in ('reduce', 'p2p'):
        assert d[transport]['bits_equal_sync'] and d[transport]['bits_equal_async'] and d[transport]['p2p_messages'] == 0
        assert d[transport]['inflight_after_drain'] == 0
    env = dict(env, RCU_BENCH_FORCE_PG='1')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--mc', '4', '--no-cpu-baseline']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=800, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    b = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert b['n_gpus'] == 1 and b['n_ranks_seen'] == 1 and b['rccl_rehearsal']['backend'] == 'nccl'
    assert b['forwards_per_rank'] == [2 * 5] and b['parity']['ece_delta_same_maps'] is not None if 'ece_delta_same_maps' in b['parity'] else True
    assert b['all_outputs']['value'] > 0


def _launch_two_ranks(script, cfg, env, out_root=None, attempts=2, ranks=2):
    """`python -m torch.distributed.run --nproc-per-node 2 <script> -config_file <cfg>`; gloo's loopback rendezvous has been seen to hang once with
    three processes on one GPU, so a hung or failed attempt gets one retry (with the output root of the first attempt removed)."""
    import shutil
    last = None
    for _ in range(attempts):
        if out_root is not None:
            shutil.rmtree(out_root, ignore_errors=True)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port()), script, '-config_file', cfg]
        try:
            last = subprocess.run(cmd, capture_output=True, text=True, timeout=400 if ranks <= 2 else 700, cwd=ROOT, env=env)
        except subprocess.TimeoutExpired as exc:
            last = subprocess.CompletedProcess(cmd, 124, stdout=str(exc.stdout or ''), stderr='timed out: ' + str(exc.stderr or ''))
            continue
        if last.returncode == 0:
            break
    return last


def _script_setup(tmp_path, seeds=(20,), mc=6, coalesce=None, shape=(32, 32)):
    """Two small BraTS-like subjects, model dir(s) + checkpoint(s), split and two YAML files that differ in their test_dir only.
    ``coalesce``: ``others.coalesce_pixels`` (0: the loader's batches of 4 slices as they are; default: the scripts' -- everything in one step)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import test_gpu_scripts as tgs
    cfg, vols, _, _ = tgs._setup(tmp_path, mc=mc if len(seeds) == 1 else None, seeds=seeds, shape=shape)
    with open(cfg) as f:
        text = f.read()
    if coalesce is not None:
        assert '  others:\n' in text
        text = text.replace('  others:\n', '  others:\n    coalesce_pixels: {}\n'.format(int(coalesce)), 1)
    cfgs = []
    for tag in ('one', 'two'):
        path = str(tmp_path / 'cfg_{}.yaml'.format(tag))
        with open(path, 'w') as f:
            f.write(text.replace(str(tmp_path / 'out'), str(tmp_path / 'out_{}'.format(tag))))
        cfgs.append(path)
    return cfgs, vols


def _written(out_root):
    import glob
    dirs = glob.glob(os.path.join(out_root, '*'))
    assert len(dirs) == 1, dirs
    files = {os.path.basename(f): open(f, 'rb').read() for f in sorted(glob.glob(os.path.join(dirs[0], '*')))
             if f.endswith(('.nii.gz', 'metrics.csv'))}
    return files


@pytest.mark.gpu
@pytest.mark.timeout(1200)
@pytest.mark.parametrize('script,seeds', [('brats_test_default.py', (20,)), ('brats_test_ensemble.py', (20, 21, 22))], ids=['mc', 'ensemble'])
def test_drop_in_script_under_torch_distributed_run_writes_the_one_process_files(tmp_path, script, seeds):
    """`python -m torch.distributed.run --nproc-per-node 2 bin-dl/brats_test_default.py -config_file ...` (both ranks on the one GPU of the
    test box: the scripts fall back to gloo when the node has fewer GPUs than ranks) shards the MC passes / ensemble members of every batch
    over the two ranks behind the unchanged script surface and must write, byte for byte, the .nii.gz files and the metrics.csv of the plain
    one-process run of the same script under the same YAML file: masks are a function of (seed, batch, pass), statistics are exact sums."""
    (cfg_one, cfg_two), vols = _script_setup(tmp_path, seeds=seeds)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    path = os.path.join(ROOT, 'bin-dl', script)
    r1 = subprocess.run([sys.executable, path, '-config_file', cfg_one], capture_output=True, text=True, timeout=500, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    r2 = _launch_two_ranks(path, cfg_two, env, str(tmp_path / 'out_two'))
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
    one, two = _written(str(tmp_path / 'out_one')), _written(str(tmp_path / 'out_two'))
    assert sorted(one) == sorted(two) and len(one) == 2 * len(vols) + 1
    for name in one:
        assert one[name] == two[name], name
    # both ranks worked: the sharded steps report their share when the run ends
    import re
    shares = {int(m.group(1)): (int(m.group(2)), int(m.group(3)))
              for m in re.finditer(r'rank (\d) of 2: (\d+) forward passes in (\d+) batches', r2.stdout + r2.stderr)}
    assert sorted(shares) == [0, 1], (r2.stdout[-1500:], r2.stderr[-1500:])
    batches = shares[0][1]
    jobs = batches * (len(seeds) if len(seeds) > 1 else 6 + 1)         # K members, or T = 6 passes + the weight-scaling pass, per batch
    assert shares[1][1] == batches and shares[0][0] + shares[1][0] == jobs and abs(shares[0][0] - shares[1][0]) <= 1


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_isic_script_under_torch_distributed_run_writes_the_one_process_files(tmp_path):
    """bin-dl/isic_test_default.py (every image a subject, labels prepared per batch, inputs linked next to the outputs) with `others.mc` under
    the launcher, two ranks on the one GPU: the files of the plain run, byte for byte."""
    from PIL import Image
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import test_gpu_scripts as tgs
    from oracle import unet_oracle as uo
    from rcu_amd import management as mgt
    params = dict(nb_classes=2, in_channels=3, depth=4, start_filters=8, dropout=0.2)
    prefix = tmp_path / 'isic' / 'ISIC-2017_Test_v2'
    img_dir, lab_dir = str(prefix) + '_Data', str(prefix) + '_Part1_GroundTruth'
    os.makedirs(img_dir)
    os.makedirs(lab_dir)
    rng = np.random.RandomState(8)
    ids = ['ISIC_00002{:02d}'.format(i) for i in range(5)]
    for i, id_ in enumerate(ids):
        Image.fromarray(rng.randint(0, 255, (64, 64, 3)).astype(np.uint8)).save(os.path.join(img_dir, id_ + '.jpg'))
        lab = np.zeros((64, 64), np.uint8)
        lab[4 * i:4 * i + 20, 8:40] = 255
        Image.fromarray(lab).save(os.path.join(lab_dir, id_ + '_segmentation.png'))
    mf = mgt.ModelFiles(str(tmp_path / 'train'), 'isic')
    mgt.save_model(mf, 'unet', params, uo.synthetic_state(27, **params))
    cfgs = []
    for tag in ('one', 'two'):
        path = str(tmp_path / 'cfg_{}.yaml'.format(tag))
        with open(path, 'w') as f:
            f.write(tgs.ISIC_MC_YAML.format(test_dir=str(tmp_path / ('out_' + tag)), model_dir=mf.model_dir, dataset=str(prefix))
                    .replace('mc: 2', 'mc: 5').replace('batch_size: 1', 'batch_size: 2'))
        cfgs.append(path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    script = os.path.join(ROOT, 'bin-dl', 'isic_test_default.py')
    r1 = subprocess.run([sys.executable, script, '-config_file', cfgs[0]], capture_output=True, text=True, timeout=500, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    r2 = _launch_two_ranks(script, cfgs[1], env, str(tmp_path / 'out_two'))
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
    one, two = _written(str(tmp_path / 'out_one')), _written(str(tmp_path / 'out_two'))
    assert sorted(one) == sorted(two) and len(one) == 2 * len(ids) + 1
    for name in one:
        assert one[name] == two[name], name
    assert 'rank 1 of 2: ' in r2.stdout + r2.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_deterministic_script_under_the_launcher_is_rank_zeros_alone(tmp_path):
    """A configuration with one deterministic forward pass per batch (no `others.mc`) has nothing to shard over samples: under
    `torch.distributed.run` rank 0 runs it and writes the plain run's files, the other rank leaves at once -- and nobody waits for it."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import test_gpu_scripts as tgs
    cfg, vols, _, _ = tgs._setup(tmp_path)            # mc=None: SegmentationPredictStep
    with open(cfg) as f:
        text = f.read()
    cfgs = []
    for tag in ('one', 'two'):
        path = str(tmp_path / 'cfg_{}.yaml'.format(tag))
        with open(path, 'w') as f:
            f.write(text.replace(str(tmp_path / 'out'), str(tmp_path / 'out_{}'.format(tag))))
        cfgs.append(path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    script = os.path.join(ROOT, 'bin-dl', 'brats_test_default.py')
    r1 = subprocess.run([sys.executable, script, '-config_file', cfgs[0]], capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    r2 = _launch_two_ranks(script, cfgs[1], env, str(tmp_path / 'out_two'))
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
    one, two = _written(str(tmp_path / 'out_one')), _written(str(tmp_path / 'out_two'))
    assert sorted(one) == sorted(two) and len(one) == 2 * len(vols) + 1 and all(one[k] == two[k] for k in one)
    assert 'rank 1: this configuration has one forward pass per batch' in r2.stdout + r2.stderr


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_full_size_script_run_is_the_same_bytes_on_two_ranks(tmp_path):
    """The headline configuration behind the drop-in script, at its own size: two BraTS-sized subjects (160 slices of 4 x 192 x 128), the shipped
    `batch_size: 32`, `mc: 20`, once with loader batches as they are and once coalesced to one volume per step (640-sample pass groups, the
    folded 12x8 kernel in the plan): `bin-dl/brats_test_default.py` under the launcher with two ranks (on the one GPU) writes the plain
    run's files byte for byte -- canonical plans, exact statistics and seeded masks at full width."""
    import json
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import bench
    from rcu_amd import data as data_mod
    from rcu_amd import management as mgt
    from rcu_amd import nifti
    from test_script_surface_cpu import BRATS_MC_YAML
    x, _, target = bench.make_volume(20)
    names = []
    for i in range(2):
        name = 'Brats18_FULL_{:03d}_1'.format(i)
        props = nifti.ImageProperties((bench.WIDTH, bench.HEIGHT, bench.SLICES), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
        data_mod.write_volume(str(tmp_path / 'ds'), name, (x + 0.05 * i).permute(0, 2, 3, 1).numpy(), target.numpy(), props)
        names.append(name)
    torch.manual_seed(20)
    from rcu_amd.model import UNet
    model = UNet(**bench.MODEL_PARAMS)
    mf = mgt.ModelFiles(str(tmp_path / 'train'), 'full')
    mgt.save_model(mf, 'unet', bench.MODEL_PARAMS, {k: v.cpu() for k, v in model.state_dict().items()})
    split = str(tmp_path / 'split.json')
    with open(split, 'w') as f:
        json.dump({'train': [], 'valid': [], 'test': names}, f)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    script = os.path.join(ROOT, 'bin-dl', 'brats_test_default.py')
    for tag, extra in (('plain', ''), ('coalesced', '    coalesce_pixels: {}\n'.format(bench.SLICES * bench.HEIGHT * bench.WIDTH))):
        cfgs = []
        for world in ('one', 'two'):
            out = tmp_path / 'out_{}_{}'.format(tag, world)
            text = BRATS_MC_YAML.format(test_dir=str(out), model_dir=mf.model_dir, split=split, dataset=str(tmp_path / 'ds'))
            text = text.replace('  others:\n', '  others:\n' + extra, 1)
            path = str(tmp_path / 'cfg_{}_{}.yaml'.format(tag, world))
            with open(path, 'w') as f:
                f.write(text)
            cfgs.append((path, str(out)))
        r1 = subprocess.run([sys.executable, script, '-config_file', cfgs[0][0]], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
        r2 = _launch_two_ranks(script, cfgs[1][0], env, cfgs[1][1])
        assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
        one, two = _written(cfgs[0][1]), _written(cfgs[1][1])
        assert sorted(one) == sorted(two) and len(one) == 2 * len(names) + 1, tag
        for name in one:
            assert one[name] == two[name], (tag, name)
        p = nifti.read(os.path.join([d for d in [os.path.join(cfgs[0][1], e) for e in os.listdir(cfgs[0][1])]][0], names[0] + '_probabilities.nii.gz'))[0]
        assert p.shape == (bench.SLICES, bench.HEIGHT, bench.WIDTH) and 0 <= float(p.min()) and float(p.max()) <= 1 and float(p.std()) > 0


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_eight_ranks_write_the_one_process_files_too(tmp_path):
    """World size 8 (all eight ranks on the one GPU, gloo): T + 1 = 7 jobs per batch leave most ranks one job and one rank none per batch, the
    weight-scaling pass moves from rank to rank, eight partial sums meet in one reduce whose order is gloo's business -- and the files are
    still the one-process run's, byte for byte: exact sums are associative.  (``coalesce_pixels: 0``: the loader's four batches of 4 slices, so that
    the jobs rotate over the ranks from batch to batch; the two-rank tests above run the scripts' default, one coalesced step.)"""
    (cfg_one, cfg_two), vols = _script_setup(tmp_path, seeds=(20,), coalesce=0)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    path = os.path.join(ROOT, 'bin-dl', 'brats_test_default.py')
    r1 = subprocess.run([sys.executable, path, '-config_file', cfg_one], capture_output=True, text=True, timeout=500, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    r8 = _launch_two_ranks(path, cfg_two, env, str(tmp_path / 'out_two'), ranks=8)
    assert r8.returncode == 0, r8.stdout[-2000:] + r8.stderr[-4000:]
    one, eight = _written(str(tmp_path / 'out_one')), _written(str(tmp_path / 'out_two'))
    assert sorted(one) == sorted(eight) and len(one) == 2 * len(vols) + 1
    for name in one:
        assert one[name] == eight[name], name
    import re
    shares = {int(m.group(1)): int(m.group(2)) for m in re.finditer(r'rank (\d) of 8: (\d+) forward passes in (\d+) batches', r8.stdout + r8.stderr)}
    assert sorted(shares) == list(range(8)) and sum(shares.values()) == 4 * 7 and max(shares.values()) - min(shares.values()) <= 1


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_two_ranks_on_the_references_real_slice_size_write_the_one_process_files(tmp_path):
    """240 x 240 slices (padded levels) under the launcher: which kernel and which allocated extent a level gets is part of the plan,
    and plans are canonical -- two ranks write the one-process run's bytes here too."""
    (cfg_one, cfg_two), vols = _script_setup(tmp_path, seeds=(20,), mc=4, shape=(240, 240))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    path = os.path.join(ROOT, 'bin-dl', 'brats_test_default.py')
    r1 = subprocess.run([sys.executable, path, '-config_file', cfg_one], capture_output=True, text=True, timeout=500, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    r2 = _launch_two_ranks(path, cfg_two, env, str(tmp_path / 'out_two'))
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
    one, two = _written(str(tmp_path / 'out_one')), _written(str(tmp_path / 'out_two'))
    assert sorted(one) == sorted(two) and len(one) == 2 * len(vols) + 1
    for name in one:
        assert one[name] == two[name], name
