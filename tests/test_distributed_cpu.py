"""The N>1 path on CPU: world_size-2 gloo processes run ShardedMcRunner with an oracle-backed engine
(tests may use the oracle; the product engine is HIP-only) and must reproduce the single-process
result -- shard sizes uneven, weight-scaling pass on either rank, rotation over steps."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PARAMS = dict(nb_classes=2, in_channels=4, depth=4, start_filters=4, dropout=0.3)
T = 5


class OracleEngine:
    """Same interface as rcu_amd.distributed.HipEngine, arithmetic by the oracle (float32 sums)."""

    def __init__(self, state):
        from oracle import unet_oracle as uo
        self.state, self.uo = state, uo

    def buffers(self, x, with_ws):
        n, _, h, w = x.shape
        c = PARAMS['nb_classes']
        flat = torch.zeros(n * c * h * w * (2 if with_ws else 1))
        stats = flat[:n * c * h * w].view(n, c, h, w)
        ws = flat[n * c * h * w:].view(n, c, h, w) if with_ws else None
        return flat, stats, ws

    def ws_pass(self, x, ws_out):
        ws_out.copy_(torch.softmax(self.uo.unet_forward(self.state, x, None, **PARAMS), 1))

    def sample_masks(self, x, generator, passes=1):
        _, sites = self.uo.unet_plan(**PARAMS)
        return self.uo.sample_masks(sites, x.shape[0], PARAMS['dropout'], generator)

    def mc_pass(self, x, stats, masks=None, passes=1):
        for ms in ([masks] if passes == 1 else masks):      # a pass group: list of `passes` mask sets
            stats += torch.softmax(self.uo.unet_forward(self.state, x, ms, **PARAMS), 1)

    def member_pass(self, member_state, x, stats):
        stats += torch.softmax(self.uo.unet_forward(member_state, x, None, **PARAMS), 1)

    def finalize(self, stats, count):
        from oracle import summary_oracle as so
        p = stats / count
        return {'probabilities': p, 'entropy': so.torch_entropy(p, dim=1, keepdim=True)}


class OracleAleatoricEngine:
    """Same interface as rcu_amd.distributed.AleatoricHipEngine (sigma-head extension): the per-pass sigmas are plain sums
    in the reduce buffer, [statistics | sigma sum | ws probabilities | ws sigma]."""

    def __init__(self, state):
        from oracle import unet_oracle as uo
        self.state, self.uo = state, uo
        self.params = dict(PARAMS, sigma_out=True)

    def buffers(self, x, with_ws):
        n, _, h, w = x.shape
        vol = n * PARAMS['nb_classes'] * h * w
        shape = (n, PARAMS['nb_classes'], h, w)
        flat = torch.zeros(vol * (4 if with_ws else 2))
        stats = flat[:vol].view(shape)
        stats.sigma_sum = flat[vol:2 * vol].view(shape)
        ws = flat[2 * vol:].view((2,) + shape) if with_ws else None
        return flat, stats, ws

    def ws_pass(self, x, ws_out):
        lg, raw = self.uo.unet_forward(self.state, x, None, **self.params)
        ws_out[0].copy_(torch.softmax(lg, 1))
        ws_out[1].copy_(raw.abs())

    def mc_pass(self, x, stats, masks=None):
        lg, raw = self.uo.unet_forward(self.state, x, masks, **self.params)
        stats += torch.softmax(lg, 1)
        stats.sigma_sum += raw.abs()

    def finalize(self, stats, count):
        from oracle import summary_oracle as so
        p = stats / count
        return {'probabilities': p, 'entropy': so.torch_entropy(p, dim=1, keepdim=True), 'sigma': stats.sigma_sum / count}

    def ws_outputs(self, ws):
        return {'ws_probabilities': ws[0], 'ws_sigma': ws[1]}


def _sigma_inputs():
    from oracle import unet_oracle as uo
    params = dict(PARAMS, sigma_out=True)
    state = uo.synthetic_state(9, **params)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 4, 32, 32, generator=g)
    _, sites = uo.unet_plan(**params)
    return params, state, x, [uo.sample_masks(sites, 2, 0.3, g) for _ in range(T)]


def _inputs():
    from oracle import unet_oracle as uo
    state = uo.synthetic_state(7, **PARAMS)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 32, 32, generator=g)
    _, sites = uo.unet_plan(**PARAMS)
    mask_sets = [uo.sample_masks(sites, 2, 0.3, g) for _ in range(T)]
    return state, x, mask_sets


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rcu_amd.distributed import ShardedMcRunner
    torch.set_num_threads(2)
    state, x, mask_sets = _inputs()
    runner = ShardedMcRunner(None, T, ws_pass=True, rank=rank, world=world, engine=OracleEngine(state))
    for step in range(3):
        out = runner.step(x, step, mask_sets)
        if rank == 0:
            np.savez(os.path.join(out_dir, 'step{}.npz'.format(step)), **{k: v.numpy() for k, v in out.items()})
        else:
            assert out is None
    # pipelined form: results collected after all steps were issued
    pend = [runner.step_async(x, step, mask_sets) for step in range(3)]
    for step, p in enumerate(pend):
        out = p.result()
        if rank == 0:
            np.savez(os.path.join(out_dir, 'async{}.npz'.format(step)), **{k: v.numpy() for k, v in out.items()})
        else:
            assert out is None
    runner.drain()
    # the weight-scaling probabilities by send / recv from their owner instead of in the tail of the reduce buffer
    # (root 1: with T + 1 = 6 jobs on 2 ranks job 0 always runs on rank 0, so the tail really travels)
    p2p = ShardedMcRunner(None, T, ws_pass=True, rank=rank, world=world, engine=OracleEngine(state), ws_transport='p2p', root=1)
    pend = [p2p.step_async(x, step, mask_sets) for step in range(3)] + [PendingNow(p2p.step(x, 3, mask_sets))]
    for step, p in enumerate(pend):
        out = p.result()
        if rank == 1:
            np.savez(os.path.join(out_dir, 'p2p{}.npz'.format(step)), **{k: v.numpy() for k, v in out.items()})
        else:
            assert out is None
    p2p.drain()
    # masks drawn per (seed, volume, pass): the result must not depend on the world size, with and without pass groups
    for name, group in (('seeded', 1), ('seeded_grouped', 2)):
        rs = ShardedMcRunner(None, T, ws_pass=True, rank=rank, world=world, engine=OracleEngine(state), seed=5, pass_group=group)
        out = rs.step(x, 1)
        assert rs.forwards_run == len(rs.jobs_of(1, rank))
        if rank == 0:
            np.savez(os.path.join(out_dir, name + '.npz'), **{k: v.numpy() for k, v in out.items()})
    # ensemble members instead of MC passes: 3 members on 2 ranks, rotation over the steps
    from oracle import unet_oracle as uo
    from rcu_amd.distributed import ShardedEnsembleRunner
    members = [uo.synthetic_state(100 + k, **PARAMS) for k in range(3)]
    ens = ShardedEnsembleRunner(members, rank=rank, world=world, engine=OracleEngine(state))
    assert sorted(j for r in range(world) for j in ens.jobs_of(1, r)) == [1, 2, 3]
    pend = [ens.step_async(x, step) for step in range(2)]
    for step, p in enumerate(pend):
        out = p.result()
        if rank == 0:
            assert 'ws_probabilities' not in out
            np.savez(os.path.join(out_dir, 'ens{}.npz'.format(step)), **{k: v.numpy() for k, v in out.items()})
    ens.drain()
    # sigma-head extension: the sigma sums travel in the same reduce as the statistics
    _, sstate, sx, smasks = _sigma_inputs()
    sig = ShardedMcRunner(None, T, ws_pass=True, rank=rank, world=world, engine=OracleAleatoricEngine(sstate))
    pend = [sig.step_async(sx, step, smasks) for step in range(2)]
    for step, p in enumerate(pend):
        out = p.result()
        if rank == 0:
            np.savez(os.path.join(out_dir, 'sigma{}.npz'.format(step)), **{k: v.numpy() for k, v in out.items()})
    sig.drain()
    dist.destroy_process_group()


class OracleExactEngine(OracleEngine):
    """The exact statistics of the product engine (include/rcu.h RCU_MC_EXACT) restated on the CPU: float64 sums of addends rounded to
    multiples of 2^-40, (x + 6144) - 6144 -- every addition exact, so the merged sums cannot depend on the world size."""

    def buffers(self, x, with_ws):
        n, _, h, w = x.shape
        c = PARAMS['nb_classes']
        flat = torch.zeros(n * c * h * w * (2 if with_ws else 1), dtype=torch.float64)
        stats = flat[:n * c * h * w].view(n, c, h, w)
        ws = flat[n * c * h * w:].view(n, c, h, w) if with_ws else None
        return flat, stats, ws

    def mc_pass(self, x, stats, masks=None, passes=1):
        for ms in ([masks] if passes == 1 else masks):
            p = torch.softmax(self.uo.unet_forward(self.state, x, ms, **PARAMS), 1).double()
            stats += (p + 6144.0) - 6144.0


def _step_worker(rank, world, port, out_dir):
    """The step seam of the scripts: ShardedMcPredictStep + MultiPredictionSummary called by every rank with the same batches."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from rcu_amd import steps
    from rcu_amd.distributed import ShardedMcPredictStep, World
    torch.set_num_threads(2)
    state, x, _ = _inputs()
    step = ShardedMcPredictStep(T, World(rank, world, rank, 'cpu', 'gloo'), seed=5, engine_factory=lambda model: OracleExactEngine(state),
                                group_pixels=2 * x.shape[0] * 32 * 32)
    ctx = steps.TorchTestContext('cpu', None)
    for k in range(3):
        bc = steps.BatchContext({'images': x + 0.1 * k}, k)
        step(bc, None, ctx)
        multi = bc.output['multi_probabilities']
        if rank == 0:
            assert multi.count == T and bc.output['ws_probabilities'].shape == x[:, :2].shape
            np.savez(os.path.join(out_dir, 'w{}_batch{}.npz'.format(world, k)), sums=multi.numpy(), ws=bc.output['ws_probabilities'].numpy())
        else:
            assert multi is None and 'ws_probabilities' not in bc.output
            steps.MultiPredictionSummary()(bc, None, ctx)          # nothing to finalise off the root: no outputs, no error
            assert bc.output == {}
    step.finish()
    assert step._runner.forwards_run == sum(len(step._runner.jobs_of(k, rank)) for k in range(3))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_batch_step_hands_the_root_the_bits_of_the_one_process_step(tmp_path):
    """rcu_amd.distributed.ShardedMcPredictStep over gloo, two ranks, three batches: the root's ``multi_probabilities`` are the merged
    statistics -- with exact sums, the very bits the one-process step leaves there -- the other rank gets None; masks come from
    (seed, batch, pass), so no mask set is injected."""
    mp.spawn(_step_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    _step_worker(0, 1, 0, str(tmp_path))
    for k in range(3):
        two, one = np.load(os.path.join(str(tmp_path), 'w2_batch{}.npz'.format(k))), np.load(os.path.join(str(tmp_path), 'w1_batch{}.npz'.format(k)))
        assert np.array_equal(two['sums'].view(np.uint64), one['sums'].view(np.uint64))
        assert np.array_equal(two['ws'], one['ws'])
        scaled = one['sums'] * 2.0 ** 40
        assert np.array_equal(scaled, np.round(scaled)) and one['sums'].max() <= T
    a, b = np.load(os.path.join(str(tmp_path), 'w1_batch0.npz'))['sums'], np.load(os.path.join(str(tmp_path), 'w1_batch1.npz'))['sums']
    assert not np.array_equal(a, b)


class PendingNow:
    def __init__(self, value):
        self.value = value

    def result(self):
        return self.value


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_stream_lanes_are_one_lane_off_the_gpu():
    """rcu_amd.steps.StreamLanes on a CPU device: one lane whatever was asked for -- every launch runs in place, in order, on the
    statistics it was given, and there is nothing to merge (the gloo runs below rely on that)."""
    from rcu_amd import steps
    lanes = steps.StreamLanes('cpu', 3)
    assert lanes.count == 1 and lanes.streams == []
    stats = object()
    got = lanes.begin(stats, lambda: pytest.fail('no side statistics on one lane'), inputs=(torch.zeros(2),), first=5)
    assert got == [stats]
    seen = []
    for k in range(4):
        lanes.run(lambda st, lane, k=k: seen.append((k, st is stats, lane)))
    assert seen == [(k, True, 0) for k in range(4)]
    lanes.end(lambda a, b: pytest.fail('nothing to merge on one lane'))


def test_job_partition_properties():
    sys.path.insert(0, ROOT)
    from rcu_amd.distributed import ShardedMcRunner
    for world in (1, 2, 3, 4, 8):
        for ws in (True, False):
            r = ShardedMcRunner(None, 20, ws_pass=ws, rank=0, world=world, engine=object())
            for step in range(5):
                shards = [r.jobs_of(step, k) for k in range(world)]
                flat = sorted(j for s in shards for j in s)
                assert flat == ([0] if ws else []) + list(range(1, 21))        # every job exactly once
                assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
            if world == 8 and ws:   # 21 jobs over 8 ranks: the rotation evens the remainder out over 8 steps
                total = [sum(len(r.jobs_of(step, k)) for step in range(8)) for k in range(world)]
                assert total == [21] * 8


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process(tmp_path):
    from oracle import summary_oracle as so
    from oracle import unet_oracle as uo
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    state, x, mask_sets = _inputs()
    ws, multi = so.mc_probabilities(lambda xx, m: uo.unet_forward(state, xx, m, **PARAMS), x, mask_sets)
    ref = so.multi_prediction_summary(multi)
    for name in (['step{}.npz'.format(k) for k in range(3)] + ['async{}.npz'.format(k) for k in range(3)] +
                 ['p2p{}.npz'.format(k) for k in range(4)]):
        got = np.load(os.path.join(str(tmp_path), name))
        assert np.max(np.abs(got['ws_probabilities'] - ws.numpy())) < 1e-6
        assert np.max(np.abs(got['probabilities'] - ref['probabilities'].numpy())) < 1e-6
        assert np.max(np.abs(got['entropy'] - ref['entropy'].numpy())) < 2e-6
    sys.path.insert(0, ROOT)
    from rcu_amd.distributed import ShardedMcRunner, job_seed
    single = ShardedMcRunner(None, T, ws_pass=True, rank=0, world=1, engine=OracleEngine(state), seed=5).step(x, 1)
    other = ShardedMcRunner(None, T, ws_pass=True, rank=0, world=1, engine=OracleEngine(state), seed=6).step(x, 1)
    assert float((single['probabilities'] - other['probabilities']).abs().max()) > 1e-4     # the seed matters
    assert len({job_seed(5, k, j) for k in range(50) for j in range(1, 51)}) == 2500         # distinct stream per (volume, pass)
    for name in ('seeded', 'seeded_grouped'):
        got = np.load(os.path.join(str(tmp_path), name + '.npz'))
        for key in ('probabilities', 'entropy', 'ws_probabilities'):
            assert np.max(np.abs(got[key] - single[key].numpy())) < 2e-6, (name, key)
    members = [uo.synthetic_state(100 + k, **PARAMS) for k in range(3)]
    multi = so.ensemble_probabilities([lambda xx, m, st=st: uo.unet_forward(st, xx, m, **PARAMS) for st in members], x)
    ref = so.multi_prediction_summary(multi)
    for step in range(2):
        got = np.load(os.path.join(str(tmp_path), 'ens{}.npz'.format(step)))
        assert np.max(np.abs(got['probabilities'] - ref['probabilities'].numpy())) < 1e-6
        assert np.max(np.abs(got['entropy'] - ref['entropy'].numpy())) < 2e-6
    params, sstate, sx, smasks = _sigma_inputs()
    passes = [uo.unet_forward(sstate, sx, mk, **params) for mk in smasks]
    p_ref = torch.stack([torch.softmax(lg, 1) for lg, _ in passes]).mean(0)
    s_ref = torch.stack([raw.abs() for _, raw in passes]).mean(0)
    lg0, raw0 = uo.unet_forward(sstate, sx, None, **params)
    for step in range(2):
        got = np.load(os.path.join(str(tmp_path), 'sigma{}.npz'.format(step)))
        assert set(got.files) == {'probabilities', 'entropy', 'sigma', 'ws_probabilities', 'ws_sigma'}
        assert np.max(np.abs(got['probabilities'] - p_ref.numpy())) < 1e-6
        assert np.max(np.abs(got['sigma'] - s_ref.numpy())) < 1e-5 * max(1.0, float(s_ref.max()))
        assert np.max(np.abs(got['ws_probabilities'] - torch.softmax(lg0, 1).numpy())) < 1e-6
        assert np.max(np.abs(got['ws_sigma'] - raw0.abs().numpy())) < 1e-6 * max(1.0, float(raw0.abs().max()))


# ------------------------------------------------------------------------------------------------ world 8
# The shapes of the 8-GPU runs (BASELINE configs[2..4]: T = 20 MC passes, K = 10 members, T = 50 sigma-head passes) on 8 gloo
# ranks: job partition, pass groups, the lane rotation argument (lanes collapse to one on a CPU device, the rotation code still
# runs), one collective per volume, both ws transports -- against the single-process result, and the forward counts per rank.
W8_PARAMS = dict(nb_classes=2, in_channels=4, depth=2, start_filters=4, dropout=0.3)
W8_VOLUMES = 8


class TinyEngine(OracleEngine):
    """OracleEngine over a depth-2 U-Net on 8x8 images: a forward pass takes a millisecond."""

    def _fwd(self, state, x, masks):
        return torch.softmax(self.uo.unet_forward(state, x, masks, **W8_PARAMS), 1)

    def ws_pass(self, x, ws_out):
        ws_out.copy_(self._fwd(self.state, x, None))

    def sample_masks(self, x, generator, passes=1):
        _, sites = self.uo.unet_plan(**W8_PARAMS)
        return self.uo.sample_masks(sites, x.shape[0], W8_PARAMS['dropout'], generator)

    def mc_pass(self, x, stats, masks=None, passes=1, lane=0):
        for ms in ([masks] if passes == 1 else masks):
            stats += self._fwd(self.state, x, ms)

    def member_pass(self, member_state, x, stats, lane=0):
        stats += self._fwd(member_state, x, None)


def _w8_cases(rank, world):
    from oracle import unet_oracle as uo
    from rcu_amd.distributed import ShardedEnsembleRunner, ShardedMcRunner
    state = uo.synthetic_state(11, **W8_PARAMS)
    members = [uo.synthetic_state(200 + k, **W8_PARAMS) for k in range(10)]
    x = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(8))
    mk = lambda **kw: ShardedMcRunner(None, rank=rank, world=world, engine=TinyEngine(state), seed=3, lanes=2, **kw)  # noqa: E731
    return x, {
        'mc20': mk(mc_steps=20, ws_pass=True, pass_group=2),      # the default transport (round 6): point to point where the backend can -- CPU tensors over gloo can
        'mc20_p2p': mk(mc_steps=20, ws_pass=True, pass_group=2, ws_transport='p2p'),
        'mc20_reduce': mk(mc_steps=20, ws_pass=True, pass_group=2, ws_transport='reduce'),
        'mc50': mk(mc_steps=50, ws_pass=True, pass_group=2),
        'ens10': ShardedEnsembleRunner(members, rank=rank, world=world, engine=TinyEngine(state), lanes=2),
    }


def _w8_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    x, cases = _w8_cases(rank, world)
    from rcu_amd.distributed import ShardedMcRunner
    counts = {}
    for name, runner in cases.items():
        pend = [runner.step_async(x, k) for k in range(W8_VOLUMES)]
        outs = [p.result() for p in pend]
        runner.drain()
        counts[name] = runner.forwards_run
        if isinstance(runner, ShardedMcRunner) and world > 1:      # the default resolves to the point-to-point transport here; an explicit choice is kept
            assert runner.ws_transport == ('reduce' if name.endswith('_reduce') else 'p2p'), (name, runner.ws_transport)
        if rank == 0:
            np.savez(os.path.join(out_dir, name + '.npz'),
                     **{'{}_{}'.format(key, k): v.numpy() for k, out in enumerate(outs) for key, v in out.items()})
        else:
            assert all(o is None for o in outs)
    gathered = [None] * world
    dist.all_gather_object(gathered, counts)
    if rank == 0:
        import json
        with open(os.path.join(out_dir, 'counts.json'), 'w') as f:
            json.dump(gathered, f)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_eight_rank_gloo_matches_single_process_and_balances(tmp_path):
    import json
    world = 8
    mp.spawn(_w8_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    x, cases = _w8_cases(0, 1)
    with open(os.path.join(str(tmp_path), 'counts.json')) as f:
        counts = json.load(f)
    for name, runner in cases.items():
        got = np.load(os.path.join(str(tmp_path), name + '.npz'))
        for k in range(W8_VOLUMES):
            ref = runner.step(x, k)
            assert set(ref) == {key[:-len('_{}'.format(k))] for key in got.files if key.endswith('_{}'.format(k))}
            for key, v in ref.items():
                assert np.max(np.abs(got['{}_{}'.format(key, k)] - v.numpy())) < 2e-6, (name, k, key)
        per_rank = [c[name] for c in counts]
        jobs = runner.jobs_per_step * W8_VOLUMES
        assert sum(per_rank) == jobs and runner.forwards_run == jobs
        # the rotation of the job list over the volumes: 8 volumes hand every rank the same number of forward passes
        assert max(per_rank) - min(per_rank) <= (0 if jobs % world == 0 else 1), (name, per_rank)
    # the three shapes: 21, 51 and 10 jobs per volume over 8 ranks
    assert [c['mc20'] for c in counts] == [21] * 8 and [c['mc50'] for c in counts] == [51] * 8 and [c['ens10'] for c in counts] == [10] * 8


# ------------------------------------------------------------------------------------------------ world 1 through the exchange path
def _w1_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import unet_oracle as uo
    from rcu_amd.distributed import ShardedMcRunner
    state = uo.synthetic_state(11, **W8_PARAMS)
    x = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(9))
    mk = lambda **kw: ShardedMcRunner(None, 6, rank=0, world=1, engine=TinyEngine(state), seed=4, pass_group=2, **kw)  # noqa: E731
    plain = mk()
    ok = True
    for transport in ('reduce', 'p2p'):
        forced = mk(force_exchange=True, ws_transport=transport)
        ref = [plain.step(x, k) for k in range(3)]
        pend = [forced.step_async(x, k, depth=2) for k in range(3)]          # more volumes than `depth`: the oldest is retired on the way
        outs = [p.result() for p in pend]
        forced.drain()
        same = all(torch.equal(a[key], b[key]) for a, b in zip(outs, ref) for key in b) and all(set(a) == set(b) for a, b in zip(outs, ref))
        sync = forced.step(x, 1)
        same = same and all(torch.equal(sync[key], ref[1][key]) for key in ref[1])
        ok = ok and same and forced.p2p_messages == 0 and len(forced._inflight) == 0 and forced.ws_owner(2) == forced.root
    with open(out_path, 'w') as f:
        f.write('ok' if ok else 'mismatch')
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_one_rank_group_through_the_exchange_path_gives_the_plain_step(tmp_path):
    """`force_exchange` (the one-GPU rehearsal of the RCCL path, tools/rccl_world1_rehearsal.py) on CPU over gloo: a one-rank group, every volume
    through `_exchange` -- asynchronous work handles, at most `depth` in flight, drain -- for both ws transports gives the bits of the plain
    world-1 step, and the point-to-point transport sends nothing when the root owns the weight-scaling pass."""
    out = str(tmp_path / 'w1.txt')
    mp.spawn(_w1_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    assert open(out).read() == 'ok'
