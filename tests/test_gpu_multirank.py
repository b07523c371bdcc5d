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
    assert worst < 1e-5      # float32 sums in a different order across ranks


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
