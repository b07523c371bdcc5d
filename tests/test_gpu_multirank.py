"""Two ranks sharing the one GPU of the test box run the pipelined sharded step (ShardedMcRunner.step_async:
async reduce, finalize on a side stream of the root) on the HIP engine and must reproduce the single-rank result.
RCCL refuses two ranks on one device, so the process group is gloo with device tensors; the RCCL launch itself
(one rank per GPU) is what bench.py --gpus N does."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_pipelined_matches_single_rank():
    # a child process: the ranks are spawned from an interpreter that has not touched the GPU
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'multirank_single_gpu_probe.py'), 'gloo'],
                       capture_output=True, text=True, timeout=550, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if 'max |pipelined' in ln][-1]
    worst = float(line.rsplit('=', 1)[1])
    assert worst < 1e-5      # float32 sums in a different order across ranks
