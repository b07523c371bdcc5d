"""Multi-GPU: MC passes / ensemble members sharded over the ranks of one node, one process per GPU.

The reference has no distributed code (single process, `'cuda'`); this is the
MI355X-native addition BASELINE.json asks for.  Passes are independent given the input and the
weights (rechun/dl/customsteps.py:30-34; bin-dl/brats_test_ensemble.py:85-92), so every rank runs
its share of the T stochastic passes (and at most one rank the deterministic weight-scaling pass)
into a local statistics blob; because the blob holds plain sums (include/rcu.h, rcu_mc_*), ONE
sum-reduce over RCCL/xGMI merges the shards, whatever the shard sizes, and the root finalises
with divisor T.  The weight-scaling probabilities ride in the same buffer (zeros on every rank
but the one that computed them), so there is exactly one collective per volume.

Job j of step k (j = 0 is the weight-scaling pass, 1..T the MC passes) runs on rank
(j + k * jobs_per_step) mod world: the rotation evens out the remainder of T+1 over the ranks
across consecutive volumes.
"""
import torch
import torch.distributed as dist

from . import steps as steps_mod


class HipEngine:
    """The product engine: fused forward + softmax + accumulate on librcu_hip."""

    def __init__(self, model, do_mi=False, do_var=False):
        self.model = model
        self.do_mi, self.do_var = do_mi, do_var

    def buffers(self, x, with_ws):
        """-> (flat reduce buffer, statistics object living in its head, ws tensor or None, ws_apart):
        ws lives in the tail of the flat buffer unless the statistics are float64 (ws_apart=True)."""
        n, _, h, w = x.shape
        c = self.model.nb_classes
        dtype = torch.float64 if self.do_var else torch.float32
        n_stats = steps_mod.McStatistics.blob_elements(n, c, h * w, self.do_mi, self.do_var)
        n_ws = n * c * h * w if (with_ws and dtype == torch.float32) else 0
        flat = torch.empty(n_stats + n_ws, device=x.device, dtype=dtype)
        stats = steps_mod.McStatistics(n, c, h, w, x.device, self.do_mi, self.do_var, blob=flat[:n_stats])
        ws = None
        if with_ws:
            ws = flat[n_stats:].view(n, c, h, w) if n_ws else torch.empty((n, c, h, w), device=x.device)
            ws.zero_()
        return flat, stats, ws, bool(with_ws and not n_ws)

    def ws_pass(self, x, ws_out):
        steps_mod.set_dropout_mode(self.model, False)
        ws_out.copy_(steps_mod.softmax(self.model(x)))

    def mc_pass(self, x, stats, masks=None):
        steps_mod.set_dropout_mode(self.model, True)
        try:
            self.model.forward_accumulate(x, stats, masks)
        finally:
            steps_mod.set_dropout_mode(self.model, False)

    def finalize(self, stats, count):
        return stats.finalize(self.do_mi, self.do_var, count=count)


class ShardedMcRunner:

    def __init__(self, model, mc_steps, ws_pass=True, rank=0, world=1, engine=None, do_mi=False, do_var=False,
                 root=0):
        self.engine = engine if engine is not None else HipEngine(model, do_mi, do_var)
        self.mc_steps = mc_steps
        self.ws_pass = ws_pass
        self.rank, self.world, self.root = rank, world, root
        self.jobs_per_step = mc_steps + (1 if ws_pass else 0)

    def job_list(self):
        """Job ids of one step: 0 = weight-scaling pass (when enabled), 1..T = MC passes."""
        return ([0] if self.ws_pass else []) + list(range(1, self.mc_steps + 1))

    def jobs_of(self, step, rank):
        jobs = self.job_list()
        return [j for i, j in enumerate(jobs) if (i + step * len(jobs)) % self.world == rank]

    def step(self, x, step_index=0, mask_sets=None):
        """One volume.  Returns the summary dict on the root rank (probabilities, entropy, ... and
        ws_probabilities when enabled), None elsewhere.  ``mask_sets``: optional list of T injected
        mask sets, indexed by MC pass."""
        flat, stats, ws, ws_apart = self.engine.buffers(x, self.ws_pass)
        for job in self.jobs_of(step_index, self.rank):
            if job == 0:
                self.engine.ws_pass(x, ws)
            else:
                self.engine.mc_pass(x, stats, None if mask_sets is None else mask_sets[job - 1])
        if self.world > 1:
            dist.reduce(flat, dst=self.root, op=dist.ReduceOp.SUM)          # statistics (+ ws) in one collective
            if ws_apart:
                dist.reduce(ws, dst=self.root, op=dist.ReduceOp.SUM)        # float64 statistics: ws travels apart
        if self.rank != self.root:
            return None
        out = self.engine.finalize(stats, self.mc_steps)
        if ws is not None:
            out['ws_probabilities'] = ws
        return out
