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
across consecutive volumes.  To let it do so the ranks must not meet at every volume: ``step_async``
issues the reduce on RCCL's own stream, finalises on a side stream of the root and hands back a
``PendingSummary``; the compute stream of every rank goes straight on to its share of the next
volume (T+1 = 21 jobs on 8 GPUs: 21 forward passes per rank per 8 volumes instead of 3 per volume).

Behind the drop-in scripts: ``ShardedMcPredictStep`` / ``ShardedEnsemblePredictionStep`` are the ``BatchStep`` forms of the runners;
``rcu_amd.scripts`` picks them when the script runs under ``python -m torch.distributed.run`` (``world_from_env``): every rank iterates
the same loader, the root alone assembles, evaluates and writes.  Masks are a function of (seed, batch, pass) and the statistics are
exact sums (rcu_amd.steps.McStatistics), so the files an N-rank run writes are the files the one-process run writes, byte for byte.
"""
import collections
import os

import torch
import torch.distributed as dist

from . import steps as steps_mod


class HipEngine:
    """The product engine: fused forward + softmax + accumulate on librcu_hip."""

    def __init__(self, model, do_mi=False, do_var=False, exact=True):
        self.model = model
        self.do_mi, self.do_var = do_mi, do_var
        # exact sums (rcu_amd.steps.McStatistics, include/rcu.h RCU_MC_EXACT): float64 planes whose additions are all exact -- the merged
        # statistics carry the same bits for every world size, job rotation, lane count and reduction tree of the collective
        self.exact = bool(exact)

    def _statistics(self, x, blob=None):
        n, _, h, w = x.shape
        return steps_mod.McStatistics(n, self.model.nb_classes, h, w, x.device, self.do_mi, self.do_var, blob=blob, exact=self.exact)

    def buffers(self, x, with_ws):
        """-> (flat reduce buffer, statistics object living in its head, ws tensor or None): the weight-scaling
        probabilities live in the tail of the ONE flat buffer (as float64 when the statistics are float64 -- a float32
        value is exact in float64, and a sum with the other ranks' zeros is too), so a volume is always one collective."""
        n, _, h, w = x.shape
        c = self.model.nb_classes
        dtype = steps_mod.McStatistics.dtype_of(self.do_var, self.exact)
        n_stats = steps_mod.McStatistics.blob_elements(n, c, h * w, self.do_mi, self.do_var, self.exact)
        n_ws = n * c * h * w if with_ws else 0
        flat = torch.empty(n_stats + n_ws, device=x.device, dtype=dtype)
        stats = self._statistics(x, flat[:n_stats])
        ws = None
        if with_ws:
            ws = flat[n_stats:].view(n, c, h, w)
            ws.zero_()
        return flat, stats, ws

    def reserve(self, x, mc_steps, group, lanes):
        """The canonical plans of the batch (rcu_amd.steps.reserve_canonical_plans): the bits of a pass must not depend on which launches
        THIS rank happens to make."""
        n, _, h, w = x.shape
        steps_mod.reserve_canonical_plans(self.model, n, h, w, mc_steps, group, lanes)

    def ws_pass(self, x, ws_out):
        steps_mod.set_dropout_mode(self.model, False)
        if ws_out.dtype == torch.float32 and ws_out.is_contiguous():
            steps_mod.softmax(self.model(x), out=ws_out)          # straight into the reduce buffer's tail
        else:                                                     # float64 statistics: the tail is float64 too
            ws_out.copy_(steps_mod.softmax(self.model(x)))

    def seeded_masks(self, x, seeds, first_sample=0):
        """The masks of the passes seeded with ``seeds`` over x, in the layout of their group launch (UNet.seeded_masks: one kernel, the factors
        of sample i in pass t a function of seeds[t] and the sample's global index ``first_sample + i`` alone)."""
        steps_mod.set_dropout_mode(self.model, True)
        try:
            return self.model.seeded_masks(x.shape[0], x.device, seeds, first_sample)
        finally:
            steps_mod.set_dropout_mode(self.model, False)

    def sample_masks(self, x, generator, passes=1):
        """Dropout factors of ``passes`` stochastic passes over x (rows [site][passes * N][C_site]) from ``generator``."""
        steps_mod.set_dropout_mode(self.model, True)
        try:
            return self.model.sample_masks(x.shape[0] * passes, x.device, generator=generator)
        finally:
            steps_mod.set_dropout_mode(self.model, False)

    def mc_pass(self, x, stats, masks=None, passes=1, lane=0):
        steps_mod.set_dropout_mode(self.model, True)
        try:
            self.model.forward_accumulate(x, stats, masks, passes=passes, lane=lane)
        finally:
            steps_mod.set_dropout_mode(self.model, False)

    def member_pass(self, member, x, stats, lane=0):
        steps_mod.set_dropout_mode(member, False)
        member.forward_accumulate(x, stats, lane=lane)

    def side_statistics(self, x):
        """Fresh (zeroed) statistics for a stream lane of its own; ``merge`` adds them into the volume's statistics."""
        return self._statistics(x)

    def merge(self, stats, side):
        stats.blob.add_(side.blob)          # plain sums (include/rcu.h, rcu_mc_*)

    def finalize(self, stats, count):
        return stats.finalize(self.do_mi, self.do_var, count=count)

    def ws_outputs(self, ws):
        return {'ws_probabilities': ws if ws.dtype == torch.float32 else ws.float()}


class AleatoricHipEngine(HipEngine):
    """EXTENSION (BASELINE config "aleatoric + MC", see rcu_amd.steps.AleatoricMcPredictStep): passes of a sigma-head U-Net.  The
    per-pass sigmas are plain sums like the statistics, so they ride in the same reduce buffer:
    flat = [statistics | sigma sum [n,C,H,W] | ws probabilities | ws sigma]."""

    def __init__(self, model, is_log_sigma=False, do_mi=False):
        # float32 statistics: the sigma sums share the buffer (unbounded addends: no exact form), one dtype per collective
        super().__init__(model, do_mi, False, exact=False)
        if not getattr(model, 'sigma_out', False):
            raise ValueError('AleatoricHipEngine needs a model built with sigma_out=True')
        self.is_log_sigma = is_log_sigma

    def buffers(self, x, with_ws):
        n, _, h, w = x.shape
        c = self.model.nb_classes
        n_stats = steps_mod.McStatistics.blob_elements(n, c, h * w, self.do_mi, False)
        n_vol = n * c * h * w
        flat = torch.empty(n_stats + n_vol * (3 if with_ws else 1), device=x.device, dtype=torch.float32)
        stats = steps_mod.McStatistics(n, c, h, w, x.device, self.do_mi, False, blob=flat[:n_stats])
        flat[n_stats:].zero_()
        stats.sigma_sum = flat[n_stats:n_stats + n_vol].view(n, c, h, w)
        ws = flat[n_stats + n_vol:].view(2, n, c, h, w) if with_ws else None
        return flat, stats, ws

    def ws_pass(self, x, ws_out):
        steps_mod.set_dropout_mode(self.model, False)
        logits, raw = self.model(x)
        n, c, h, w = logits.shape
        lib = steps_mod._lib
        lib.check(lib.load().rcu_aleatoric(lib.ptr(logits), lib.ptr(raw.contiguous()), n, h * w, c, int(self.is_log_sigma),
                                           lib.ptr(ws_out[0]), lib.ptr(ws_out[1]), None, None, lib.current_stream()))

    def mc_pass(self, x, stats, masks=None, passes=1, lane=0):
        steps_mod.set_dropout_mode(self.model, True)
        try:
            self.model.forward_accumulate_sigma(x, stats, stats.sigma_sum, masks, self.is_log_sigma, lane=lane, passes=passes)
        finally:
            steps_mod.set_dropout_mode(self.model, False)

    def side_statistics(self, x):
        stats = super().side_statistics(x)
        stats.sigma_sum = torch.zeros((x.shape[0], self.model.nb_classes) + tuple(x.shape[2:]), device=x.device)
        return stats

    def merge(self, stats, side):
        super().merge(stats, side)
        stats.sigma_sum.add_(side.sigma_sum)

    def finalize(self, stats, count):
        out = stats.finalize(self.do_mi, False, count=count)
        out['sigma'] = stats.sigma_sum / float(max(count, 1))
        return out

    def ws_outputs(self, ws):
        return {'ws_probabilities': ws[0], 'ws_sigma': ws[1]}


job_seed = steps_mod.job_seed      # seed of a torch generator for the masks of MC pass ``job`` of volume ``step_index`` (engines without the library's draw)
pass_seed = steps_mod.pass_seed    # key of the library's counter-based draw for MC pass ``job``: the counter carries the sample's global index


class ShardedMcRunner:
    """``seed``: base seed of the dropout masks.  The masks of MC pass j over the samples of volume / batch k are drawn under the key
    ``pass_seed(seed, j)`` at the counters of the samples' global indices -- ``sample_offsets[k]`` (what the script's loop counted) or k x n on
    (an engine with ``seeded_masks``: the library's own counter-based draw, include/rcu.h rcu_dropout_masks; else from a torch generator seeded
    with ``job_seed(seed, k, j)``), so they do not depend on the rank that runs the pass nor on the world size: every world size
    aggregates the same T samples (ranks that are all seeded alike, as the reference's ``do_seed`` does with
    ``config.seed``, would otherwise draw the same mask sequence on every rank and the T passes would hold only about T / world
    distinct samples).  ``seed=None`` draws from the device's default generator after seeding it per rank once.
    ``pass_group``: MC passes of one rank run ``pass_group`` at a time as one batch (rcu_unet_forward_accumulate_passes).
    ``lanes``: the launches of a volume go to ``lanes`` HIP streams in turn -- one activation workspace and one statistics blob per
    lane, the side lanes' statistics added into the volume's before the reduce.  Consecutive layers of one forward pass depend on
    each other, so a stream has nothing to run while a layer's last workgroups finish and the next layer's start; the passes are
    independent, and a second lane fills those gaps (tools/stream_overlap_probe.py: 6.63 -> 6.27 ms per pass with pass pairs).  The
    assignment launch -> lane is fixed, so the result does not depend on timing."""

    def __init__(self, model, mc_steps, ws_pass=True, rank=0, world=1, engine=None, do_mi=False, do_var=False,
                 root=0, seed=0, pass_group=1, lanes=1, ws_transport=None, force_exchange=False, exact=True):
        self.engine = engine if engine is not None else HipEngine(model, do_mi, do_var, exact and mc_steps <= steps_mod._lib.RCU_MC_EXACT_MAX_PASSES)
        # force_exchange: run the exchange step (reduce / send-recv, asynchronous work handles, side-stream finalize) at world size 1 too --
        # the rehearsal of the RCCL path on a box with ONE GPU (tools/rccl_world1_rehearsal.py; needs an initialised process group).  A
        # sum-reduce over one rank leaves the buffer as it is, so the result carries the bits of the plain world-1 step.
        self.force_exchange = bool(force_exchange)
        # How the weight-scaling probabilities reach the root: 'reduce' -- in the tail of the ONE reduce buffer (zeros on every rank
        # but their owner; one collective per volume, twice the bytes on every link) -- or 'p2p': the reduce carries the statistics
        # only and the owner of job 0 sends its tail to the root (nothing when the root owns it): half the bytes on the links the
        # send does not use.  None (the default, round 6): 'p2p' where the backend has device send / recv -- RCCL ("nccl") -- so that the one
        # collective of a volume carries the 63 MB of statistics and nothing else (SURVEY 8e budgets 31.5 MB of float32 sums; the exact float64
        # sums are twice that; the tail would double it again, zeros from every rank but one); 'reduce' on gloo.  Resolved at the first exchange.
        if ws_transport not in (None, 'reduce', 'p2p'):
            raise ValueError('ws_transport must be "reduce" or "p2p"')
        self.ws_transport = ws_transport
        self.p2p_messages = 0          # send / recv pairs this rank took part in (0 whenever the root owns the weight-scaling pass)
        self._p2p_checked = False
        self.mc_steps = mc_steps
        self.ws_pass = ws_pass
        self.rank, self.world, self.root = rank, world, root
        self.jobs_per_step = mc_steps + (1 if ws_pass else 0)
        self.seed = seed
        self.pass_group = max(1, int(pass_group))
        # group_samples (optional attribute): samples per launch instead of passes per launch -- a step over a batch of n images then groups
        # group_samples // n passes (a runner that is handed one volume or several consecutive ones keeps its launches the same size)
        self.group_samples = None
        self.lanes = max(1, int(lanes))
        self._generator = None
        self.sample_offsets = {}       # step index -> global index of the batch's first sample (the sharded predict steps fill it in; else k x n)
        self.forwards_run = 0          # launches of this rank (a pass group counts its passes)
        self.reserve_plans = True      # (the ensemble runner: every launch is one member on n samples, nothing to make canonical)

    def masks_of(self, x, step_index, job):
        """The device mask tensor MC pass ``job`` (1..T) of volume ``step_index`` runs under (None without a seed / an engine
        that samples on its own)."""
        seeded = getattr(self.engine, 'seeded_masks', None)
        if self.seed is not None and seeded is not None:
            return seeded(x, [pass_seed(self.seed, job)], self.first_sample(x, step_index))
        sample = getattr(self.engine, 'sample_masks', None)
        if self.seed is None or sample is None:
            return None
        if self._generator is None:
            self._generator = torch.Generator(device=x.device)
        self._generator.manual_seed(job_seed(self.seed, step_index, job))
        return sample(x, self._generator)

    def first_sample(self, x, step_index):
        """Global index of the first sample of step ``step_index``'s batch x."""
        offset = self.sample_offsets.get(step_index)
        return int(step_index) * int(x.shape[0]) if offset is None else int(offset)

    def _ws_outputs(self, ws):
        hook = getattr(self.engine, 'ws_outputs', None)
        return hook(ws) if hook is not None else {'ws_probabilities': ws}

    def job_list(self):
        """Job ids of one step: 0 = weight-scaling pass (when enabled), 1..T = MC passes."""
        return ([0] if self.ws_pass else []) + list(range(1, self.mc_steps + 1))

    def jobs_of(self, step, rank):
        jobs = self.job_list()
        return [j for i, j in enumerate(jobs) if (i + step * len(jobs)) % self.world == rank]

    def _run_job(self, job, x, stats, ws, mask_sets, step_index=0, lane=0):
        if job == 0:
            self.engine.ws_pass(x, ws)
        else:
            masks = self.masks_of(x, step_index, job) if mask_sets is None else mask_sets[job - 1]
            if lane:
                self.engine.mc_pass(x, stats, masks, lane=lane)
            else:
                self.engine.mc_pass(x, stats, masks)
        self.forwards_run += 1

    def _lane_of(self, job, count):
        """Fixed lane of a job, or None: the launches of a volume take the lanes in turn."""
        return None

    def _run_jobs(self, x, step_index, mask_sets):
        flat, stats, ws = self.engine.buffers(x, self.ws_pass)
        jobs = self.jobs_of(step_index, self.rank)
        # stream lanes (rcu_amd.steps.StreamLanes): lane 0 = the caller's stream and the volume's statistics
        lanes = steps_mod.StreamLanes(x.device, self.lanes if (x.is_cuda and hasattr(self.engine, 'side_statistics')) else 1)
        pass_group = self.pass_group if self.group_samples is None else max(1, int(self.group_samples) // int(x.shape[0]))
        if hasattr(self.engine, 'reserve') and self.reserve_plans:
            self.engine.reserve(x, self.mc_steps, pass_group, lanes.count)
        lanes.begin(stats, lambda: self.engine.side_statistics(x), inputs=(x,), first=step_index if self.world > 1 else 0)
        on_lane = lanes.run

        # group sizes of this rank's MC passes: rounds of one group per lane (steps.balanced_groups), so that the lanes carry the same load
        sizes = steps_mod.balanced_groups(sum(1 for j in jobs if j != 0), pass_group, lanes.count)
        i = 0
        while i < len(jobs):
            group = [j for j in jobs[i:i + sizes[0]] if j != 0] if jobs[i] != 0 else []
            if group:
                sizes.pop(0)
            if len(group) > 1:     # consecutive MC passes of this rank as one batch of N * g samples
                def run_group(st, lane, group=group):
                    if mask_sets is None:
                        seeded = getattr(self.engine, 'seeded_masks', None)
                        if self.seed is not None and seeded is not None:
                            ms = seeded(x, [pass_seed(self.seed, j) for j in group], self.first_sample(x, step_index))
                        else:
                            ms = [self.masks_of(x, step_index, j) for j in group]
                            ms = None if any(m is None for m in ms) else ms
                    else:
                        ms = [mask_sets[j - 1] for j in group]
                    if lane:
                        self.engine.mc_pass(x, st, ms, passes=len(group), lane=lane)
                    else:
                        self.engine.mc_pass(x, st, ms, passes=len(group))
                on_lane(run_group)
                self.forwards_run += len(group)
                i += len(group)
            elif jobs[i] == 0:     # the weight-scaling pass writes into the volume's buffer: lane 0, outside the rotation
                self._run_job(0, x, stats, ws, mask_sets, step_index)
                i += 1
            else:
                on_lane(lambda st, lane, job=jobs[i]: self._run_job(job, x, st, ws, mask_sets, step_index, lane),
                        self._lane_of(jobs[i], lanes.count))
                i += 1
        if lanes.count > 1:
            lanes.end(self.engine.merge)
        return flat, stats, ws

    def ws_owner(self, step_index):
        """Rank that runs the weight-scaling pass (job 0) of volume ``step_index``."""
        return (step_index * self.jobs_per_step) % self.world

    def _exchange(self, flat, ws, step_index, async_op):
        """The data exchange of one volume -> list of work handles (empty: everything completed).  'reduce': ONE sum-reduce of
        [statistics | ws].  'p2p': the sum-reduce covers the statistics (everything in front of the ws tail) and the owner of the
        weight-scaling pass sends the tail to the root."""
        if self.ws_transport is None:      # the default: point to point where the backend can (RCCL), else inside the reduce
            self.ws_transport = 'p2p' if (dist.get_backend() == 'nccl' or not flat.is_cuda) else 'reduce'
            self._p2p_checked = True
        if ws is None or self.ws_transport == 'reduce':
            w = dist.reduce(flat, dst=self.root, op=dist.ReduceOp.SUM, async_op=async_op)
            return [w] if async_op else []
        if not self._p2p_checked:
            # gloo has no send / recv for device tensors (it would fail, or exchange garbage, on the eight-ranks-on-one-GPU-over-gloo
            # rehearsal): the tail then rides in the reduce, which every backend can do
            self._p2p_checked = True
            if flat.is_cuda and dist.get_backend() != 'nccl':
                import warnings
                warnings.warn('ws_transport="p2p" needs device send / recv, which the {} backend does not have: falling back to '
                              '"reduce" (the tail rides in the one sum-reduce)'.format(dist.get_backend()))
                self.ws_transport = 'reduce'
                return self._exchange(flat, ws, step_index, async_op)
        head = flat[:flat.numel() - ws.numel()]
        tail = flat[flat.numel() - ws.numel():]
        works = [dist.reduce(head, dst=self.root, op=dist.ReduceOp.SUM, async_op=async_op)]
        owner = self.ws_owner(step_index)
        if owner != self.root:
            if self.rank == owner:
                works.append(dist.isend(tail, dst=self.root) if async_op else dist.send(tail, dst=self.root))
                self.p2p_messages += 1
            elif self.rank == self.root:
                works.append(dist.irecv(tail, src=owner) if async_op else dist.recv(tail, src=owner))
                self.p2p_messages += 1
        return [w for w in works if w is not None] if async_op else []

    def step(self, x, step_index=0, mask_sets=None):
        """One volume.  Returns the summary dict on the root rank (probabilities, entropy, ... and
        ws_probabilities when enabled), None elsewhere.  ``mask_sets``: optional list of T injected
        mask sets, indexed by MC pass."""
        flat, stats, ws = self._run_jobs(x, step_index, mask_sets)
        if self.world > 1 or self.force_exchange:
            for w in self._exchange(flat, ws, step_index, async_op=False):
                w.wait()
        if self.rank != self.root:
            return None
        out = self.engine.finalize(stats, self.mc_steps)
        if ws is not None:
            out.update(self._ws_outputs(ws))
        return out

    def step_async(self, x, step_index=0, mask_sets=None, depth=2):
        """Like ``step`` but does not make this rank's compute stream wait for the other ranks: returns a
        ``PendingSummary`` whose ``result()`` is the summary dict on the root (None elsewhere).  At most
        ``depth`` reduces stay in flight per rank; their buffers are kept alive until they completed."""
        if self.world == 1 and not self.force_exchange:
            return PendingSummary(self.step(x, step_index, mask_sets))
        if not hasattr(self, '_inflight'):
            self._inflight = collections.deque()
        self._side_stream(x.device)
        while len(self._inflight) >= depth:
            self._inflight.popleft().retire()
        flat, stats, ws = self._run_jobs(x, step_index, mask_sets)
        works = self._exchange(flat, ws, step_index, async_op=True)
        pending = PendingSummary(None, works=works, keep=(flat, ws))
        if self.rank == self.root:
            if getattr(self, '_side', None) is not None:
                with torch.cuda.stream(self._side):
                    for w in works:
                        w.wait()                       # the SIDE stream waits for RCCL's stream, compute does not
                    out = self.engine.finalize(stats, self.mc_steps)
                    if ws is not None:      # (out of the reduce buffer's tail -- a float64 tail is converted here, BEHIND the collective)
                        out.update(self._ws_outputs(ws))
                    pending.ready = torch.cuda.Event()
                    pending.ready.record(self._side)
                flat.record_stream(self._side)
                pending.works = []
            else:
                for w in works:
                    w.wait()
                out = self.engine.finalize(stats, self.mc_steps)
                if ws is not None:
                    out.update(self._ws_outputs(ws))
                pending.works = []
            pending.value = out
        self._inflight.append(pending)
        return pending

    def reduce_async(self, x, step_index=0, mask_sets=None, depth=2):
        """The step-seam form: this rank's jobs of the batch and the exchange, WITHOUT the finalize.  Root -> (``PendingStatistics``, None):
        the merged statistics once the collective has completed -- ``MultiPredictionSummary`` finalises them on a side stream that waits for
        the collective, so the root's compute stream goes straight on to the next batch's passes like every other rank's.  Other ranks ->
        (None, None): the collective stays in flight (at most ``depth`` per rank, their buffers kept alive)."""
        if not hasattr(self, '_inflight'):
            self._inflight = collections.deque()
        while len(self._inflight) >= depth:
            self._inflight.popleft().retire()
        flat, stats, ws = self._run_jobs(x, step_index, mask_sets)
        works = self._exchange(flat, ws, step_index, async_op=True) if (self.world > 1 or self.force_exchange) else []
        if self.rank != self.root:
            self._inflight.append(PendingSummary(None, works=works, keep=(flat, ws)))
            return None, None
        if not isinstance(stats, steps_mod.McStatistics):      # (an engine of the CPU tests: plain tensors, nothing to finalise later)
            for w in works:
                w.wait()
            stats.count = self.mc_steps
            return stats, (self._ws_outputs(ws) if ws is not None else None)
        return PendingStatistics(self, stats, ws, works, flat), None

    def _side_stream(self, device):
        if device.type != 'cuda':
            return None
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def drain(self):
        """Retire every reduce still in flight (call before destroying the process group)."""
        while getattr(self, '_inflight', None):
            self._inflight.popleft().retire()


class ShardedAleatoricMcRunner(ShardedMcRunner):
    """ShardedMcRunner over AleatoricHipEngine (BASELINE config "BraTS aleatoric + MC: sigma-head U-Net, T = 50, samples sharded
    over 8 MI355X"): the summary gains ``sigma`` (mean over the passes) and ``ws_sigma``."""

    def __init__(self, model, mc_steps, is_log_sigma=False, ws_pass=True, rank=0, world=1, do_mi=False, root=0, seed=0, lanes=1,
                 pass_group=1):
        super().__init__(model, mc_steps, ws_pass=ws_pass, rank=rank, world=world,
                         engine=AleatoricHipEngine(model, is_log_sigma, do_mi), do_mi=do_mi, root=root, seed=seed, lanes=lanes,
                         pass_group=pass_group)


class ShardedEnsembleRunner(ShardedMcRunner):
    """K ensemble members instead of T MC passes (bin-dl/brats_test_ensemble.py:78-94; BASELINE config
    "K=10 checkpoints, members sharded over 8 MI355X with RCCL reduce").  Job j = member j-1 in eval mode, no
    weight-scaling pass, divisor K.  A member's packed weights are 35 MB, so every rank holds all K members and
    the job rotation of the base class applies (K = 10 on 8 GPUs: 10 forwards per rank per 8 volumes instead of
    a static 2,2,1,1,1,1,1,1 split).  ``share_workspace``: members 2..K borrow the activation workspace of the first (one per
    stream lane; rcu_unet_create_with, include/rcu.h) -- K members cost lanes x 6.1 GB + K x 35 MB at 160 slices, not K x 6.1 GB."""

    def __init__(self, members, rank=0, world=1, engine=None, do_mi=False, do_var=False, root=0, lanes=1, share_workspace=True,
                 exact=True):
        members = list(members)
        super().__init__(members[0] if members else None, len(members), ws_pass=False, rank=rank, world=world,
                         engine=engine, do_mi=do_mi, do_var=do_var, root=root, lanes=lanes, exact=exact)
        self.members = members
        self.reserve_plans = False
        if share_workspace:
            steps_mod.share_member_workspaces(members)

    def _lane_of(self, job, count):
        # A member keeps to one lane whatever the job rotation of a multi-rank run does: one plan (35 MB of packed weights) per member, and
        # the members that share a lane's workspace run on that lane's stream, one after the other
        return (job - 1) % count

    def _run_job(self, job, x, stats, ws, mask_sets, step_index=0, lane=0):
        if lane:
            self.engine.member_pass(self.members[job - 1], x, stats, lane=lane)
        else:
            self.engine.member_pass(self.members[job - 1], x, stats)
        self.forwards_run += 1


class PendingStatistics:
    """What a sharded predict step leaves under ``multi_probabilities`` on the root: the merged statistics of the batch, valid once the
    collective has completed.  ``MultiPredictionSummary`` calls ``finalize_when_merged``: the finalize (and the hand-over of the
    weight-scaling outputs out of the reduce buffer's tail) run on a side stream that waits for the collective; the compute stream does not."""

    def __init__(self, runner, stats, ws, works, flat):
        self.runner, self.stats, self.ws, self.works, self.flat = runner, stats, ws, list(works), flat

    def finalize_when_merged(self, do_mi=False, do_var=False):
        """-> (dict of the summary's outputs incl. the weight-scaling ones, event recorded behind them or None)."""
        stats, runner = self.stats, self.runner
        if (do_mi and not stats.do_mi) or (do_var and not stats.do_var):
            raise ValueError('the sharded predict step did not track {}: construct it with the flags of the summary'.format(
                'the entropy sum (do_mi)' if do_mi and not stats.do_mi else 'the squared sums (do_var)'))
        side = runner._side_stream(stats.blob.device)

        def finish():
            for w in self.works:
                w.wait()            # RCCL: the CURRENT (side) stream waits for the collective's stream; gloo: the host does
            out = stats.finalize(do_mi, do_var, count=runner.mc_steps)
            if self.ws is not None:
                out.update(runner._ws_outputs(self.ws))
            return out

        if side is None:
            return finish(), None
        side.wait_stream(torch.cuda.current_stream(stats.blob.device))      # this rank's own passes wrote the buffer on the compute stream
        with torch.cuda.stream(side):
            out = finish()
            event = torch.cuda.Event()
            event.record(side)
        self.flat.record_stream(side)
        self.works = []
        return out, event


class PendingSummary:
    """Result of ``ShardedMcRunner.step_async``."""

    def __init__(self, value, works=(), keep=()):
        self.value, self.works, self.keep = value, list(works), keep
        self.ready = None

    def retire(self):
        """Order the calling rank's current stream after the collective and drop the buffers."""
        for w in self.works:
            w.wait()
        self.works = []
        if self.ready is not None:
            torch.cuda.current_stream().wait_event(self.ready)
            for t in (self.value or {}).values():
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(torch.cuda.current_stream())
            self.ready = None
        self.keep = ()

    def result(self):
        self.retire()
        return self.value


# ------------------------------------------------------------------------------------------------------------------------------
# the runners as batch steps of the drop-in scripts
# ------------------------------------------------------------------------------------------------------------------------------
class World:
    """This process's place in a ``torch.distributed.run`` launch (RANK / WORLD_SIZE / LOCAL_RANK)."""

    def __init__(self, rank=0, world=1, local_rank=0, device='cuda', backend=None):
        self.rank, self.world, self.local_rank, self.device, self.backend = rank, world, local_rank, device, backend

    @property
    def is_root(self):
        return self.rank == 0


def world_from_env(device='cuda', backend=None):
    """-> ``World``.  With WORLD_SIZE > 1 in the environment (a ``python -m torch.distributed.run`` launch) the process group is
    initialised here: ``nccl`` (= RCCL over xGMI) with one GPU per rank -- the rank's device is ``cuda:LOCAL_RANK`` -- or, when the node
    has fewer GPUs than ranks (the two-ranks-on-one-GPU rehearsal of the tests), ``gloo`` with every rank on the visible devices in turn.
    Without WORLD_SIZE: rank 0 of a world of one, no process group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return World(device=device)
    rank, local = int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    n_dev = torch.cuda.device_count()
    if str(device).startswith('cuda'):
        if n_dev < 1:
            raise RuntimeError('rcu_amd needs a GPU (librcu_hip); there is no CPU fallback')
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
        if backend is None:
            backend = 'nccl' if n_dev >= local_world else 'gloo'
        device = 'cuda:{}'.format(local % n_dev)
        torch.cuda.set_device(torch.device(device))
    elif backend is None:
        backend = 'gloo'
    if not dist.is_initialized():
        # (no device_id=: the communicator is created at the first collective; eager creation costs 2-3.5 % per step on this image)
        dist.init_process_group(backend)
    return World(rank, world, local, device, backend)


class _ShardedStepBase(steps_mod.BatchStep):
    """What the sharded predict steps share: a runner per model, bounded run-ahead, the root's hand-over to MultiPredictionSummary."""
    MAX_AHEAD = 3       # batches a rank's host may enqueue ahead of its GPU (the buffers of every batch in flight stay allocated)

    def __init__(self, world, do_mi, do_var, lanes, exact, engine_factory=None):
        super().__init__()
        self.rank, self.world = world.rank, world.world
        self.engine_factory = engine_factory      # tests: model -> engine (the CPU tests of the step seam run an oracle-backed engine)
        self.do_mi, self.do_var = do_mi, do_var
        self.lanes = steps_mod.McPredictStep.LANES if lanes is None else max(1, int(lanes))
        self.exact = bool(exact)
        self._runner, self._runner_key = None, None
        self._events = collections.deque()
        self._batches = 0

    def _engine(self, model):
        return None if self.engine_factory is None else self.engine_factory(model)

    def finish(self):
        """Retire the collectives still in flight (before the process group goes away)."""
        if self._runner is not None:
            self._runner.drain()
            import logging
            logging.info('rank {} of {}: {} forward passes in {} batches'.format(self.rank, self.world, self._runner.forwards_run, self._batches))
        self._events.clear()

    def _throttle(self, device):
        if device.type != 'cuda':
            return
        ev = torch.cuda.Event()
        ev.record()
        self._events.append(ev)
        while len(self._events) > self.MAX_AHEAD:
            self._events.popleft().synchronize()

    def _hand_over(self, batch_context, runner, x, step_index, mask_sets=None):
        """This rank's jobs of the batch + the exchange.  Root: the merged statistics go under ``multi_probabilities`` (the compute stream
        waits for the collective), the weight-scaling probabilities under ``ws_probabilities``; other ranks: ``None`` there
        (MultiPredictionSummary then has nothing to do) and the reduce stays in flight behind the next batch's passes."""
        offsets = getattr(runner, 'sample_offsets', None)
        if offsets is not None:       # the seeded masks are keyed by the samples' global indices, as the one-process step keys them
            offsets[step_index] = steps_mod.first_sample_of(batch_context, x.shape[0])
            offsets.pop(step_index - 64, None)
        stats, ws = runner.reduce_async(x, step_index, mask_sets)
        self._batches += 1
        batch_context.output['multi_probabilities'] = stats      # root: PendingStatistics (the summary finalises it on a side stream); else None
        if ws is not None:
            batch_context.output.update(ws)
        self._throttle(x.device)


class ShardedMcPredictStep(_ShardedStepBase):
    """``McPredictStep`` with the T stochastic passes (and the weight-scaling pass) of every batch sharded over the ranks of the process
    group (``ShardedMcRunner``): rechun/dl/customsteps.py:10-39 behind bin-dl/brats_test_default.py:39,46-54 on N GPUs.  Every rank calls
    the step with the SAME batch; the root's batch context receives what ``McPredictStep`` leaves there."""

    def __init__(self, mc_steps, world, do_mi=False, do_var=False, masks=None, ws_pass=True, group_pixels=None, lanes=None, seed=0,
                 exact=True, ws_transport=None, engine_factory=None) -> None:
        super().__init__(world, do_mi, do_var, lanes, exact, engine_factory)
        if seed is None:
            raise ValueError('a sharded MC step needs a seed: the masks of a pass must not depend on the rank that runs it')
        self.mc_steps, self.masks, self.ws_pass, self.seed = mc_steps, masks, ws_pass, seed
        self.group_pixels = steps_mod.McPredictStep.GROUP_PIXELS if group_pixels is None else group_pixels
        self.ws_transport = ws_transport

    def __call__(self, batch_context, task_context, context) -> None:
        steps_mod._check_context(context)
        images = steps_mod._images_to_device(batch_context, context)
        model = context.model
        n, _, h, w = images.shape
        group = steps_mod.pass_group_size(model, n, h, w, self.group_pixels)
        group = max(1, min(group, self.mc_steps))
        key = id(model)
        if self._runner_key != key:
            if self._runner is not None:      # another model: retire the old runner's collectives before it goes
                self._runner.drain()
            self._runner = ShardedMcRunner(model, self.mc_steps, ws_pass=self.ws_pass, rank=self.rank, world=self.world, do_mi=self.do_mi,
                                           do_var=self.do_var, seed=self.seed, pass_group=group, lanes=self.lanes,
                                           ws_transport=self.ws_transport, exact=self.exact, engine=self._engine(model))
            self._runner_key = key
        # the pass group follows the batch (a smaller last batch, a coalesced one): ONE runner keeps its collectives in flight, its counters
        # and its side stream through the run
        self._runner.pass_group = group
        self._hand_over(batch_context, self._runner, images, batch_context.batch_index, self.masks)


class ShardedEnsemblePredictionStep(_ShardedStepBase):
    """``EnsemblePredictionStep`` (bin-dl/brats_test_ensemble.py:72-94) with the K members of every batch sharded over the ranks
    (``ShardedEnsembleRunner``: every rank holds all members -- 107 MB of packed weights each, one shared workspace per lane)."""

    def __init__(self, additional_models, world, do_mi=False, do_var=False, lanes=None, exact=True, share_workspace=True,
                 engine_factory=None) -> None:
        super().__init__(world, do_mi, do_var, lanes, exact, engine_factory)
        self.additional_models = additional_models
        self.share_workspace = share_workspace

    def __call__(self, batch_context, task_context, context) -> None:
        steps_mod._check_context(context)
        images = steps_mod._images_to_device(batch_context, context)
        members = [context.model] + list(self.additional_models)
        key = tuple(id(m) for m in members)
        if self._runner_key != key:
            self._runner = ShardedEnsembleRunner(members, rank=self.rank, world=self.world, do_mi=self.do_mi, do_var=self.do_var,
                                                 lanes=min(self.lanes, len(members)), share_workspace=self.share_workspace, exact=self.exact,
                                                 engine=self._engine(members[0]))
            self._runner_key = key
        self._hand_over(batch_context, self._runner, images, batch_context.batch_index)
