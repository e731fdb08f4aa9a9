"""
AOI-sharded data parallelism for the cosmos step (new work: the reference has no distributed code
at all -- SURVEY.md section 2a / 8e).

One process per GPU.  AOIs (dimension Nt) are partitioned contiguously over ranks; every rank keeps
its AOIs' images, all per-AOI and per-(k,n,f) variational parameters and their Adam state.  Per step:

  1. every rank draws the SAME global latents (gain, pi, lamda, proximity): the Philox stream of the
     global sites is keyed by (seed, step) only;
  2. local latents use the GLOBAL unit index in their Philox key (``n_offset``), so a run gives the
     same draws however the AOIs are sharded;
  3. each rank evaluates its units; plate scales use the global Nt / nb;
  4. ONE all-reduce (sum) of ``gsum`` = [d/d gain, d/d cs, ELBO, (d/d rho, d/d a, d/d c) per dye]
     -- 6 doubles for one dye -- over RCCL (xGMI); latency-bound, so nothing is bucketed or ringed;
  5. every rank applies the identical Adam update to the replicated global parameters; local
     parameters need no communication.
"""

import torch

from tapqir_amd.utils.dataset import CosmosDataset


def shard_bounds(Nt, rank, world):
    """Contiguous AOI range [lo, hi) of ``rank``; sizes differ by at most one."""
    base, rem = divmod(Nt, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_dataset(data, rank, world):
    """The AOIs of ``rank`` as a CosmosDataset + (n_offset, Nt_global) for CosmosEngine."""
    lo, hi = shard_bounds(data.images.shape[0], rank, world)
    labels = None
    if data.labels is not None:
        # labels hold one row per ON-TARGET AOI, in AOI order (dataset.py: labels[: N]; simulate.py:124-131)
        on = data.is_ontarget.cpu()
        row = torch.cumsum(on.long(), 0) - 1
        labels = data.labels[row[lo:hi][on[lo:hi]].numpy()]
    sub = CosmosDataset(
        data.images[lo:hi], data.xy[lo:hi], data.is_ontarget[lo:hi], mask=data.mask[lo:hi], labels=labels,
        offset_samples=data.offset.samples, offset_weights=data.offset.weights, device=data.device,
        name=data.name, channels=data.channels)
    return sub, lo, data.images.shape[0]


class Collective:
    """The few blocking collectives a sharded fit needs OUTSIDE the step (initialisation from global data statistics,
    agreeing on the NaN-recovery branch and its new seed): tiny tensors, a handful of calls per fit."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def _dev(self, t):
        # the "nccl" backend (= RCCL) moves device tensors only
        return t.cuda() if self.dist.get_backend(self.group) == "nccl" and not t.is_cuda else t

    def reduce_(self, t, op="sum"):
        ops = {"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX}
        d = self._dev(t)
        self.dist.all_reduce(d, op=ops[op], group=self.group)
        if d is not t:
            t.copy_(d)
        return t

    def any(self, flag: bool) -> bool:
        return bool(self.reduce_(torch.tensor([1.0 if flag else 0.0]), "max")[0] > 0)

    def broadcast_int(self, value: int, src: int = 0) -> int:
        t = self._dev(torch.tensor([int(value)], dtype=torch.int64))
        self.dist.broadcast(t, src=src, group=self.group)
        return int(t[0])


def make_allreduce(group=None, async_op=False, direct=None):
    """The single collective of a step: in-place sum of the cross-unit sums over all ranks.

    ``direct`` (default: on when the process group's backend is "nccl", off with TAPQIR_AMD_RCCL_DIRECT=0): the all-reduce is
    ``ncclAllReduce`` of RCCL issued on the launch stream itself (``tapqir_amd.rccl.RcclDirect``): no second stream, no event
    hand-overs; the tail of the step then rides in the next step's sampling launch in stream order.  Otherwise
    ``torch.distributed.all_reduce``; with ``async_op`` the collective is left in flight and its handle returned:
    CosmosEngine.step then overlaps it with the next step's local guide sampling (full-batch steps)."""
    import os

    import torch.distributed as dist

    if direct is None:
        direct = dist.get_backend(group) == "nccl" and os.environ.get("TAPQIR_AMD_RCCL_DIRECT", "1") != "0"
    if direct:
        from tapqir_amd.rccl import RcclDirect

        inst = RcclDirect.checked(group)  # None (on every rank) if the direct path does not work here
        if inst is not None:
            return inst

    def allreduce(gsum: torch.Tensor):
        return dist.all_reduce(gsum, op=dist.ReduceOp.SUM, group=group, async_op=async_op) if async_op else \
            dist.all_reduce(gsum, op=dist.ReduceOp.SUM, group=group)

    return allreduce


def attach(model, group=None, async_op=True):
    """Turn a loaded (``model.load``) but not yet initialised model into one rank of an AOI-sharded fit: keep this
    rank's AOIs, key the RNG streams and plate scales by the global AOI indices, install the step's all-reduce, and give
    every rank its own checkpoint directory.  Call between ``model.load(path)`` and ``model.init(...)`` with
    ``torch.distributed`` initialised (backend "nccl" = RCCL on the GPUs; "gloo" works for rehearsals)."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sub, lo, Nt_global = shard_dataset(model.data, rank, world)
    model.data = sub
    model.engine = None
    model._engine_kwargs = dict(n_offset=lo, Nt_global=Nt_global)
    model.allreduce = make_allreduce(group, async_op=async_op)
    model.collective = Collective(group)
    if world > 1 and model.run_path is not None:
        # every rank keeps its own checkpoint AND writes its own statistics (its AOIs only): no file is shared
        model.run_path = model.run_path / f"rank{rank}"
        model.stats_path = model.path / f"rank{rank}"
    return model
