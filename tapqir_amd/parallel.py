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
    sub = CosmosDataset(
        data.images[lo:hi], data.xy[lo:hi], data.is_ontarget[lo:hi], mask=data.mask[lo:hi], labels=None,
        offset_samples=data.offset.samples, offset_weights=data.offset.weights, device=data.device,
        name=data.name, channels=data.channels)
    return sub, lo, data.images.shape[0]


def make_allreduce(group=None, async_op=False):
    """The single collective of a step: in-place sum of the cross-unit sums over all ranks.  With ``async_op`` the
    collective is left in flight and its handle returned: CosmosEngine.step then overlaps it with the next step's
    local guide sampling (full-batch steps)."""
    import torch.distributed as dist

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
    if world > 1 and model.run_path is not None:
        model.run_path = model.run_path / f"rank{rank}"
    return model
