"""
AOI-sharded data parallelism on the GPU: two rank processes drive the HIP library (libtapqir_hip.so, not the host
build) through the sharded launch sequence -- sampling split around the in-flight all-reduce, tq_cosmos_tail_reduced
carried by the next step's sampling launch -- and must reproduce the single-process fit.

  * backend "nccl" (= RCCL): one GPU per rank; skipped on a box with fewer than two GPUs.  The step's all-reduce is then
    ``ncclAllReduce`` issued on the launch stream (tapqir_amd/rccl.py) -- that path also runs with ONE rank on a one-GPU box
    (test_rccl_direct_single_rank: communicator from a broadcast unique id, collective in stream order, the tail of a step
    inside the next step's sampling launch);
  * backend "gloo": both ranks share cuda:0 and the 48-byte all-reduce is staged through the host -- the same kernels,
    launch order and stream semantics, runnable on a one-GPU box.
"""

import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, backend, minibatch):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import make_dataset, make_oracle, oracle_to_engine
    from tapqir_amd import _lib
    from tapqir_amd.models.engine import CosmosEngine
    from tapqir_amd.parallel import make_allreduce, shard_dataset

    K, N, F = 2, 6, 5
    d = make_dataset(N=N, F=F, K=K)
    o = make_oracle(d, K, perturb=0.2)
    sub, n_off, Nt_global = shard_dataset(d, rank, world)
    eng = CosmosEngine(sub, K=K, device=dev, seed=5, n_offset=n_off, Nt_global=Nt_global)
    assert eng.lib is _lib.load()  # the HIP library, not a host build
    full = CosmosEngine(d, K=K, device=dev, seed=5)
    oracle_to_engine(o, full)
    fv, sv = full.named("params"), eng.named("params")
    lo, hi = n_off, n_off + sub.images.shape[0]
    for n in sv:
        sv[n].copy_(fv[n][:, lo:hi] if sv[n].dim() == 4 else (fv[n][lo:hi] if sv[n].dim() == 3 else fv[n]))
    allreduce = make_allreduce(async_op=True)
    subs = [([0, 2], [1, 3, 4]), ([1, 2], [0, 2]), ([0, 1], [0, 1, 4]), ([0, 2], [2, 3])] if minibatch else [(None, None)] * 4
    t = lambda v: None if v is None else torch.tensor(v)
    for ndx, fdx in subs:
        eng.step(t(ndx), t(fdx), allreduce=allreduce)
    pending = eng._pending is not None
    eng.join()
    torch.cuda.synchronize()
    out = {"rank": rank, "lo": lo, "hi": hi, "elbo": float(eng.elbo_out[0]), "pending": pending,
           "ranks": dist.get_world_size(), "backend": dist.get_backend(), "allreduce": getattr(allreduce, "backend", "torch"),
           "params": {n: v.detach().cpu().clone().numpy() for n, v in eng.named("params").items()}}
    if rank == 0:
        per = N // world
        for ndx, fdx in subs:
            gn = None if ndx is None else [r * per + j for r in range(world) for j in ndx]
            full.step(t(gn), t(fdx))
        full.join()
        torch.cuda.synchronize()
        out["full_elbo"] = float(full.elbo_out[0])
        out["full_params"] = {n: v.detach().cpu().clone().numpy() for n, v in full.named("params").items()}
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("minibatch", [False, True], ids=["full_batch_overlapped", "minibatch_lazy_adam"])
@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_two_gpu_ranks_equal_single_process(backend, minibatch):
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank; this box has one")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 17 * int(minibatch) + 31 * (backend == "gloo")) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend, minibatch)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=480) for _ in range(world)], key=lambda o: o["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = outs[0]
    assert all(o["ranks"] == 2 and o["backend"] == backend for o in outs)
    if not minibatch:
        assert all(o["pending"] for o in outs)  # the last step's tail was waiting for its collective
    for o in outs:
        assert abs(o["elbo"] - ref["full_elbo"]) <= 2e-6 * abs(ref["full_elbo"])
        lo, hi = o["lo"], o["hi"]
        for n, v in o["params"].items():
            fp = ref["full_params"][n]
            want = fp[:, lo:hi] if v.ndim == 4 else (fp[lo:hi] if v.ndim == 3 else fp)
            assert abs(v - want).max() <= 5e-6, n


@pytest.mark.timeout(600)
@pytest.mark.parametrize("minibatch", [False, True], ids=["full_batch_in_stream", "minibatch_lazy_adam"])
def test_rccl_direct_single_rank(minibatch):
    """One rank, backend "nccl": the sharded launch sequence with the all-reduce issued by RCCL on the launch stream."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + 17 * int(minibatch)) % 2000
    p = ctx.Process(target=_worker, args=(0, 1, port, q, "nccl", minibatch))
    p.start()
    o = q.get(timeout=480)
    p.join(60)
    assert p.exitcode == 0
    assert o["ranks"] == 1 and o["backend"] == "nccl" and o["allreduce"] == "rccl-direct"
    if not minibatch:
        assert o["pending"]
    assert abs(o["elbo"] - o["full_elbo"]) <= 2e-6 * abs(o["full_elbo"])
    for n, v in o["params"].items():
        assert abs(v - o["full_params"][n]).max() <= 5e-6, n
