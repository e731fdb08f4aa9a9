"""
AOI-sharded data parallelism, world_size 2 over gloo on the CPU.  The ranks drive the g++ host
build of the kernels' math (tests/hostcheck) -- the arithmetic is not what is under test here; the
sharding, the global-index RNG keys, the plate scales and the position of the single all-reduce are.
"""

import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, async_op=False, minibatch=False):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import CosmosEngine, load_hostcheck, make_dataset, make_oracle, oracle_to_engine
    from tapqir_amd.parallel import make_allreduce, shard_dataset

    hc = load_hostcheck()
    K, N, F = 2, 6, 5
    d = make_dataset(N=N, F=F, K=K)
    o = make_oracle(d, K, perturb=0.2)
    sub, n_off, Nt_global = shard_dataset(d, rank, world)
    eng = CosmosEngine(sub, K=K, device="cpu", lib=hc, seed=5, n_offset=n_off, Nt_global=Nt_global)
    # copy this rank's slice of the parameters
    full = CosmosEngine(d, K=K, device="cpu", lib=hc, seed=5)
    oracle_to_engine(o, full)
    fv, sv = full.named("params"), eng.named("params")
    lo, hi = n_off, n_off + sub.images.shape[0]
    for n in sv:
        if sv[n].dim() == 4:
            sv[n].copy_(fv[n][:, lo:hi])
        elif sv[n].dim() == 3:
            sv[n].copy_(fv[n][lo:hi])
        else:
            sv[n].copy_(fv[n])
    allreduce = make_allreduce(async_op=async_op)
    nsteps = 3 if async_op else 2
    # minibatch mode: every rank subsamples its own AOIs (local indices) and the common frames; with the lazy Adam
    # clock the units outside the minibatches are caught up when the parameters are read
    subs = [([0, 2], [1, 3, 4]), ([1, 2], [0, 2]), ([0, 1], [0, 1, 4]), ([0, 2], [2, 3])] if minibatch else [(None, None)] * nsteps
    if minibatch:
        eng.lazy_adam = full.lazy_adam = True
    t = lambda v: None if v is None else torch.tensor(v)
    for ndx, fdx in subs:
        eng.step(t(ndx), t(fdx), allreduce=allreduce)
    if async_op:
        assert eng._pending is not None  # the last step's global tail is still waiting for its collective
    if minibatch:
        assert eng._stale
    eng.join()
    out = {"rank": rank, "lo": lo, "hi": hi, "elbo": float(eng.elbo_out[0]),
           "params": {n: v.clone().numpy() for n, v in eng.named("params").items()}}
    if rank == 0:
        per = N // world
        for ndx, fdx in subs:
            gn = None if ndx is None else [r * per + j for r in range(world) for j in ndx]  # the union of the ranks' AOIs
            full.step(t(gn), t(fdx))
        out["full_elbo"] = float(full.elbo_out[0])
        out["full_params"] = {n: v.clone().numpy() for n, v in full.named("params").items()}
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("async_op,minibatch", [(False, False), (True, False), (True, True)],
                         ids=["blocking_allreduce", "overlapped_allreduce", "minibatch_lazy_adam"])
def test_two_rank_sharded_steps_equal_single_process(async_op, minibatch):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7 * int(async_op) + 13 * int(minibatch)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, async_op, minibatch)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda o: o["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = outs[0]
    # the ELBO every rank reports is the global one and equals the single-process value
    for o in outs:
        assert abs(o["elbo"] - ref["full_elbo"]) <= 1e-6 * abs(ref["full_elbo"])
    for o in outs:
        lo, hi = o["lo"], o["hi"]
        for n, v in o["params"].items():
            fp = ref["full_params"][n]
            want = fp[:, lo:hi] if v.ndim == 4 else (fp[lo:hi] if v.ndim == 3 else fp)
            assert abs(v - want).max() <= 2e-6, n


def test_shard_bounds_cover_everything():
    from tapqir_amd.parallel import shard_bounds

    for Nt in (1, 7, 400, 3200):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(Nt, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == Nt
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _model_worker(rank, world, port, q, tmp):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np

    from helpers import HostCheckEngine
    from tapqir_amd.models import cosmos
    from tapqir_amd.models.cosmos import data_median
    from tapqir_amd.parallel import attach

    m = cosmos(K=2, device="cpu")
    m.load(tmp)
    full_labels = np.asarray(m.data.labels["z"]).copy()
    full_median = data_median(m.data.images)
    attach(m)
    out = {"rank": rank}
    # labels follow the shard (on-target AOIs are the first half of 6: rank 0 holds 3 labelled AOIs, rank 1 none)
    lo, hi = (0, 3) if rank == 0 else (3, 6)
    out["labels_ok"] = bool(np.array_equal(np.asarray(m.data.labels["z"]), full_labels[lo:min(hi, 3)]))
    m._make_engine(engine_cls=HostCheckEngine)
    m.init(lr=0.005, nbatch_size=3, fbatch_size=5)
    # every rank initialises from the median of the WHOLE dataset (counting median summed over ranks)
    bml = m.engine.layout.constrained(m.engine.params)["background_mean_loc"]
    out["median_ok"] = bool(torch.allclose(bml.double().cpu(), (full_median - m.data.offset.mean).expand_as(bml), rtol=1e-6))
    m.run(2, progress_bar=lambda r: r)
    out["stats_dir"] = str(m.stats_path)
    # a NaN on ONE rank makes EVERY rank take the recovery branch (model.py:220-232), with one common new seed
    if rank == 1:
        m.engine.params[0] = float("nan")
    m.iter_loss = 0.0
    try:
        m.save_checkpoint()
        out["raised"] = False
    except ValueError:
        out["raised"] = True
    import random

    random.seed(100 + rank)  # the ranks' own draws differ: 0 decides
    out["seed"] = m.collective.broadcast_int(random.randint(0, 100))
    m2 = cosmos(K=2, device="cpu")
    m2.load(tmp)
    attach(m2)
    m2._make_engine(engine_cls=HostCheckEngine)
    m2.init(lr=0.005, nbatch_size=3, fbatch_size=5)
    m2.compute_stats()
    out["stats_files"] = sorted(os.listdir(m2.stats_path))
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_model_ranks_stay_in_step(tmp_path):
    """Model-level behaviour of an AOI-sharded fit (tapqir_amd.parallel.attach): global median at initialisation, labels
    sliced with the shard, NaN recovery and reseeding agreed between the ranks, per-rank output files."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from tapqir_amd.utils.dataset import save
    from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

    d = simulate(2, 6, 5, 1, 14, 0, TEST_PARAMS)
    d.images[:3] += 40.0  # the two shards have different local medians
    save(d, tmp_path)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 101) % 2000
    procs = [ctx.Process(target=_model_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda o: o["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(o["labels_ok"] and o["median_ok"] and o["raised"] for o in outs), outs
    assert outs[0]["seed"] == outs[1]["seed"]
    assert outs[0]["stats_dir"] != outs[1]["stats_dir"]
    for o in outs:
        assert "cosmos_params.tpqr" in o["stats_files"] and "cosmos_summary.csv" in o["stats_files"]


def _ckpt_worker(rank, world, port, q, tmp):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import HostCheckEngine
    from test_model_api import _FakeWriter
    from tapqir_amd.models import cosmos
    from tapqir_amd.parallel import attach

    def fresh():
        m = cosmos(K=2, device="cpu")
        m.load(tmp)
        attach(m)
        m._make_engine(engine_cls=HostCheckEngine)
        m.init(lr=0.005, nbatch_size=3, fbatch_size=5)
        return m

    m = fresh()
    m.run(2, progress_bar=lambda r: r)
    n = m.engine.params.numel()
    m._in_run = True
    # the helper of rank 1 is busy at iteration 200, that of rank 0 is not: BOTH ranks leave the file out
    w = m._ckpt_process = _FakeWriter(n, busy=(rank == 1))
    m.iter = 200
    m._write_state_file()
    out = {"rank": rank, "stale_200": m._ckpt_file_stale, "submitted_200": list(w.submitted)}
    w._busy = False
    m.iter = 400
    m._write_state_file()
    out["stale_400"], out["submitted_400"] = m._ckpt_file_stale, list(w.submitted)
    m._in_run = False
    m._ckpt_process = None
    # end of run(): both ranks write the same iteration; a NaN on ONE rank keeps the previous file on BOTH
    m._ckpt_file_stale = True
    m.iter = 450
    if rank == 1:
        m.engine.params[0] = float("nan")
    m._final_state_file()
    if rank == 1:
        m.engine.params[0] = 0.0
    m2 = fresh()  # resume: same iteration and same global parameters on every rank
    out["resume_iter"] = m2.iter
    g = m2.engine.layout.constrained(m2.engine.params, {"gain_loc", "proximity_loc", "lamda_loc"})
    out["globals"] = [float(g["gain_loc"]), float(g["proximity_loc"]), float(g["lamda_loc"][0])]
    m.iter = 460
    m._ckpt_file_stale = True
    m._final_state_file()
    out["resume_iter_2"] = fresh().iter
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_checkpoint_files_hold_the_same_iteration_on_every_rank(tmp_path):
    """whether a checkpoint FILE is deferred, and whether the end-of-run file is written, is decided
    collectively, so that the ranks of a sharded fit always resume from the same iteration."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from tapqir_amd.utils.dataset import save
    from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

    save(simulate(2, 6, 5, 1, 14, 0, TEST_PARAMS), tmp_path)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 211) % 2000
    procs = [ctx.Process(target=_ckpt_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda o: o["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = outs
    assert a["stale_200"] and b["stale_200"] and a["submitted_200"] == b["submitted_200"] == []
    assert not a["stale_400"] and not b["stale_400"] and a["submitted_400"] == b["submitted_400"] == [400]
    assert a["resume_iter"] == b["resume_iter"] == 0  # run(2) wrote iteration 0; the NaN of rank 1 kept it on both ranks
    assert a["globals"] == b["globals"]
    assert a["resume_iter_2"] == b["resume_iter_2"] == 460
