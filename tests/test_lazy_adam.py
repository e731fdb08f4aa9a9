"""Lazy Adam of minibatch steps (tq_cosmos_adam_catchup) against the dense Adam kernel: torch.optim.Adam updates every
element at every step (model.py:169-171); the lazy path replays the zero-gradient steps of a unit when the unit next
enters a minibatch, or when the buffers are read.  Same trajectories, up to the rounding of 1 - beta^s."""

import pytest
import torch

from helpers import make_dataset, make_oracle, oracle_to_engine
from tapqir_amd.models.engine import CosmosEngine


def engines(C=1, K=2, crosstalk=False, F=16):
    d = make_dataset(N=6, F=F, C=C, K=K, seed=5)
    o = make_oracle(d, K, perturb=0.2, seed=2, crosstalk=crosstalk)
    out = []
    for lazy in (False, True):
        eng = CosmosEngine(d, K=K, device=torch.device("cuda"), seed=11, crosstalk=crosstalk)
        eng.lazy_adam = lazy
        oracle_to_engine(o, eng)
        out.append(eng)
    return out


def close(a, b, tol):
    scale = b.abs().max().clamp(min=1e-30)
    return float((a - b).abs().max() / scale) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("crosstalk,F,fb", [(False, 16, 5), (True, 16, 5), (False, 40, 17)],
                         ids=["cosmos", "crosstalk", "cosmos_one_launch_steps"])
def test_lazy_minibatch_adam_follows_the_dense_trajectory(crosstalk, F, fb):
    """fb >= 16: the lazy engine's minibatch steps run as tq_cosmos_minibatch_step (one launch: catch-up, draws,
    likelihood, per-unit terms, pending tail), the dense engine's as the staged launches + dense Adam kernel."""
    dense, lazy = engines(C=2 if crosstalk else 1, crosstalk=crosstalk, F=F)
    assert lazy.fused_minibatch
    gen = torch.Generator().manual_seed(0)
    touched = torch.zeros(6, F, dtype=torch.bool)
    losses = []
    for it in range(40):
        if it in (17, 31):  # full-batch steps in between: every unit catches up first
            ndx = fdx = None
        else:
            ndx = torch.randperm(6, generator=gen)[:2]
            fdx = torch.randperm(F, generator=gen)[:fb]
            touched[ndx[:, None], fdx[None, :]] = True
        for eng in (dense, lazy):
            eng.step(ndx, fdx)
        if it in (9, 25, 39):  # reading the buffers completes the owed steps
            for name, tol in (("params", 2e-5), ("exp_avg", 2e-4), ("exp_avg_sq", 2e-4)):
                assert close(getattr(lazy, name), getattr(dense, name), tol), (it, name)
            assert not lazy._stale
            losses.append((float(dense.elbo_out), float(lazy.elbo_out)))
    assert not bool(touched.all())  # some units were never in a minibatch: their state is pure replay
    for a, b in losses:
        assert abs(a - b) <= 1e-5 * abs(a)
    assert dense.adam_step == lazy.adam_step == 40


@pytest.mark.gpu
def test_lazy_adam_replay_is_the_dense_update_with_zero_gradient():
    """Straight comparison of the two kernels: after one minibatch step, units outside it hold exactly what the dense
    kernel wrote (the replayed bias corrections agree with the host's to the last bit here)."""
    dense, lazy = engines()
    ndx, fdx = torch.tensor([1, 4]), torch.tensor([0, 3, 7])
    for _ in range(3):
        for eng in (dense, lazy):
            eng.step(ndx, fdx)
    assert lazy._stale
    rows = lazy.layout.views(lazy.params)
    rows_d = dense.layout.views(dense.params)
    for name in ("h_loc", "w_mean", "size", "b_loc"):
        a, b = rows[name], rows_d[name]
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), name
    assert close(lazy.exp_avg, dense.exp_avg, 1e-5) and close(lazy.exp_avg_sq, dense.exp_avg_sq, 1e-5)


@pytest.mark.gpu
def test_model_minibatch_fit_and_checkpoint_round_trip(tmp_path):
    """Model.run with minibatches: checkpoint written from lazily updated buffers, reloaded, continued."""
    from tapqir_amd.models import models
    from tapqir_amd.utils.dataset import save
    from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

    save(simulate(2, 6, 16, 1, 14, params=TEST_PARAMS), tmp_path)
    m = models["cosmos"](S=1, K=2, device="cuda", dtype="double")
    m.load(tmp_path)
    m.init(lr=0.005, nbatch_size=2, fbatch_size=5)
    m.run(30, progress_bar=None)
    assert m.engine.lazy_adam
    p1 = m.engine.params.clone()
    m.save_checkpoint()
    m2 = models["cosmos"](S=1, K=2, device="cuda", dtype="double")
    m2.load(tmp_path)
    m2.init(lr=0.005, nbatch_size=2, fbatch_size=5)  # resumes from the checkpoint
    assert m2.engine.adam_step == m.engine.adam_step and not m2.engine._stale
    assert torch.equal(m2.engine.params, p1)
    m2.run(5, progress_bar=None)
    assert torch.isfinite(m2.engine.params).all()


@pytest.mark.gpu
def test_staged_calls_after_lazy_steps_see_current_parameters():
    """make_args (used by TraceELBO.loss_and_grads and by staged test calls) completes the owed updates first."""
    dense, lazy = engines()
    gen = torch.Generator().manual_seed(1)
    for _ in range(6):
        ndx, fdx = torch.randperm(6, generator=gen)[:2], torch.randperm(16, generator=gen)[:5]
        for eng in (dense, lazy):
            eng.step(ndx, fdx)
    assert lazy._stale
    out = []
    for eng in (dense, lazy):
        a = eng.make_args(torch.tensor([0, 3, 5]), torch.tensor([1, 2, 9, 15]))
        for name in ("cosmos_sample_globals", "cosmos_sample_locals", "cosmos_elbo_grads", "cosmos_globals_grad"):
            eng.call(name, a)
        out.append(float(eng.elbo_out))
    assert not lazy._stale
    assert abs(out[0] - out[1]) <= 1e-5 * abs(out[0])


def test_lazy_adam_on_the_host_build():
    """The same comparison without a GPU: the kernels' inline code compiled with g++ (tests/hostcheck), lazy against
    dense Adam over mixed minibatch / full-batch steps, and the replay of never-touched units against the closed loop."""
    from helpers import HostCheckEngine

    d = make_dataset(N=4, F=6, K=2, seed=5)
    o = make_oracle(d, 2, perturb=0.2, seed=2)
    engs = []
    for lazy in (False, True):
        eng = HostCheckEngine(d, K=2, device=torch.device("cpu"), seed=11)
        eng.lazy_adam = lazy
        oracle_to_engine(o, eng)
        engs.append(eng)
    dense, lazy = engs
    gen = torch.Generator().manual_seed(3)
    for it in range(14):
        if it == 8:
            ndx = fdx = None
        else:
            ndx, fdx = torch.randperm(4, generator=gen)[:2], torch.randperm(6, generator=gen)[:3]
        for eng in engs:
            eng.step(ndx, fdx)
        if it in (5, 13):
            assert lazy._stale
            for name, tol in (("params", 2e-5), ("exp_avg", 2e-4), ("exp_avg_sq", 2e-4)):
                assert close(getattr(lazy, name), getattr(dense, name), tol), (it, name)
            assert not lazy._stale
    assert dense.adam_step == lazy.adam_step == 14
