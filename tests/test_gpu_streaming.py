"""
Data sets larger than the device memory their images may take (the reference streams every minibatch from host
memory, tapqir/utils/dataset.py:140-151): ``CosmosEngine(hbm_budget=...)`` / ``TAPQIR_AMD_HBM_BUDGET`` keep the images in
page-locked host memory and bring the AOIs of a step through two device windows, group by group.

  * a streamed step IS the resident step: same draws (keyed by the global unit index), same parameters after mixed
    full-batch / minibatch trajectories (up to the order in which the groups' cross-unit sums are added);
  * the oracle replays streamed minibatch steps from the device's own draws (-ELBO 2e-5, parameters 1e-4);
  * ``Model`` with the budget forced below the data size reproduces the resident fit, and its statistics.
"""

import os

import pytest
import torch

from helpers import CosmosEngine, make_dataset, make_oracle, oracle_to_engine
from test_gpu_production_kernels import replay

pytestmark = pytest.mark.gpu


def _budget(d, window_aois):
    return 2 * window_aois * 4 * d.images[0].numel()


@pytest.mark.parametrize("K", [1, 2])
def test_streamed_steps_equal_resident_steps(K):
    N, F = 12, 40
    d = make_dataset(N=N, F=F, K=K)
    o = make_oracle(d, K, perturb=0.2)
    res = CosmosEngine(d, K=K, device="cuda:0", seed=3)
    stm = CosmosEngine(d, K=K, device="cuda:0", seed=3, hbm_budget=_budget(d, 5))
    assert stm.streamed and stm.window_aois == 5 and stm.images is None and not res.streamed
    if stm.pixstats is not None:
        assert torch.equal(stm.pixstats, res.pixstats)  # data statistics computed window by window
    for e in (res, stm):
        oracle_to_engine(o, e)
    g = torch.Generator().manual_seed(0)
    plan = [(None, None)] * 3 + [(torch.randperm(N, generator=g)[:7], torch.randperm(F, generator=g)[:16]) for _ in range(3)] \
        + [(None, None)] * 2 + [(torch.randperm(N, generator=g)[:3], None)]
    for it, (nd, fd) in enumerate(plan):
        for e in (res, stm):
            e.step(nd, fd)
            e.join()
        torch.cuda.synchronize()
        er, es = float(res.elbo_out[0]), float(stm.elbo_out[0])
        assert abs(er - es) <= 2e-6 * abs(er), (it, er, es)
    vr, vs = res.named("params"), stm.named("params")
    for n in vr:
        assert float((vr[n] - vs[n]).abs().max()) <= 5e-6, n
    assert float((res.exp_avg - stm.exp_avg).abs().max()) <= 1e-5 * float(res.exp_avg.abs().max())
    # post-fit statistics of the streamed engine: window by window
    sr, cr = res.snr_chi2(90.0, 4.0)
    ss, cs = stm.snr_chi2(90.0, 4.0)
    assert torch.allclose(sr, ss, rtol=1e-5, atol=1e-6) and torch.allclose(cr, cs, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("K", [1, 2])
def test_streamed_minibatch_against_oracle(K):
    """The window holds the whole minibatch (one group): the oracle replays each step from the device's draws."""
    N, F = 6, 24
    d = make_dataset(N=N, F=F, K=K)
    o = make_oracle(d, K, perturb=0.3)
    o.make_optim(lr=0.005)
    eng = CosmosEngine(d, K=K, device="cuda:0", seed=11, hbm_budget=_budget(d, 4))
    assert eng.streamed and eng.window_aois == 4
    oracle_to_engine(o, eng)
    replay(eng, o, N, F, steps=3, nb=3, fb=16)


def test_model_fit_with_forced_budget(tmp_path, monkeypatch):
    """`Model` (load -> init -> run -> compute_stats) with TAPQIR_AMD_HBM_BUDGET below the data size: the streamed fit
    reproduces the resident fit's parameters."""
    from tapqir_amd.models import cosmos
    from tapqir_amd.utils.dataset import save
    from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

    class _M:
        K, device = 2, torch.device("cpu")

    data = simulate(_M, 8, 30, 1, 14, seed=4, params=TEST_PARAMS)
    # the same subsample sequence on both sides: host draws (a resident fit would otherwise draw its subsamples on the device,
    # CosmosEngine.step_subsampled, which a streamed engine cannot: it gathers the batch's AOIs on the host)
    monkeypatch.setenv("TAPQIR_AMD_DEVICE_SUBSAMPLE", "0")
    fits = {}
    for mode in ("resident", "streamed"):
        path = tmp_path / mode
        path.mkdir()
        save(data, path)
        if mode == "streamed":
            monkeypatch.setenv("TAPQIR_AMD_HBM_BUDGET", str(_budget(data, 3)))
        else:
            monkeypatch.delenv("TAPQIR_AMD_HBM_BUDGET", raising=False)
        m = cosmos(K=2, device="cuda", dtype="float")
        m.load(path)
        m.init(lr=0.005, nbatch_size=4, fbatch_size=16)
        assert m.engine.streamed == (mode == "streamed")
        m.run(30, progress_bar=lambda x, **k: x)
        m.engine.join()
        fits[mode] = {n: v.detach().cpu().clone() for n, v in m.engine.named("params").items()}
    for n, v in fits["resident"].items():
        assert float((v - fits["streamed"][n]).abs().max()) <= 2e-5, n
