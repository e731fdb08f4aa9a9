"""cosmos.compute_probs (cosmos.py:609-672): HIP kernel math (host build on CPU, device build on GPU)
against the oracle on identical particle draws, plus the sampled read-out against an oracle Monte Carlo."""

import pytest
import torch

from helpers import CosmosEngine, load_hostcheck, make_dataset, make_oracle, oracle_to_engine
from tapqir_amd.models.posterior import probs_args, run_probs


def _particles(o, S, seed=0):
    nd, fd = torch.arange(o.data.Nt), torch.arange(o.data.F)
    torch.manual_seed(seed)
    lats = []
    with torch.no_grad():
        for _ in range(S):
            lat = o.sample_guide(o.params, nd, fd)
            lats.append({k: v.float().double() for k, v in lat.items()})
            lats[-1]["pi"] = torch.stack([1 - lats[-1]["pi"][..., 1], lats[-1]["pi"][..., 1]], -1)
    return nd, fd, lats


def _given_inputs(o, eng, lats, nd, fd):
    """gbase_p [S][14] doubles and xy_given [S][2K][U] floats for tq_cosmos_probs(draw=0)."""
    S, K, Q = len(lats), o.K, o.Q
    U = o.data.Nt * o.data.F * o.data.C
    with torch.no_grad():
        dists = o._guide_dists(o.constrained(o.params), nd, fd)
        gb = torch.zeros(S, eng.gbase.numel(), dtype=torch.float64)  # sizeof(TqGlobalBase) / 8
        xy = torch.zeros(S, 2 * K, U, dtype=torch.float32)
        for s, lat in enumerate(lats):
            base = o.base_draws(lat, dists)
            gb[s, 0] = 1.0
            gb[s, 1] = base["proximity_t"]
            gb[s, 2:2 + Q] = base["lamda_g"]
            gb[s, 6:6 + 2 * Q] = base["pi_x"].reshape(-1)
            xy[s, :K] = lat["x"].reshape(K, U).float()
            xy[s, K:] = lat["y"].reshape(K, U).float()
    return gb.reshape(-1).to(eng.device), xy.reshape(-1).to(eng.device)


def _run_given(device, lib, K, dkw, S=6):
    d = make_dataset(K=K, **dkw)
    o = make_oracle(d, K, perturb=0.4)
    eng = CosmosEngine(d, K=K, device=device, lib=lib)
    oracle_to_engine(o, eng)
    nd, fd, lats = _particles(o, S)
    gb, xy = _given_inputs(o, eng, lats, nd, fd)
    a, ws = probs_args(eng, S, seed=1, draw=False, gbase_p=gb, xy_given=xy)
    run_probs(eng, a)
    if eng.device.type == "cuda":
        torch.cuda.synchronize()
    z_ref, t_ref = o.compute_probs(nd, fd, lats)
    on = d.is_ontarget
    z = ws["z_probs"].cpu().double()
    t = ws["theta_probs"].cpu().double()
    assert (z[on] - z_ref[on]).abs().max() < 2e-5
    assert (t[:, on] - t_ref[:, on]).abs().max() < 2e-5
    assert float(z[~on].abs().max()) == 0.0 and float(t[:, ~on].abs().max()) == 0.0  # cosmos.py:615-623


@pytest.mark.parametrize("K,dkw", [(2, dict(N=4, F=5)), (1, dict(N=2, F=4)), (3, dict(N=2, F=3, P=9)),
                                    (2, dict(N=4, F=3, C=2))])
def test_probs_math_matches_oracle_on_given_particles(K, dkw):
    _run_given("cpu", load_hostcheck(), K, dkw)


@pytest.mark.gpu
@pytest.mark.parametrize("K,dkw", [(2, dict(N=4, F=5)), (1, dict(N=2, F=4)), (3, dict(N=2, F=3, P=9)),
                                    (2, dict(N=4, F=3, C=2))])
def test_probs_kernel_matches_oracle_on_given_particles(K, dkw):
    _run_given("cuda:0", None, K, dkw)


def _sampled(device, lib):
    K = 2
    d = make_dataset(N=4, F=6, K=K)
    o = make_oracle(d, K, perturb=0.4)
    eng = CosmosEngine(d, K=K, device=device, lib=lib)
    oracle_to_engine(o, eng)
    a, ws = probs_args(eng, 400, seed=3)
    run_probs(eng, a)
    nd, fd, lats = _particles(o, 400, seed=9)
    z_ref, t_ref = o.compute_probs(nd, fd, lats)
    on = d.is_ontarget
    z = ws["z_probs"].cpu().double()
    t = ws["theta_probs"].cpu().double()
    # two independent 400-particle Monte-Carlo estimates of probabilities in [0, 1]
    assert (z[on] - z_ref[on]).abs().max() < 0.12
    assert (t[:, on] - t_ref[:, on]).abs().max() < 0.12
    assert (z[on].sum(-1) - 1).abs().max() < 1e-5


def test_sampled_probs_agree_with_oracle_monte_carlo_host():
    _sampled("cpu", load_hostcheck())


@pytest.mark.gpu
def test_sampled_probs_agree_with_oracle_monte_carlo():
    _sampled("cuda:0", None)
