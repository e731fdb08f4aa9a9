"""
CPU checks of the kernels' inline math (tapqir_amd/csrc/*.h compiled with g++, driven by
tests/hostcheck) against the oracle: the hand-derived ELBO gradient must equal fp64 autograd
of the dense restatement.  These tests never touch the product path's launch code; the same
comparisons run through the C ABI on the GPU in test_gpu_parity.py.
"""

import pytest
import torch

from helpers import (GIVEN_STAGES, CosmosEngine, fp32_latents, load_hostcheck, make_dataset, make_oracle, oracle_grads,
                     oracle_to_engine, put_latents, rel_err)

CASES = [
    # id, dataset kwargs, K, ndx, fdx
    ("K2_minibatch", dict(N=4, F=6), 2, [0, 2, 3], [1, 4, 5, 0]),
    ("K1_minibatch", dict(N=4, F=5), 1, [3, 0], [0, 2, 4]),
    ("K3_P9", dict(N=2, F=3, P=9), 3, [1, 0], [2, 0, 1]),
    ("K4_max_spots", dict(N=2, F=2), 4, None, None),  # TQ_MAX_K: 16 spot-presence combinations
    ("K2_fullbatch", dict(N=4, F=3), 2, None, None),
    ("K2_two_channels", dict(N=4, F=3, C=2), 2, [0, 3], [0, 2]),
    ("K2_offset_histogram", dict(N=2, F=3, offsets="hist"), 2, None, None),
    ("K2_offsets_partly_masked", dict(N=2, F=3, offsets="wide"), 2, None, None),
    ("K2_offsets_peaked_weights", dict(N=2, F=3, offsets="peaked"), 2, None, None),
    ("K2_P20", dict(N=2, F=2, P=20), 2, None, None),
    ("K2_masked_aoi", dict(N=4, F=3, mask=torch.tensor([True, False, True, True])), 2, None, None),
]


def run_case(dkw, K, ndx, fdx, perturb=0.3):
    hc = load_hostcheck()
    d = make_dataset(K=K, **dkw)
    o = make_oracle(d, K, perturb=perturb)
    eng = CosmosEngine(d, K=K, device="cpu", lib=hc)
    oracle_to_engine(o, eng)
    nd = torch.arange(d.images.shape[0]) if ndx is None else torch.tensor(ndx)
    fd = torch.arange(d.images.shape[1]) if fdx is None else torch.tensor(fdx)
    lat32, base = fp32_latents(o, nd, fd)
    elbo_o, g_o = oracle_grads(o, nd, fd, base)
    a = eng.make_args(None if ndx is None else nd, None if fdx is None else fd, draw_globals=False)
    put_latents(eng, lat32, base)
    for stage in GIVEN_STAGES:
        eng.call(stage, a)
    return o, eng, elbo_o, g_o


@pytest.mark.parametrize("name,dkw,K,ndx,fdx", CASES, ids=[c[0] for c in CASES])
def test_elbo_and_gradients_match_oracle(name, dkw, K, ndx, fdx):
    o, eng, elbo_o, g_o = run_case(dkw, K, ndx, fdx)
    elbo_k = float(eng.elbo_out[0])
    assert abs(elbo_k - elbo_o) <= 1e-5 * abs(elbo_o), (elbo_k, elbo_o)  # north_star: 1e-4 relative
    gv = eng.named("grad")
    for n, ref in g_o.items():
        got = gv[n].double().reshape(ref.shape)
        assert rel_err(got, ref) < 1e-4, (n, rel_err(got, ref))


def test_per_term_log_likelihood():
    """ll(m) from the pixel math vs the oracle's dense KSMOGN for every combination."""
    o, eng, _, _ = run_case(dict(N=3, F=4), 2, None, None)
    B = 3 * 4
    ll_k = eng.pix[: 4 * B].view(4, B).double()
    ll_o = o.last_terms["ll"].detach().reshape(4, B)
    assert rel_err(ll_k, ll_o) < 2e-6


def test_unperturbed_initial_parameters():
    """Gradients at the reference's initial parameter values (cosmos.py:471-598)."""
    o, eng, elbo_o, g_o = run_case(dict(N=4, F=4), 2, None, None, perturb=0.0)
    assert abs(float(eng.elbo_out[0]) - elbo_o) <= 1e-5 * abs(elbo_o)
    gv = eng.named("grad")
    for n, ref in g_o.items():
        assert rel_err(gv[n].double().reshape(ref.shape), ref) < 1e-4, n


def test_extreme_gamma_draw_far_below_the_mean():
    """A height draw many orders of magnitude below h_loc (Gamma concentration < 1, as h_beta adapts during a fit)
    must keep log q and the gradients finite and equal to the oracle's (regression: ln(v/loc) via log1p rounded to -inf)."""
    import math

    hc = load_hostcheck()
    K = 2
    d = make_dataset(N=2, F=3, K=K)
    o = make_oracle(d, K, perturb=0.2)
    o.params["h_loc"].data[0, 0, 0, 0] = math.log(394.0)
    o.params["h_beta"].data[0, 0, 0, 0] = math.log(0.00197)
    for u in o.params.values():
        u.data = u.data.float().double()
    eng = CosmosEngine(d, K=K, device="cpu", lib=hc)
    oracle_to_engine(o, eng)
    nd, fd = torch.arange(2), torch.arange(3)
    lat32, _ = fp32_latents(o, nd, fd)
    lat32["height"][0, 0, 0, 0] = torch.tensor(2.5122093916252197e-07).float().double()
    with torch.no_grad():
        base = o.base_draws(lat32, o._guide_dists(o.constrained(o.params), nd, fd))
    elbo_o, g_o = oracle_grads(o, nd, fd, base)
    a = eng.make_args(None, None, draw_globals=False)
    put_latents(eng, lat32, base)
    for stage in GIVEN_STAGES:
        eng.call(stage, a)
    elbo_k = float(eng.elbo_out[0])
    assert math.isfinite(elbo_k) and abs(elbo_k - elbo_o) <= 1e-5 * abs(elbo_o), (elbo_k, elbo_o)
    gv = eng.named("grad")
    for n, ref in g_o.items():
        got = gv[n].double().reshape(ref.shape)
        assert torch.isfinite(got).all(), n
        assert rel_err(got, ref) < 1e-4, (n, rel_err(got, ref))
