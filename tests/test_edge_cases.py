"""
Edge cases of the data site (ksmogn.py:222-238): a pixel at or below every camera offset has probability zero
(log-likelihood -inf for every combination of that unit, and no gradient), units next to it are unaffected; the C ABI
rejects malformed arguments with an error code and a message instead of launching.
"""

import ctypes as C

import pytest
import torch

from helpers import GIVEN_STAGES, CosmosEngine, fp32_latents, load_hostcheck, make_dataset, make_oracle, oracle_to_engine, put_latents
from tapqir_amd import _lib


def _run(device, lib, offsets, il_min_units=None, crosstalk=False):
    K = 2
    d = make_dataset(N=2, F=3, C=2 if crosstalk else 1, K=K, offsets=offsets)
    lowest = float(d.offset.samples.min())
    d.images[1, 2, 0, 3, 4] = lowest  # one pixel at the smallest offset: no offset sample lies below it
    o = make_oracle(d, K, crosstalk=crosstalk)
    eng = CosmosEngine(d, K=K, device=device, lib=lib, crosstalk=crosstalk)
    if il_min_units is not None:
        eng.il_min_units = il_min_units
    oracle_to_engine(o, eng)
    nd, fd = torch.arange(2), torch.arange(3)
    lat32, base = fp32_latents(o, nd, fd)
    with torch.no_grad():
        o.elbo(o.params, nd, fd, o.latents_from_base(o.params, nd, fd, base))
    a = eng.make_args(None, None, draw_globals=False)
    put_latents(eng, lat32, base)
    for stage in GIVEN_STAGES[:3]:  # tables, site terms, likelihood + per-unit terms
        eng.call(stage, a)
    if eng.device.type == "cuda":
        torch.cuda.synchronize()
    return d, o, eng


def _check(d, o, eng, crosstalk=False):
    C_ = d.images.shape[2]
    B = 2 * 3 * C_
    M = 4
    ll = eng.pix[: M * B].view(M, B).cpu().double()
    ll_o = o.last_terms["ll"].detach()
    if crosstalk:  # per-dye marginals of the joint likelihoods: -inf for both dyes of the AOI-frame with the dead pixel
        bad = torch.zeros(2, 3, C_, dtype=torch.bool)
        bad[1, 2, :] = True
        assert torch.isinf(ll_o[:, 1, 2]).all() and torch.isfinite(ll_o[:, 0, 0]).all()
    else:
        ll_o = ll_o.reshape(M, B)
        bad = torch.isinf(ll_o[0]).reshape(2, 3, C_)
        assert bad.sum() == 1 and bool(bad[1, 2, 0])
        ok = ~bad.reshape(-1)
        assert (ll[:, ok] - ll_o[:, ok]).abs().max() <= 1e-5 * ll_o[:, ok].abs().max()  # north_star: 1e-4 relative
    badf = bad.reshape(-1)
    assert torch.isinf(ll[:, badf]).all() and (ll[:, badf] < 0).all()
    assert torch.isfinite(ll[:, ~badf]).all()
    # pathwise gradient rows of the dead unit are zero (not NaN); the others are finite
    K = 2
    g = eng.pix[M * B:(M + 2 + 4 * K) * B].view(2 + 4 * K, B).cpu()
    assert torch.isfinite(g[:, ~badf]).all()
    assert not torch.isnan(g).any()
    assert (g[:, badf] == 0).all()


@pytest.mark.parametrize("offsets", ["sim", "hist"])
def test_dead_pixel_host_math(offsets):
    _check(*_run("cpu", load_hostcheck(), offsets))


def test_dead_pixel_crosstalk_host_math():
    _check(*_run("cpu", load_hostcheck(), "sim", crosstalk=True), crosstalk=True)


@pytest.mark.gpu
@pytest.mark.parametrize("offsets,il", [("sim", None), ("sim", 1), ("hist", None), ("hist", 1)])
def test_dead_pixel_hip(offsets, il):
    _check(*_run("cuda:0", None, offsets, il_min_units=il))


@pytest.mark.gpu
@pytest.mark.parametrize("il", [None, 1])
def test_dead_pixel_crosstalk_hip(il):
    _check(*_run("cuda:0", None, "sim", il_min_units=il, crosstalk=True), crosstalk=True)


def _clamped_height(device, lib):
    """A height draw on the clamp at the smallest normal float (Gamma with concentration < 1: the height of an absent spot in
    a converged fit; torch's rsample clamps there too) in a minibatch whose plate scale is 40: d log q / d v = (alpha - 1) / v
    is -3e37 and, scaled, left the range of float32 -- inf * (dv / dalpha ~ 1e-36) - inf made the unit's height parameters
    NaN once in ~7000 steps of a fit (tq_gamma_site_chain).  Every gradient must be finite and agree with the oracle."""
    from helpers import oracle_grads, rel_err

    K, N, F = 2, 8, 20
    d = make_dataset(N=N, F=F, K=K)
    o = make_oracle(d, K)
    nd, fd = torch.tensor([1, 5]), torch.tensor([0, 3, 7, 12])
    tiny = 1.1754943508222875e-38
    with torch.no_grad():
        o.params["h_loc"][1, 5, 7, 0] = torch.tensor(468.0).log()
        o.params["h_beta"][1, 5, 7, 0] = torch.tensor(0.00138).log()
        o.params["m_probs"][1, 5, 7, 0] = 4.0  # the guide believes in the spot: its weight in the per-unit terms is ~1
    eng = CosmosEngine(d, K=K, device=device, lib=lib)
    oracle_to_engine(o, eng)
    lat32, base = fp32_latents(o, nd, fd)
    lat32["height"][1, 1, 2, 0] = tiny  # position (AOI 5, frame 7) of the batch
    with torch.no_grad():
        base = o.base_draws(lat32, o._guide_dists(o.constrained(o.params), nd, fd))
    elbo_o, g_o = oracle_grads(o, nd, fd, base)
    a = eng.make_args(nd, fd, draw_globals=False)
    put_latents(eng, lat32, base)
    for stage in GIVEN_STAGES:
        eng.call(stage, a)
    if eng.device.type == "cuda":
        torch.cuda.synchronize()
    assert abs(float(eng.elbo_out[0]) - elbo_o) <= 2e-5 * abs(elbo_o)
    gv = eng.named("grad")
    for n, ref in g_o.items():
        got = gv[n].cpu().double().reshape(ref.shape)
        assert bool(torch.isfinite(got).all()), n
        if n in ("h_loc", "h_beta"):
            # the clamped unit: dv/dalpha is taken at a DENORMAL standard-gamma value (v beta = 1.6e-41) in fp32
            assert abs(float(got[1, 5, 7, 0]) - float(ref[1, 5, 7, 0])) <= 0.05 * abs(float(ref[1, 5, 7, 0])), (n, got[1, 5, 7, 0], ref[1, 5, 7, 0])
            got, ref = got.clone(), ref.clone()
            got[1, 5, 7, 0] = ref[1, 5, 7, 0] = 0.0
        assert rel_err(got, ref) < 1e-4, (n, rel_err(got, ref))


def _wide_histogram_small_gain(device, lib, il_min_units=None):
    """An offset histogram 260 counts wide with the gain at 2.5: (delta_max - delta_min) / gain = 104 > 88, where the weights
    exp(beta (delta_o - delta_min)) of the offset sum alone leave the range of float32.  The sum is taken relative to the
    reference point INCLUDING its exp(-beta v) factor (tq_mo_reference); before that a fit with such a histogram, its gain
    drifting down to 2.9, got d/d gain = -inf for a few units at step 3301 and NaN everywhere two steps later.  Every
    likelihood and gradient row must be finite and agree with the oracle."""
    from helpers import oracle_grads, rel_err

    K, N, F = 2, 2, 6
    d = make_dataset(N=N, F=F, K=K, offsets="wide")
    d.images[:, :, 0] += 150.0  # bright tiles: the reference point lies far below the largest valid v
    o = make_oracle(d, K)
    with torch.no_grad():
        o.params["gain_loc"].fill_(float(torch.tensor(2.5).log()))
    eng = CosmosEngine(d, K=K, device=device, lib=lib)
    if il_min_units is not None:
        eng.il_min_units = il_min_units
    oracle_to_engine(o, eng)
    nd, fd = torch.arange(N), torch.arange(F)
    lat32, base = fp32_latents(o, nd, fd)
    assert float(lat32["gain"]) < 3.0
    elbo_o, g_o = oracle_grads(o, nd, fd, base)
    a = eng.make_args(None, None, draw_globals=False)
    put_latents(eng, lat32, base)
    for stage in GIVEN_STAGES:
        eng.call(stage, a)
    if eng.device.type == "cuda":
        torch.cuda.synchronize()
    B, M = N * F, 4
    rows = eng.pix[: (M + 2 + 4 * K) * B].view(M + 2 + 4 * K, B).cpu()
    assert bool(torch.isfinite(rows).all())
    ll_o = o.last_terms["ll"].detach().reshape(M, B)
    assert (rows[:M].double() - ll_o).abs().max() <= 1e-5 * ll_o.abs().max()
    assert abs(float(eng.elbo_out[0]) - elbo_o) <= 2e-5 * abs(elbo_o)
    gv = eng.named("grad")
    for n, ref in g_o.items():
        got = gv[n].cpu().double().reshape(ref.shape)
        assert bool(torch.isfinite(got).all()), n
        assert rel_err(got, ref) < 2e-4, (n, rel_err(got, ref))


def test_wide_histogram_small_gain_host_math():
    _wide_histogram_small_gain("cpu", load_hostcheck())


@pytest.mark.gpu
@pytest.mark.parametrize("il", [None, 1])
def test_wide_histogram_small_gain_hip(il):
    _wide_histogram_small_gain("cuda:0", None, il_min_units=il)


def test_height_draw_on_the_clamp_host_math():
    _clamped_height("cpu", load_hostcheck())


@pytest.mark.gpu
def test_height_draw_on_the_clamp_hip():
    _clamped_height("cuda:0", None)


def test_abi_rejects_malformed_arguments():
    """No launch, TQ_ERR_ARG (= 1) and a message; checked on the CPU (the checks run before any HIP call)."""
    lib = _lib.load()
    assert lib.tq_ksmogn_log_prob(None, None) == 1
    assert b"NULL" in lib.tq_last_error()
    k = _lib.KsmognArgs()
    assert lib.tq_ksmogn_log_prob(C.byref(k), None) == 1
    x = _lib.XtalkArgs()
    assert lib.tq_ksmogn_crosstalk_log_prob(C.byref(x), None) == 1
    a = _lib.CosmosArgs()
    for name in ("tq_cosmos_step", "tq_cosmos_sample_globals", "tq_cosmos_elbo_grads", "tq_cosmos_adam"):
        assert getattr(lib, name)(C.byref(a), None) == 1, name
    assert b"NULL" in lib.tq_last_error()
    buf = torch.zeros(64)
    a.params = a.globals = a.gbase = buf.data_ptr()
    a.K, a.P, a.C, a.Nt, a.F, a.nb, a.fb, a.O = 9, 14, 1, 2, 2, 2, 2, 1  # K above TQ_MAX_K
    assert lib.tq_cosmos_step(C.byref(a), None) == 1 and b"unsupported" in lib.tq_last_error()
    a.K, a.nb = 2, 0  # empty batch
    assert lib.tq_cosmos_step(C.byref(a), None) == 1
    a.nb, a.crosstalk = 2, 1  # crosstalk needs two channels
    assert lib.tq_cosmos_step(C.byref(a), None) == 1 and b"crosstalk" in lib.tq_last_error()
    p = _lib.ProbsArgs()
    assert lib.tq_cosmos_probs(C.byref(p), None) == 1
    assert lib.tq_images_interleave(None, None, 0, 14, None) == 1
