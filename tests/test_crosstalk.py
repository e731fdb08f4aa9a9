"""
Crosstalk model (tapqir/models/crosstalk.py, SURVEY §8 a14 / BASELINE config c4): oracle self-consistency,
the kernels' inline math on the CPU (tests/hostcheck) and, on a GPU, the HIP path through the C ABI.
"""

import pytest
import torch

from helpers import (GIVEN_STAGES, CosmosEngine, fp32_latents, load_hostcheck, make_dataset, make_oracle, oracle_grads,
                     oracle_to_engine, put_latents, read_engine_latents, rel_err)
from oracle.crosstalk import elbo_bruteforce_crosstalk

XT_CASES = [
    # id, dataset kwargs, K, ndx, fdx
    ("K2_fullbatch", dict(N=3, F=4, C=2), 2, None, None),
    ("K2_minibatch", dict(N=4, F=5, C=2), 2, [3, 0], [0, 2, 4]),
    ("K1", dict(N=3, F=3, C=2), 1, None, None),
    ("K2_offset_histogram", dict(N=2, F=2, C=2, offsets="hist"), 2, None, None),
    ("K2_masked_aoi", dict(N=3, F=2, C=2, mask=torch.tensor([True, False, True])), 2, None, None),
]


def run_xt_case(dkw, K, ndx, fdx, gpu, perturb=0.3, il_min_units=None):
    d = make_dataset(K=K, **dkw)
    o = make_oracle(d, K, perturb=perturb, crosstalk=True)
    if gpu:
        eng = CosmosEngine(d, K=K, device="cuda:0", crosstalk=True)
    else:
        eng = CosmosEngine(d, K=K, device="cpu", lib=load_hostcheck(), crosstalk=True)
    if il_min_units is not None:
        eng.il_min_units = il_min_units
    oracle_to_engine(o, eng)
    nd = torch.arange(d.images.shape[0]) if ndx is None else torch.tensor(ndx)
    fd = torch.arange(d.images.shape[1]) if fdx is None else torch.tensor(fdx)
    lat32, base = fp32_latents(o, nd, fd)
    elbo_o, g_o = oracle_grads(o, nd, fd, base)
    a = eng.make_args(None if ndx is None else nd, None if fdx is None else fd, draw_globals=False)
    put_latents(eng, lat32, base)
    for stage in GIVEN_STAGES:
        eng.call(stage, a)
    if gpu:
        torch.cuda.synchronize()
    return o, eng, elbo_o, g_o


def check(o, eng, elbo_o, g_o):
    elbo_k = float(eng.elbo_out[0])
    assert abs(elbo_k - elbo_o) <= 1e-5 * abs(elbo_o), (elbo_k, elbo_o)  # north_star: 1e-4 relative
    gv = eng.named("grad")
    assert set(g_o) == set(gv)
    for n, ref in g_o.items():
        got = gv[n].cpu().double().reshape(ref.shape)
        assert rel_err(got, ref) < 1e-4, (n, rel_err(got, ref))


def test_oracle_dense_equals_bruteforce():
    d = make_dataset(N=2, F=2, C=2, P=6, K=2)
    o = make_oracle(d, 2, crosstalk=True)
    nd, fd = torch.arange(2), torch.arange(2)
    torch.manual_seed(0)
    with torch.no_grad():
        lat = o.sample_guide(o.params, nd, fd)
        dense = float(o.elbo(o.params, nd, fd, lat))
    brute = elbo_bruteforce_crosstalk(o, o.params, nd, fd, lat)
    assert abs(dense - brute) <= 1e-12 * abs(brute)


def test_oracle_reduces_to_cosmos_without_crosstalk():
    """alpha = identity: channel c sees dye c only, so the ELBO differs from cosmos by the alpha site alone."""
    d = make_dataset(N=2, F=3, C=2, K=2)
    oc = make_oracle(d, 2)
    ox = make_oracle(d, 2, crosstalk=True)
    for n in oc.params:
        ox.params[n].data.copy_(oc.params[n].data)
    nd, fd = torch.arange(2), torch.arange(3)
    torch.manual_seed(1)
    with torch.no_grad():
        lat = oc.sample_guide(oc.params, nd, fd)
        e_c = float(oc.elbo(oc.params, nd, fd, lat))
        lat_x = dict(lat)
        tiny = 1e-300
        lat_x["alpha"] = torch.tensor([[1 - tiny, tiny], [tiny, 1 - tiny]], dtype=torch.float64)
        e_x = float(ox.elbo(ox.params, nd, fd, lat_x))
        g = ox._guide_dists(ox.constrained(ox.params), nd, fd)["alpha"]
        import torch.distributions as D
        site = float((D.Dirichlet(ox.alpha_prior_conc()).log_prob(lat_x["alpha"]) - g.log_prob(lat_x["alpha"])).sum())
    assert abs((e_x - site) - e_c) <= 1e-9 * abs(e_c)


@pytest.mark.parametrize("name,dkw,K,ndx,fdx", XT_CASES, ids=[c[0] for c in XT_CASES])
def test_hostcheck_elbo_and_gradients(name, dkw, K, ndx, fdx):
    check(*run_xt_case(dkw, K, ndx, fdx, gpu=False))


@pytest.mark.gpu
@pytest.mark.parametrize("name,dkw,K,ndx,fdx", XT_CASES, ids=[c[0] for c in XT_CASES])
def test_gpu_elbo_and_gradients(name, dkw, K, ndx, fdx):
    check(*run_xt_case(dkw, K, ndx, fdx, gpu=True))


@pytest.mark.gpu
@pytest.mark.parametrize("minibatch", [False, True])
def test_gpu_full_step_trajectory(minibatch):
    """Three complete HIP steps of the crosstalk model (device sampling + Adam) replayed by the oracle with the
    device's own draws, its own autograd and torch.optim.Adam."""
    K, N, F = 2, 3, 5
    d = make_dataset(N=N, F=F, C=2, K=K)
    o = make_oracle(d, K, perturb=0.0, crosstalk=True)
    o.make_optim(lr=0.005)
    eng = CosmosEngine(d, K=K, device="cuda:0", seed=11, crosstalk=True)
    oracle_to_engine(o, eng)
    g = torch.Generator().manual_seed(5)
    for it in range(3):
        nd = torch.randperm(N, generator=g)[:2] if minibatch else torch.arange(N)
        fd = torch.randperm(F, generator=g)[:3] if minibatch else torch.arange(F)
        eng.step(nd if minibatch else None, fd if minibatch else None)
        eng.join()  # full-batch steps leave their global tail pending for the next launch
        torch.cuda.synchronize()
        lat32 = read_engine_latents(eng, len(nd), len(fd))
        with torch.no_grad():
            base = o.base_draws(lat32, o._guide_dists(o.constrained(o.params), nd, fd))
        loss_o = o.step(nd, fd, base=base)
        assert abs(-float(eng.elbo_out[0]) - loss_o) <= 2e-5 * abs(loss_o)
        views = eng.named("params")
        for n, u in o.params.items():
            got = views[n].cpu().double().reshape(u.shape)
            assert (got - u.detach()).abs().max() < 1e-4, (it, n, float((got - u.detach()).abs().max()))
        oracle_to_engine(o, eng)


@pytest.mark.gpu
def test_gpu_model_api_fit_and_posteriors(tmp_path):
    """models["crosstalk"]: load / init / run / compute_stats with the reference's shapes (crosstalk.py:466-574)."""
    from tapqir_amd.models import models
    from tapqir_amd.utils.dataset import save

    d = make_dataset(N=4, F=6, C=2, K=2)
    save(d, tmp_path)
    m = models["crosstalk"](S=1, K=2, device="cuda", dtype="float")
    m.load(tmp_path)
    m.init(lr=0.005, nbatch_size=4, fbatch_size=6)
    m.run(3, progress_bar=None)
    assert m.z_probs.shape == (4, 6, 2) and m.theta_probs.shape == (2, 4, 6, 2) and m.z_map.dtype == torch.bool
    cp = m.engine.layout.constrained(m.engine.params)
    assert cp["alpha_mean"].shape == (2, 2) and torch.allclose(cp["alpha_mean"].sum(-1).cpu(), torch.ones(2))
    assert torch.isfinite(m.engine.params).all()


def test_registry_names():
    from tapqir_amd.models import models

    assert set(models) == {"cosmos", "crosstalk", "cosmos+hmm"}  # tapqir/models/__init__.py:17-21
    with pytest.raises(NotImplementedError):
        models["cosmos+hmm"]()


XT_IL_CASES = [  # contiguous full batches through the packed lane-per-AOI-frame kernel
    ("K2", dict(N=3, F=4, C=2), 2),
    ("K2_partial_wave_and_two_waves", dict(N=5, F=15, C=2), 2),      # 75 AOI-frames
    ("K2_P20", dict(N=2, F=2, C=2, P=20), 2),
    ("K2_masked_aoi", dict(N=3, F=2, C=2, mask=torch.tensor([True, False, True])), 2),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,dkw,K", XT_IL_CASES, ids=[c[0] for c in XT_IL_CASES])
def test_gpu_packed_kernel_matches_oracle(name, dkw, K):
    o, eng, elbo_o, g_o = run_xt_case(dkw, K, None, None, gpu=True, il_min_units=1)
    assert eng.images_il is not None and eng.pixstats is not None
    check(o, eng, elbo_o, g_o)


@pytest.mark.gpu
def test_gpu_packed_and_16_lane_kernels_agree():
    outs = []
    for il in (1, 1 << 30):
        _, eng, _, _ = run_xt_case(dict(N=5, F=15, C=2), 2, None, None, gpu=True, il_min_units=il)
        outs.append(eng.pix.cpu().double().clone())
    scale = outs[1].abs().max()
    assert (outs[0] - outs[1]).abs().max() <= 2e-5 * scale
