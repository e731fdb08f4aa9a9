"""The oracle against what pins it: the reference's own util.py vectors,
installed torch.distributions, brute-force loops."""

import math
import os

import numpy as np
import pytest
import torch
import torch.distributions as D

from helpers import make_dataset, make_oracle
from oracle import dist_util as du
from oracle.cosmos import AffineBeta, elbo_bruteforce
from oracle.ksmogn import ksmogn_log_prob, ksmogn_log_prob_bruteforce

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "util_golden.npz"))
T = lambda k: torch.tensor(G[k])


@pytest.mark.parametrize("K", [1, 2, 3, 4])
def test_probs_tables_match_reference_vectors(K):
    lam = T("lamda")
    assert np.abs(du.probs_m(lam, K).numpy() - G[f"probs_m_K{K}"]).max() < 1e-15
    assert np.abs(du.truncated_poisson_probs(lam, K).numpy() - G[f"trpois_K{K}"]).max() < 1e-15
    assert np.array_equal(du.probs_theta(K).numpy(), G[f"probs_theta_K{K}"])


def test_known_answers_from_survey():
    """lamda=0.15."""
    lam = torch.tensor(0.15, dtype=torch.float64)
    assert abs(float(du.probs_m(lam, 1)[0, 0]) - 0.1392920235749422) < 1e-16
    p2 = du.probs_m(lam, 2)
    assert abs(float(p2[0, 0]) - 0.07473892534306288) < 1e-16 and abs(float(p2[1, 1]) - 0.1392920235749422) < 1e-16
    assert float(p2[1, 0]) == 1.0 and float(p2[2, 1]) == 1.0
    assert abs(float(du.probs_m(lam, 3)[0, 0]) - 0.04999357102084245) < 1e-16


def test_expand_offtarget():
    assert np.array_equal(du.expand_offtarget(T("pi")).numpy(), G["expand_offtarget"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gaussian_spots_match_reference_vectors(tag):
    a = {n: T(f"gs_{tag}_{n}") for n in "h w x y tl m".split()}
    P = int(G[f"gs_{tag}_P"])
    out = du.gaussian_spots(a["h"], a["w"], a["x"], a["y"], a["tl"], P).numpy()
    ref = G[f"gs_{tag}_out"]
    assert np.abs(out - ref).max() <= 1e-14 * np.abs(ref).max()
    outm = du.gaussian_spots(a["h"], a["w"], a["x"], a["y"], a["tl"], P, a["m"]).numpy()
    assert np.abs(outm - G[f"gs_{tag}_out_m"]).max() <= 1e-14 * np.abs(ref).max()


def test_gaussian_spots_canonical_and_orientation():
    h = torch.full((2, 3, 1, 2), 3000.0, dtype=torch.float64)
    out = du.gaussian_spots(h, torch.full_like(h, 1.4), torch.zeros_like(h), torch.zeros_like(h),
                            torch.full((2, 3, 1, 1, 2), 6.5, dtype=torch.float64), 14)
    assert np.abs(out.numpy() - G["gs_canon"]).max() < 1e-12
    assert abs(float(out[0, 0, 0, 0].sum()) - 2999.9980) < 1e-3
    one = torch.ones(1, dtype=torch.float64)
    o = du.gaussian_spots(one * 1000, one * 1.4, one * 2.0, one * 0.0, torch.full((1, 2), 6.5, dtype=torch.float64), 14)
    assert np.abs(o.numpy() - G["gs_orient"]).max() < 1e-12
    j, i = np.unravel_index(np.argmax(o[0].numpy()), (14, 14))
    assert i in (8, 9) and j in (6, 7)  # x moves along columns, y along rows


def test_affine_beta_against_torch_beta():
    mean, size, lo, hi = torch.tensor(0.3, dtype=torch.float64), torch.tensor(50.0, dtype=torch.float64), -7.5, 7.5
    ab = AffineBeta(mean, size, lo, hi)
    y = torch.tensor(1.234, dtype=torch.float64)
    t = (y - lo) / (hi - lo)
    ref = D.Beta(size * (mean - lo) / (hi - lo), size * (hi - mean) / (hi - lo)).log_prob(t) - math.log(hi - lo)
    assert abs(float(ab.log_prob(y) - ref)) < 1e-14
    # rsample clamps into [low + eps*scale, high - eps*scale]
    eps = torch.finfo(torch.float64).eps * (hi - lo)
    assert float(ab.from_base(torch.tensor(0.0, dtype=torch.float64))) == lo + eps
    assert float(ab.from_base(torch.tensor(1.0, dtype=torch.float64))) == hi - eps


def test_ksmogn_dense_against_per_pixel_gamma_loop():
    g = torch.Generator().manual_seed(0)
    P = 14
    val = torch.floor(240 + 50 * torch.rand(P, P, generator=g, dtype=torch.float64))
    h = torch.tensor([3000.0, 2000.0], dtype=torch.float64)
    w = torch.tensor([1.4, 1.2], dtype=torch.float64)
    x = torch.tensor([0.3, -2.0], dtype=torch.float64)
    y = torch.tensor([-0.5, 1.7], dtype=torch.float64)
    tl = torch.tensor([6.5, 6.5], dtype=torch.float64)
    b, gain = torch.tensor(150.0, dtype=torch.float64), torch.tensor(7.0, dtype=torch.float64)
    for offs, ow in ((torch.tensor([90.0, 90.0, 90.0]), torch.ones(3) / 3),
                     (torch.arange(80.0, 100.0), torch.softmax(torch.randn(20, generator=g), 0))):
        offs, ow = offs.double(), ow.double()
        for m in ([0, 0], [1, 0], [0, 1], [1, 1]):
            mt = torch.tensor(m, dtype=torch.float64)
            dense = ksmogn_log_prob(val, h, w, x, y, tl, b, gain, offs, ow.log(), P, mt)
            brute = ksmogn_log_prob_bruteforce(val, h, w, x, y, tl, b, gain, offs, ow, P, mt)
            assert abs(float(dense) - brute) < 1e-9 * abs(brute)


def test_ksmogn_masks_offsets_at_or_above_the_pixel():
    P = 4
    val = torch.full((P, P), 95.0, dtype=torch.float64)
    z = torch.zeros(1, dtype=torch.float64)
    args = (z + 10, z + 1.4, z, z, torch.tensor([1.5, 1.5], dtype=torch.float64), torch.tensor(5.0, dtype=torch.float64),
            torch.tensor(2.0, dtype=torch.float64))
    offs = torch.tensor([90.0, 95.0, 99.0], dtype=torch.float64)
    lw = torch.log(torch.tensor([0.2, 0.5, 0.3], dtype=torch.float64))
    full = ksmogn_log_prob(val, *args, offs, lw, P)
    only_first = ksmogn_log_prob(val, *args, offs[:1], lw[:1], P)
    assert abs(float(full - only_first)) < 1e-12  # D == delta and D < delta contribute nothing


@pytest.mark.parametrize("K", [1, 2, 3])
def test_dense_elbo_equals_bruteforce_enumeration(K):
    d = make_dataset(N=4, F=6, K=K)
    o = make_oracle(d, K, eps=None)
    ndx, fdx = torch.tensor([0, 2, 3]), torch.tensor([1, 4])
    torch.manual_seed(0)
    lat = o.sample_guide(o.params, ndx, fdx)
    dense = float(o.elbo(o.params, ndx, fdx, lat))
    brute = elbo_bruteforce(o, o.params, ndx, fdx, lat)
    assert abs(dense - brute) < 1e-12 * abs(brute)


def test_base_draw_path_reproduces_native_rsample_gradients():
    d = make_dataset(N=4, F=6, K=2)
    o = make_oracle(d, 2, eps=None)
    ndx, fdx = torch.tensor([0, 1, 3]), torch.tensor([0, 2, 5])
    torch.manual_seed(5)
    lat = o.sample_guide(o.params, ndx, fdx)
    e = o.elbo(o.params, ndx, fdx, lat)
    g1 = torch.autograd.grad(e, list(o.params.values()))
    base = o.base_draws(lat, o._guide_dists(o.constrained(o.params), ndx, fdx))
    e2 = o.elbo(o.params, ndx, fdx, o.latents_from_base(o.params, ndx, fdx, base))
    g2 = torch.autograd.grad(e2, list(o.params.values()))
    assert abs(float(e - e2)) < 1e-9 * abs(float(e))
    for a, b in zip(g1, g2):
        assert float((a - b).abs().max()) <= 1e-8 * float(a.abs().max() + 1e-30)


def test_elbo_scales_with_plate_subsampling():
    """Full batch equals the mean over disjoint frame halves of the scaled minibatch ELBOs
    (plate scaling F/fb), local part only."""
    d = make_dataset(N=2, F=4, K=1)
    o = make_oracle(d, 1, eps=None)
    nd, fd = torch.arange(2), torch.arange(4)
    torch.manual_seed(1)
    lat = o.sample_guide(o.params, nd, fd)
    full = float(o.elbo(o.params, nd, fd, lat))
    Gterm, Aterm = float(o.last_terms["G"]), float(o.last_terms["A"].sum())
    halves = []
    for sl in (slice(0, 2), slice(2, 4)):
        sub = {k: (v[..., :, sl, :] if v.dim() >= 3 else v) for k, v in lat.items()}
        halves.append(float(o.elbo(o.params, nd, fd[sl], sub)))
    local_full = full - Gterm - Aterm
    local_halves = sum(h - Gterm - Aterm for h in halves) / 2
    assert abs(local_full - local_halves) < 1e-9 * abs(local_full)
