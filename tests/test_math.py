"""Scalar building blocks of the kernels (tq_math.h, g++ build) against torch / known answers."""

import ctypes as C

import numpy as np
import torch

from helpers import load_hostcheck


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_philox4x32_10_known_answers():
    """Random123 known-answer vectors for philox4x32-10 (counter, key) -> output.
    tq_philox_init maps (seed, step, site, elem) to counter = (0, elem_lo, elem_hi ^ site<<20, step),
    key = seed, so the vectors below are reachable through the public stream constructor."""
    hc = load_hostcheck()
    out = (C.c_uint32 * 4)()
    hc.hc_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.c_int]
    hc.hc_philox(0, 0, 0, 0, out, 4)
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]


def test_lgamma_digamma():
    hc = load_hostcheck()
    a = _f32(np.concatenate([np.logspace(-3, 5, 400), [0.5, 1, 2, 7.999, 8.0, 8.001, 22500.0]]))
    lg, dg = np.empty_like(a), np.empty_like(a)
    hc.hc_lgamma_digamma(_p(a), _p(lg), _p(dg), C.c_int64(a.size))
    t = torch.tensor(a, dtype=torch.float64)
    ref_lg, ref_dg = torch.lgamma(t).numpy(), torch.digamma(t).numpy()
    assert np.all(np.abs(lg - ref_lg) <= 6e-6 * np.maximum(1.0, np.abs(ref_lg)))  # fp32: the a<8 path forms terms of magnitude ~10
    assert np.all(np.abs(dg - ref_dg) <= 3e-6 * np.maximum(1.0, np.abs(ref_dg)))


def test_standard_gamma_grad_matches_torch():
    """Every branch of the implicit reparameterisation gradient (small x / saddle point / rational)."""
    hc = load_hostcheck()
    g = torch.Generator().manual_seed(0)
    alpha = torch.cat([10 ** (torch.rand(4000, generator=g) * 6 - 2), torch.tensor([0.3, 1.0, 8.0, 8.01, 150.0, 2000.0])])
    x = torch._standard_gamma(alpha.double(), generator=g).clamp(min=1e-30)
    # also the |x - alpha| < 0.1 alpha special branch and far tails
    alpha = torch.cat([alpha, torch.full((200,), 50.0), torch.full((200,), 50.0)])
    x = torch.cat([x, 50.0 * (0.9 + 0.2 * torch.rand(200, generator=g).double()), 50.0 * (0.3 + 2 * torch.rand(200, generator=g).double())])
    a32, x32 = _f32(alpha.numpy()), _f32(x.numpy())
    out = np.empty_like(a32)
    hc.hc_std_gamma_grad(_p(a32), _p(x32), _p(out), C.c_int64(a32.size))
    ref = torch._standard_gamma_grad(torch.tensor(a32, dtype=torch.float64), torch.tensor(x32, dtype=torch.float64)).numpy()
    assert np.all(np.abs(out - ref) <= 2e-5 * np.abs(ref) + 1e-30)


def test_dirichlet_grad_matches_torch():
    hc = load_hostcheck()
    g = torch.Generator().manual_seed(1)
    torch.manual_seed(11)  # Beta(...).sample() below draws from the global generator
    n = 60000
    c1 = 10 ** (torch.rand(n, generator=g) * 4.5 - 1.5)
    c0 = 10 ** (torch.rand(n, generator=g) * 4.5 - 1.5)
    x = torch.distributions.Beta(c1.double(), c0.double()).sample().clamp(1e-6, 1 - 1e-6)
    # force the x ~ mean branch of the saddle-point expansion too
    c1 = torch.cat([c1, torch.full((100,), 80.0)])
    c0 = torch.cat([c0, torch.full((100,), 120.0)])
    x = torch.cat([x, torch.full((100,), 0.4, dtype=torch.float64) + 1e-3 * torch.randn(100, generator=g).double()])
    x32, a32, t32 = _f32(x.numpy()), _f32(c1.numpy()), _f32((c1 + c0).numpy())
    out = np.empty_like(x32)
    hc.hc_dirichlet_grad(_p(x32), _p(a32), _p(t32), _p(out), C.c_int64(x32.size))
    ref = torch._dirichlet_grad(torch.tensor(x32, dtype=torch.float64), torch.tensor(a32, dtype=torch.float64),
                                torch.tensor(t32, dtype=torch.float64)).numpy()
    assert np.all(np.abs(out - ref) <= 5e-5 * np.abs(ref) + 1e-12)


def test_gamma_sampler_moments_and_ks():
    from scipy import stats

    hc = load_hostcheck()
    hc.hc_sample_std_gamma.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int64]
    n = 40000
    for alpha in (0.05, 0.7, 1.0, 2.0, 21.5, 300.0, 22500.0):
        a = _f32(np.full(n, alpha))
        out = np.empty_like(a)
        hc.hc_sample_std_gamma(1234, 7, 3, _p(a), _p(out), n)
        assert np.all(out >= 0) and np.all(np.isfinite(out))
        assert abs(out.mean() - alpha) < 5 * np.sqrt(alpha / n)
        assert stats.kstest(out.astype(np.float64), "gamma", args=(alpha,)).pvalue > 1e-4, alpha
    # streams differ across steps / sites / elements
    a = _f32(np.full(8, 3.0))
    o1, o2 = np.empty_like(a), np.empty_like(a)
    hc.hc_sample_std_gamma(1, 0, 0, _p(a), _p(o1), 8)
    hc.hc_sample_std_gamma(1, 1, 0, _p(a), _p(o2), 8)
    assert not np.any(o1 == o2) and len(set(o1.tolist())) == 8


def test_mean_frac_is_probs_m_row_zero():
    from oracle.dist_util import probs_m

    hc = load_hostcheck()
    hc.hc_mean_frac.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    for K in (1, 2, 3, 4):
        for lam in (1e-3, 0.15, 1.0, 3.3):
            v, dv = C.c_double(), C.c_double()
            hc.hc_mean_frac(lam, K, C.byref(v), C.byref(dv))
            t = torch.tensor(lam, dtype=torch.float64, requires_grad=True)
            ref = probs_m(t, K)[0, 0]
            (gr,) = torch.autograd.grad(ref, t)
            assert abs(v.value - float(ref)) < 1e-14 and abs(dv.value - float(gr)) < 1e-13


def test_fused_beta_gradient_pair_matches_torch():
    """tq_beta_grad_pair_mid (both directions of one Beta draw at once) against torch._dirichlet_grad."""
    hc = load_hostcheck()
    g = torch.Generator().manual_seed(2)
    torch.manual_seed(12)
    n = 20000
    c1 = 10 ** (torch.rand(n, generator=g) * 3.5 + 0.3)
    c0 = 10 ** (torch.rand(n, generator=g) * 3.5 + 0.3)
    x = torch.distributions.Beta(c1.double(), c0.double()).sample().clamp(1e-6, 1 - 1e-6)
    x32, a32, b32 = _f32(x.numpy()), _f32(c1.numpy()), _f32(c0.numpy())
    ga, gb = np.empty_like(x32), np.empty_like(x32)
    fused = np.zeros(x32.size, dtype=np.uint8)
    hc.hc_beta_grad_pair(_p(x32), _p(a32), _p(b32), _p(ga), _p(gb), _p(fused), C.c_int64(x32.size))
    assert fused.mean() > 0.5  # the fused path is the common one in this regime
    X = torch.tensor(x32, dtype=torch.float64)
    A, B = torch.tensor(a32, dtype=torch.float64), torch.tensor(b32, dtype=torch.float64)
    T = (torch.tensor(a32) + torch.tensor(b32)).double()  # the kernel forms total in float32
    ref_a = torch._dirichlet_grad(X, A, T).numpy()
    ref_b = torch._dirichlet_grad((1 - torch.tensor(x32)).double(), T - A, T).numpy()
    assert np.all(np.abs(ga - ref_a) <= 1e-4 * np.abs(ref_a) + 1e-12)
    assert np.all(np.abs(gb - ref_b) <= 1e-4 * np.abs(ref_b) + 1e-12)


def test_regime_ordered_beta_gradient_pair_is_the_two_plain_calls():
    """tq_beta_grad_pair_rest evaluates both directions of a Beta draw regime by regime (each expensive regime once per
    wave instead of once per direction); its results are those of the two tq_dirichlet_grad calls."""
    hc = load_hostcheck()
    g = torch.Generator().manual_seed(4)
    n = 60000
    c1 = 10 ** (torch.rand(n, generator=g) * 3.3 - 0.3)  # 0.5 .. 1000: every regime of the piecewise scheme
    c0 = 10 ** (torch.rand(n, generator=g) * 3.3 - 0.3)
    torch.manual_seed(3)
    x = torch.distributions.Beta(c1.double(), c0.double()).sample().clamp(1e-6, 1 - 1e-6)
    x[:200] = 0.5  # both directions in the same series regime
    x32, a32, b32 = _f32(x.numpy()), _f32(c1.numpy()), _f32(c0.numpy())
    out = [np.empty_like(x32) for _ in range(4)]
    hc.hc_beta_grad_pair_rest(_p(x32), _p(a32), _p(b32), *[_p(o) for o in out], C.c_int64(x32.size))
    ga, gb, ra, rb = out
    assert np.isfinite(ga).all() and np.isfinite(gb).all()
    # series and rational regimes: the same code on the same arguments; a direction in the saddle-point regime comes from
    # the pair routine (same formulas, one reciprocal for five): rounding-level differences
    same = (ga.view(np.uint32) == ra.view(np.uint32)) & (gb.view(np.uint32) == rb.view(np.uint32))
    assert same.mean() > 0.5
    assert np.all(np.abs(ga - ra) <= 5e-6 * np.abs(ra)) and np.all(np.abs(gb - rb) <= 5e-6 * np.abs(rb))
