"""
The two off-step users of the spot render (tapqir_amd/csrc/tq_aux.h):

  * post-fit SNR / chi2 (tapqir/utils/stats.py:29-86, 166-182): kernel ``tq_snr_chi2`` (and the g++ build of the same
    body) against the oracle restatement ``oracle.stats.snr_and_chi2`` -- tolerance 1e-5 relative (fp32 sums of 196 terms);
  * ``KSMOGN.rsample`` (tapqir/distributions/ksmogn.py:171-185): kernel ``tq_ksmogn_rsample`` against the analytic moments
    of the law (mean mu + E[offset], variance mu g + Var[offset]), the offset-category frequencies, and the oracle's
    sampler (``oracle.ksmogn.ksmogn_rsample``) through a two-sample comparison of quantiles.  Draws are distributionally,
    not bitwise, those of torch.
"""

import ctypes as C

import pytest
import torch

from helpers import HostCheckEngine, load_hostcheck, make_dataset, make_oracle, oracle_to_engine, rel_err
from oracle.ksmogn import ksmogn_image, ksmogn_rsample
from oracle.stats import snr_and_chi2 as snr_chi2_oracle
from tapqir_amd import _lib


def _oracle_snr(eng, d, o):
    cp = {n: v.detach() for n, v in o.constrained(o.params).items()}
    snr = torch.zeros(eng.K, eng.Nt, eng.F, eng.C, dtype=torch.float64)
    chi2 = torch.zeros(eng.Nt, eng.F, eng.C, dtype=torch.float64)
    for n in range(eng.Nt):  # the reference's loop (stats.py:168-182)
        snr[:, n], chi2[n] = snr_chi2_oracle(d.images[n], cp["h_loc"][:, n], cp["w_mean"][:, n], cp["x_mean"][:, n],
                                            cp["y_mean"][:, n], d.xy[n], cp["b_loc"][n], cp["gain_loc"],
                                            d.offset.mean, d.offset.var, d.P)
    return snr, chi2


def _check_snr(eng, d, o):
    snr, chi2 = eng.snr_chi2(d.offset.mean, d.offset.var)
    snr_o, chi2_o = _oracle_snr(eng, d, o)
    assert rel_err(snr.cpu(), snr_o) < 1e-5
    assert rel_err(chi2.cpu(), chi2_o) < 1e-5


@pytest.mark.parametrize("dkw,K", [(dict(N=3, F=4), 2), (dict(N=2, F=3, C=2), 2), (dict(N=2, F=2, P=20), 3), (dict(N=2, F=2, P=9), 1)])
def test_snr_chi2_host_build_matches_oracle(dkw, K):
    d = make_dataset(K=K, **dkw)
    o = make_oracle(d, K, perturb=0.3)
    eng = HostCheckEngine(d, K=K, device="cpu")
    oracle_to_engine(o, eng)
    _check_snr(eng, d, o)


@pytest.mark.gpu
@pytest.mark.parametrize("dkw,K", [(dict(N=3, F=4), 2), (dict(N=2, F=3, C=2), 2), (dict(N=2, F=2, P=20), 3), (dict(N=40, F=50), 2)])
def test_snr_chi2_kernel_matches_oracle(dkw, K):
    from tapqir_amd.models.engine import CosmosEngine

    d = make_dataset(K=K, **dkw)
    o = make_oracle(d, K, perturb=0.3)
    eng = CosmosEngine(d, K=K, device="cuda:0")
    oracle_to_engine(o, eng)
    _check_snr(eng, d, o)
    # the reference's public function on the tensors of ONE AOI (stats.py:29-44), through the same kernel
    from tapqir_amd.utils.stats import snr_and_chi2

    cp = {n: v.detach().cuda() for n, v in o.constrained(o.params).items()}
    snr1, chi1 = snr_and_chi2(d.images[1].cuda(), cp["h_loc"][:, 1], cp["w_mean"][:, 1], cp["x_mean"][:, 1], cp["y_mean"][:, 1],
                              d.xy[1].cuda(), cp["b_loc"][1], cp["gain_loc"], d.offset.mean, d.offset.var, d.P)
    snr_o, chi_o = _oracle_snr(eng, d, o)
    assert rel_err(snr1.cpu(), snr_o[:, 1]) < 1e-5 and rel_err(chi1.cpu(), chi_o[1]) < 1e-5


# ---- rsample ---------------------------------------------------------------------------------------------------------
def _rsample_problem(S, dev):
    """One unit with two spots, replicated S times (independent pixel streams): (args tensors, mu, offsets, weights)."""
    P, K = 14, 2
    f32 = torch.float32
    h = torch.tensor([[3000.0], [800.0]], dtype=f32).expand(K, S).contiguous()
    w = torch.tensor([[1.4], [1.1]], dtype=f32).expand(K, S).contiguous()
    x = torch.tensor([[0.6], [-3.2]], dtype=f32).expand(K, S).contiguous()
    y = torch.tensor([[-0.4], [2.5]], dtype=f32).expand(K, S).contiguous()
    xy = torch.full((S, 2), 6.5, dtype=f32)
    b = torch.full((S,), 150.0, dtype=f32)
    gain = torch.tensor([7.0], dtype=f32)
    offs = torch.tensor([86.0, 90.0, 97.0], dtype=f32)
    wts = torch.tensor([0.2, 0.5, 0.3], dtype=f32)
    t = {"h": h, "w": w, "x": x, "y": y, "xy": xy, "b": b, "gain": gain, "offs": offs, "logits": wts.log()}
    t = {k: v.to(dev) for k, v in t.items()}
    mu = ksmogn_image(h[:, 0].double(), w[:, 0].double(), x[:, 0].double(), y[:, 0].double(), xy[0].double(),
                      b[0].double(), P)  # (P, P)
    return t, mu, offs.double(), wts.double(), P, K


def _rsample_args(t, S, P, K, seed):
    out = torch.empty(S, P, P, dtype=torch.float32, device=t["h"].device)
    a = _lib.RsampleArgs()
    p = _lib.ptr
    a.height, a.width, a.x, a.y, a.xy, a.background, a.gain = p(t["h"]), p(t["w"]), p(t["x"]), p(t["y"]), p(t["xy"]), p(t["b"]), p(t["gain"])
    a.offset_samples, a.offset_logits, a.out = p(t["offs"]), p(t["logits"]), p(out)
    a.B, a.P, a.K, a.O, a.seed = S, P, K, 3, seed
    return a, out


def _check_law(out, mu, offs, wts, S):
    out = out.double().cpu()
    om, ov = float((offs * wts).sum()), float((offs**2 * wts).sum() - (offs * wts).sum() ** 2)
    mean, var = out.mean(0), out.var(0)
    sd = (mu * 7.0 + ov).sqrt()
    # per-pixel mean within 5 standard errors, variance within 12 % (S draws per pixel)
    assert float(((mean - (mu + om)).abs() / (sd / S**0.5)).max()) < 5.0
    assert float((var / (mu * 7.0 + ov) - 1).abs().max()) < 0.12
    # the pooled z-scores are standard: mean 0, variance 1 to 1 %
    z = (out - (mu + om)) / sd
    assert abs(float(z.mean())) < 0.01 and abs(float(z.var()) - 1) < 0.01
    # a two-sample check against the oracle's sampler (torch._standard_gamma + multinomial) at one bright and one dark pixel
    g = torch.Generator().manual_seed(1)
    q = torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95], dtype=torch.float64)
    for (j, i) in ((6, 7), (0, 0)):
        alpha = (mu[j, i] / 7.0).expand(200000)
        ref = torch._standard_gamma(alpha, generator=g) * 7.0 + offs[torch.multinomial(wts, 200000, replacement=True, generator=g)]
        dq = (torch.quantile(out[:, j, i], q) - torch.quantile(ref, q)).abs() / sd[j, i]
        assert float(dq.max()) < 0.06, (j, i, dq)


def test_rsample_host_build_has_the_law():
    hc = load_hostcheck()
    S = 4000
    t, mu, offs, wts, P, K = _rsample_problem(S, "cpu")
    a, out = _rsample_args(t, S, P, K, seed=11)
    hc.hc_ksmogn_rsample(C.byref(a))
    assert (out > 86.0).all()
    _check_law(out, mu, offs, wts, S)


@pytest.mark.gpu
def test_rsample_kernel_has_the_law_and_matches_the_host_streams():
    S = 20000
    t, mu, offs, wts, P, K = _rsample_problem(S, "cuda")
    a, out = _rsample_args(t, S, P, K, seed=11)
    _lib.check(_lib.load().tq_ksmogn_rsample(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tq_ksmogn_rsample")
    torch.cuda.synchronize()
    _check_law(out, mu, offs, wts, S)
    # same Philox streams as the g++ build of the body (rare rejection flips from the fast intrinsics allowed)
    tc, *_ = _rsample_problem(256, "cpu")
    ac, outc = _rsample_args(tc, 256, P, K, seed=11)
    load_hostcheck().hc_ksmogn_rsample(C.byref(ac))
    same = ((out[:256].cpu() - outc).abs() <= 1e-3 * outc.abs()).double().mean()
    assert same > 0.99, same


@pytest.mark.gpu
def test_ksmogn_distribution_rsample_and_simulate_on_the_device():
    """KSMOGN.rsample (cosmos and crosstalk branch) and simulate() on the GPU: shapes, integer images, means."""
    from tapqir_amd.distributions import KSMOGN
    from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    K, P, n = 2, 14, 3000
    f = lambda v: torch.full((n, K), v, device=dev)
    d = KSMOGN(f(3000.0), f(1.4), f(0.0), f(0.0), torch.full((n, 2), 6.5, device=dev), torch.full((n,), 150.0, device=dev),
               torch.tensor(7.0, device=dev), torch.full((3,), 90.0, device=dev), torch.full((3,), 1 / 3, device=dev).log(), P,
               m=torch.tensor([1.0, 0.0], device=dev).expand(n, K))
    s = d.rsample()
    assert s.shape == (n, P, P)
    want = d.image[0].double().cpu() + 90.0
    assert float(((s.double().mean(0).cpu() - want).abs() / (want * 7.0 / n).sqrt()).max()) < 6.0
    # crosstalk branch: channel c sees alpha[q, c] of dye q
    alpha = torch.tensor([[0.9, 0.1], [0.2, 0.8]], device=dev)
    g = lambda v: torch.full((n, 2, K), v, device=dev)
    dx = KSMOGN(g(3000.0), g(1.4), g(0.0), g(0.0), torch.full((n, 2, 2), 6.5, device=dev), torch.full((n, 2), 150.0, device=dev),
                torch.tensor(7.0, device=dev), torch.full((3,), 90.0, device=dev), torch.full((3,), 1 / 3, device=dev).log(), P,
                m=torch.tensor([[1.0, 0.0], [0.0, 0.0]], device=dev).expand(n, 2, K), alpha=alpha)
    sx = dx.rsample()
    assert sx.shape == (n, 2, P, P)
    wantx = dx.image[0].double().cpu() + 90.0
    assert float(((sx.double().mean(0).cpu() - wantx).abs() / (wantx * 7.0 / n).sqrt()).max()) < 6.0
    assert float(wantx[1].max() - 240.0) < 0.2 * float(wantx[0].max() - 240.0)  # 10 % of dye 0 leaks into channel 1

    class M:
        pass

    M.K, M.device = 2, dev
    data = simulate(M, 6, 50, 1, 14, seed=4, params=TEST_PARAMS)
    assert data.images.shape == (6, 50, 1, 14, 14) and bool((data.images == data.images.floor()).all())
    assert 230.0 < float(data.images.median()) < 245.0  # background 150 + offset 90 (floored)
    again = simulate(M, 6, 50, 1, 14, seed=4, params=TEST_PARAMS)
    assert torch.equal(again.images, data.images)  # same seed, same data
