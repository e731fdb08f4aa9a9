"""Distribution objects of the drop-in surface."""

import math

import pytest
import torch

from oracle.cosmos import AffineBeta as OracleAffineBeta
from oracle.ksmogn import ksmogn_log_prob
from tapqir_amd.distributions import KSMOGN, AffineBeta, KSpotGammaNoise
from tapqir_amd.exceptions import HipExtensionError


def test_affine_beta_matches_oracle_definition():
    torch.manual_seed(0)
    mean, size, lo, hi = torch.tensor([0.3, -2.0]), torch.tensor([50.0, 7.0]), -7.5, 7.5
    d = AffineBeta(mean.double(), size.double(), torch.tensor(lo).double(), torch.tensor(hi).double())
    o = OracleAffineBeta(mean.double(), size.double(), lo, hi)
    y = torch.tensor([1.234, -5.5], dtype=torch.float64)
    assert torch.allclose(d.log_prob(y), o.log_prob(y), rtol=0, atol=1e-12)
    assert torch.allclose(d.mean, mean.double())
    s = d.rsample((4000,))
    assert (s > lo).all() and (s < hi).all()
    assert abs(float(s[:, 0].mean()) - 0.3) < 0.1


def test_ksmogn_refuses_cpu_tensors():
    P = 6
    z = torch.zeros(1)
    d = KSMOGN(z + 100, z + 1.4, z, z, torch.tensor([2.5, 2.5]), torch.tensor(50.0), torch.tensor(5.0),
               torch.tensor([90.0]), torch.tensor([0.0]), P)
    assert d.batch_shape == torch.Size([]) and d.event_shape == torch.Size([P, P])
    assert KSpotGammaNoise is KSMOGN
    with pytest.raises(HipExtensionError):
        d.log_prob(torch.full((P, P), 150.0))
    with pytest.raises(HipExtensionError):
        d.rsample()


@pytest.mark.gpu
@pytest.mark.parametrize("O", [1, 12])
def test_ksmogn_log_prob_and_gradients_on_device(O):
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    N, F, K, P = 3, 4, 2, 14
    val = torch.floor(240 + 60 * torch.rand(N, F, P, P, generator=g))
    h = 500 + 3000 * torch.rand(N, F, K, generator=g)
    w = 1.0 + torch.rand(N, F, K, generator=g)
    x = 4 * torch.rand(N, F, K, generator=g) - 2
    y = 4 * torch.rand(N, F, K, generator=g) - 2
    m = (torch.rand(N, F, K, generator=g) > 0.5).float()
    tl = torch.full((N, F, 2), 6.5)
    b = 140 + 20 * torch.rand(N, F, generator=g)
    gain = torch.tensor(7.0)
    offs = torch.arange(85.0, 85.0 + O)
    logits = torch.log_softmax(torch.randn(O, generator=g), 0)
    leaves = [t.clone().to(dev).requires_grad_(True) for t in (h, w, x, y, b, gain)]
    d = KSMOGN(leaves[0], leaves[1], leaves[2], leaves[3], tl.to(dev), leaves[4], leaves[5], offs.to(dev),
               logits.to(dev), P, m=m.to(dev))
    lp = d.log_prob(val.to(dev))
    assert lp.shape == (N, F)
    wts = torch.randn(N, F, generator=g)
    (lp * wts.to(dev)).sum().backward()
    # oracle in float64
    ol = [t.clone().double().requires_grad_(True) for t in (h, w, x, y, b, gain)]
    ref = ksmogn_log_prob(val.double(), ol[0], ol[1], ol[2], ol[3], tl.double(), ol[4], ol[5], offs.double(),
                          logits.double(), P, m.double())
    (ref * wts.double()).sum().backward()
    assert (lp.detach().cpu().double() - ref.detach()).abs().max() <= 2e-6 * ref.detach().abs().max()
    for name, a, r in zip("hwxybg", leaves, ol):
        ga, gr = a.grad.cpu().double(), r.grad
        # the gain gradient is a sum over all units of cancelling per-unit terms (each ~100x the sum here)
        tol = 1e-3 if name == "g" else 1e-4
        assert (ga - gr).abs().max() <= tol * gr.abs().max(), (name, ga, gr)


def test_ksmogn_crosstalk_shapes_and_image():
    """ksmogn.py:119-144: with alpha the batch shape drops Q and K, channels move into the event."""
    from oracle.ksmogn import ksmogn_crosstalk_image

    g = torch.Generator().manual_seed(1)
    N, F, Q, K, P = 2, 3, 2, 2, 8
    h = 500 + 3000 * torch.rand(N, F, Q, K, generator=g).double()
    w = 1.0 + torch.rand(N, F, Q, K, generator=g).double()
    x = 2 * torch.rand(N, F, Q, K, generator=g).double() - 1
    m = (torch.rand(N, F, Q, K, generator=g) > 0.5).double()
    tl = torch.full((N, F, 2, 2), 3.5).double()
    b = 150 + torch.rand(N, F, 2, generator=g).double()
    alpha = torch.tensor([[0.9, 0.1], [0.2, 0.8]]).double()
    d = KSMOGN(h, w, x, -x, tl, b, torch.tensor(7.0).double(), torch.tensor([90.0]).double(), torch.tensor([0.0]).double(),
               P, m=m, alpha=alpha)
    assert d.batch_shape == (N, F) and d.event_shape == (2, P, P)
    assert torch.allclose(d.image, ksmogn_crosstalk_image(h, w, x, -x, tl, b, P, m, alpha), rtol=1e-13)
    with pytest.raises(HipExtensionError):  # sampling is tq_ksmogn_rsample on the device (tests/test_aux.py)
        d.rsample()


@pytest.mark.gpu
@pytest.mark.parametrize("O", [1, 9])
def test_ksmogn_crosstalk_log_prob_and_gradients_on_device(O):
    from oracle.ksmogn import ksmogn_crosstalk_log_prob

    dev = "cuda:0"
    g = torch.Generator().manual_seed(2)
    N, F, Q, K, P = 2, 3, 2, 2, 14
    val = torch.floor(240 + 60 * torch.rand(N, F, 2, P, P, generator=g))
    h = 500 + 3000 * torch.rand(N, F, Q, K, generator=g)
    w = 1.0 + torch.rand(N, F, Q, K, generator=g)
    x = 4 * torch.rand(N, F, Q, K, generator=g) - 2
    y = 4 * torch.rand(N, F, Q, K, generator=g) - 2
    m = (torch.rand(N, F, Q, K, generator=g) > 0.4).float()
    tl = 6.5 + 0.3 * torch.rand(N, F, 2, 2, generator=g)
    b = 140 + 20 * torch.rand(N, F, 2, generator=g)
    gain = torch.tensor(7.0)
    alpha = torch.tensor([[0.85, 0.15], [0.25, 0.75]])
    offs = torch.arange(85.0, 85.0 + O)
    logits = torch.log_softmax(torch.randn(O, generator=g), 0)
    leaves = [t.clone().to(dev).requires_grad_(True) for t in (h, w, x, y, b, gain, alpha)]
    d = KSMOGN(leaves[0], leaves[1], leaves[2], leaves[3], tl.to(dev), leaves[4], leaves[5], offs.to(dev), logits.to(dev),
               P, m=m.to(dev), alpha=leaves[6])
    lp = d.log_prob(val.to(dev))
    assert lp.shape == (N, F)
    wts = torch.randn(N, F, generator=g)
    (lp * wts.to(dev)).sum().backward()
    ol = [t.clone().double().requires_grad_(True) for t in (h, w, x, y, b, gain, alpha)]
    ref = ksmogn_crosstalk_log_prob(val.double(), ol[0], ol[1], ol[2], ol[3], tl.double(), ol[4], ol[5], offs.double(),
                                    logits.double(), P, m.double(), ol[6])
    (ref * wts.double()).sum().backward()
    assert (lp.detach().cpu().double() - ref.detach()).abs().max() <= 2e-6 * ref.detach().abs().max()
    for name, a, r in zip(["h", "w", "x", "y", "b", "g", "alpha"], leaves, ol):
        ga, gr = a.grad.cpu().double(), r.grad
        tol = 1e-3 if name in ("g", "alpha") else 1e-4  # sums over all units of cancelling per-unit terms
        assert (ga - gr).abs().max() <= tol * gr.abs().max(), (name, ga, gr)
