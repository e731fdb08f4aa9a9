"""
Synthetic-data generator with the generative law of tapqir/utils/simulate.py:12-138
(cosmos branch, ``"pi" in params``) and KSMOGN.rsample (tapqir/distributions/ksmogn.py:171-185).

The reference drives ``pyro.infer.Predictive`` over ``cosmos.model``; here the same law is
sampled directly with a seeded ``torch.Generator`` on the requested device:

  is_ontarget[:N//2] = True; target = (P-1)/2; offsets = 3 x params["offset"], weights 1/3
  z ~ Bernoulli(pi) on-target else 0; theta ~ U{1..K} if z else 0; m_k ~ Bernoulli(probs_m[theta, k])
  x_k, y_k ~ AffineBeta(0, size[theta == k+1], -(P+1)/2, (P+1)/2), size = [2, ((P+1)/(2 proximity))^2 - 1]
  image = floor(Gamma(mu / gain, 1 / gain) + offset), mu = background + sum_k m_k h N2(.)
  (crosstalk: mu_c = background + sum_q alpha[q, c] sum_k m_qk h N2(.), params["alpha"] = (Q, C) matrix)
  kinetic mode (params "kon"/"koff" or "init"/"trans", simulate.py:57-86): z follows a two-state Markov chain over the
  frames, z_0 ~ [koff, kon] / (kon + koff), P(z_f = 1 | z_{f-1}) = kon (from 0) or 1 - koff (from 1); the rest as above.
  On a HIP device the images are drawn by ``tq_ksmogn_rsample`` (KSMOGN.rsample), on the CPU by torch's sampler.
"""

import numpy as np
import torch

from tapqir_amd.distributions.util import gaussian_spots, probs_m
from tapqir_amd.utils.dataset import CosmosDataset


def _std_gamma(conc, gen):
    return torch._standard_gamma(conc, generator=gen)


def _beta(c1, c0, gen):
    a, b = _std_gamma(c1, gen), _std_gamma(c0, gen)
    return a / (a + b)


def simulate(model, N: int, F: int, C: int = 1, P: int = 14, seed: int = 0, params: dict = dict(),
             chunk: int = 64) -> CosmosDataset:
    """
    :param model: a ``cosmos`` instance (only ``K`` and ``device`` are used) or an int K.
    :return: CosmosDataset with float32 images (integer-valued), labels["z"] for the on-target AOIs.
    """
    K = model if isinstance(model, int) else model.K
    device = torch.device("cpu") if isinstance(model, int) else torch.device(model.device)
    kinetic = ("kon" in params and "koff" in params) or ("init" in params and "trans" in params)
    if "pi" not in params and not kinetic:
        raise ValueError('simulate needs "pi" (time-independent law), or "kon"/"koff" or "init"/"trans" (kinetic law)')
    # crosstalk simulations (simulate.py:27-31, 118-123): params["alpha"] is the (Q, C) matrix of the fraction of dye
    # q's signal seen in channel c (crosstalk.py:262-281)
    alpha = None
    if "alpha" in params:
        alpha = torch.as_tensor(params["alpha"], dtype=torch.float32, device=device).reshape(C, C)
    gen = torch.Generator(device=device).manual_seed(seed)
    dt = torch.float32
    H = (P + 1) / 2
    Q = C

    is_on = torch.zeros(N, dtype=torch.bool)
    is_on[: N // 2] = True
    on = is_on.to(device)
    target = torch.full((N, F, C, 2), (P - 1) / 2, dtype=dt)

    u = lambda *s: torch.rand(*s, generator=gen, device=device, dtype=dt)
    if "pi" in params:
        z = (u(N, F, Q) < params["pi"]) & on[:, None, None]
    else:
        # kinetic simulations (simulate.py:57-86; the frames of the reference's cosmos+hmm model, hmm.py:151-176, 258):
        # z_0 ~ Categorical(init), z_f | z_{f-1} ~ Categorical(trans[z_{f-1}]) for the on-target AOIs, z = 0 off-target
        # (expand_offtarget); every frame then follows the time-independent law given z
        if "kon" in params:
            kon, koff = float(params["kon"]), float(params["koff"])
            p_init, p01, p11 = kon / (kon + koff), kon, 1.0 - koff
        else:
            init = torch.as_tensor(params["init"], dtype=torch.float64).reshape(-1, 2)[0]
            trans = torch.as_tensor(params["trans"], dtype=torch.float64).reshape(-1, 2, 2)[0]
            p_init, p01, p11 = float(init[1]), float(trans[0, 1]), float(trans[1, 1])
        uz = u(N, F, Q)
        z = torch.zeros(N, F, Q, dtype=torch.bool, device=device)
        prev = uz[:, 0] < p_init
        z[:, 0] = prev
        for f in range(1, F):
            prev = uz[:, f] < torch.where(prev, torch.tensor(p11, dtype=dt, device=device), torch.tensor(p01, dtype=dt, device=device))
            z[:, f] = prev
        z &= on[:, None, None]
    theta = torch.where(z, 1 + (u(N, F, Q) * K).long().clamp(max=K - 1), torch.zeros((), dtype=torch.long, device=device))
    pm = probs_m(torch.tensor(float(params["lamda"]), dtype=torch.float64), K).to(dt).to(device)  # (1+K, K)
    m = u(N, F, Q, K) < pm[theta]  # (N,F,Q,K)
    spec = theta[..., None] == (1 + torch.arange(K, device=device))
    size = torch.where(spec, torch.tensor((H / params["proximity"]) ** 2 - 1, dtype=dt, device=device),
                       torch.tensor(2.0, dtype=dt, device=device))
    x = -H + 2 * H * _beta(size / 2, size / 2, gen)
    y = -H + 2 * H * _beta(size / 2, size / 2, gen)

    images = torch.empty(N, F, C, P, P, dtype=dt)
    h = torch.tensor(float(params["height"]), dtype=dt, device=device)
    w = torch.tensor(float(params["width"]), dtype=dt, device=device)
    gain = float(params["gain"])
    offs3 = torch.full((3,), float(params["offset"]), dtype=dt, device=device)
    logit3 = torch.full((3,), -float(np.log(3.0)), dtype=dt, device=device)
    for s in range(0, N, chunk):
        sl = slice(s, min(N, s + chunk))
        if device.type == "cuda":
            # the reference draws the images with KSMOGN.rsample (simulate.py:108-111 -> ksmogn.py:171-185): on the GPU
            # that is tq_ksmogn_rsample; seeds follow the generator of this call
            from tapqir_amd.distributions.ksmogn import KSMOGN

            n = sl.stop - sl.start
            seed_s = int(torch.randint(0, 2**62, (1,), generator=gen, device=device).item())
            if alpha is None:
                d = KSMOGN(h.expand_as(x[sl]), w.expand_as(x[sl]), x[sl], y[sl], target[sl].to(device),
                           torch.full((n, F, C), float(params["background"]), dtype=dt, device=device),
                           torch.tensor(gain, dtype=dt, device=device), offs3, logit3, P, m=m[sl].to(dt))
            else:
                d = KSMOGN(h.expand_as(x[sl]), w.expand_as(x[sl]), x[sl], y[sl], target[sl].to(device),
                           torch.full((n, F, C), float(params["background"]), dtype=dt, device=device),
                           torch.tensor(gain, dtype=dt, device=device), offs3, logit3, P, m=m[sl].to(dt), alpha=alpha)
            images[sl] = d.rsample(seed=seed_s).floor().cpu()
            continue
        tl = target[sl].to(device)[..., None, :]
        spots = gaussian_spots(h.expand_as(x[sl]), w.expand_as(x[sl]), x[sl], y[sl], tl, P, m[sl].to(dt))
        mu = spots.sum(-3)  # (n, F, Q, P, P)
        if alpha is not None:
            mu = torch.einsum("qc,nfqij->nfcij", alpha, mu)
        mu = params["background"] + mu
        val = _std_gamma(mu / gain, gen) * gain
        val = val.clamp(min=torch.finfo(dt).tiny)
        images[sl] = (val + params["offset"]).floor().cpu()

    labels = np.zeros((N // 2, F, Q), dtype=[("aoi", int), ("frame", int), ("z", int)])
    labels["aoi"] = np.arange(N // 2).reshape(-1, 1, 1)
    labels["frame"] = np.arange(F).reshape(-1, 1)
    labels["z"][:, :, :] = z[: N // 2].long().cpu().numpy()

    offset = torch.full((3,), float(params["offset"]), dtype=dt)
    return CosmosDataset(images, target, is_on, labels=labels, offset_samples=offset,
                         offset_weights=torch.ones(3, dtype=dt) / 3, device=device)


# canonical parameters of the reference's own smoke test (test/test_tapqir.py:20-50)
TEST_PARAMS = dict(width=1.4, gain=7.0, lamda=0.15, proximity=0.2, offset=90.0, height=3000, background=150, pi=0.15)
