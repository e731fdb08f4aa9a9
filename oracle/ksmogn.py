"""
Oracle restatement of ``tapqir/distributions/ksmogn.py`` (TEST INFRASTRUCTURE ONLY).

K-Spots Marginalised-Offset Gamma-Noise image likelihood:

    mu[j,i]  = b + sum_k m_k h_k N(i; x_k + tx, w_k) N(j; y_k + ty, w_k)      (ksmogn.py:146-165)
    alpha    = mu / g ,  beta = 1 / g                                          (ksmogn.py:109,167-169)
    log p(D) = sum_{j,i} log sum_o w_o Gamma(D[j,i] - delta_o ; alpha[j,i], beta) 1[D[j,i] > delta_o]
                                                                               (ksmogn.py:222-238)
"""

import math

import torch

from .dist_util import gaussian_spots


def ksmogn_image(height, width, x, y, target_locs, background, P, m=None):
    """ksmogn.py:146-165 (cosmos branch, alpha is None).
    height/width/x/y/m: (..., K); target_locs (..., 2); background (...,) -> (..., P, P)."""
    g = gaussian_spots(height, width, x, y, target_locs.unsqueeze(-2), P, m)
    return background[..., None, None] + g.sum(-3)


def ksmogn_log_prob(value, height, width, x, y, target_locs, background, gain,
                    offset_samples, offset_logits, P, m=None):
    """Dense-torch branch of KSMOGN.log_prob, ksmogn.py:222-238.  value (..., P, P)."""
    image = ksmogn_image(height, width, x, y, target_locs, background, P, m)
    rate = 1 / gain
    conc = (image / gain).unsqueeze(-1)  # (..., P, P, 1)
    v = value.unsqueeze(-1)
    mask = v > offset_samples
    nv = torch.where(mask, v - offset_samples, torch.ones((), dtype=v.dtype))
    obs = conc * torch.log(rate) + (conc - 1) * torch.log(nv) - rate * nv - torch.lgamma(conc)
    res = torch.logsumexp(obs + offset_logits + torch.log(mask.to(v.dtype)), -1)
    return res.sum((-1, -2))


def ksmogn_log_prob_bruteforce(value, height, width, x, y, target_locs, background, gain,
                               offset_samples, offset_weights, P, m=None):
    """Independent per-pixel / per-offset loop with python scalars (small inputs only).
    All tensor args are for ONE image: height.. (K,), target_locs (2,), background ()."""
    K = height.shape[0]
    g = float(gain)
    total = 0.0
    for j in range(P):
        for i in range(P):
            mu = float(background)
            for k in range(K):
                mk = 1.0 if m is None else float(m[k])
                w = float(width[k])
                dx = i - float(x[k]) - float(target_locs[0])
                dy = j - float(y[k]) - float(target_locs[1])
                mu += mk * float(height[k]) / (2 * math.pi * w * w) * math.exp(-(dx * dx + dy * dy) / (2 * w * w))
            a = mu / g
            acc = 0.0
            for o in range(offset_samples.shape[0]):
                d = float(value[j, i]) - float(offset_samples[o])
                if d > 0:
                    lp = a * math.log(1 / g) + (a - 1) * math.log(d) - d / g - math.lgamma(a)
                    acc += float(offset_weights[o]) * math.exp(lp)
            total += math.log(acc) if acc > 0 else -math.inf
    return total


def ksmogn_rsample(height, width, x, y, target_locs, background, gain, offset_samples,
                   offset_logits, P, m=None, generator=None):
    """ksmogn.py:171-185: per-pixel offset ~ Categorical(logits), value = Gamma(alpha, beta) + offset."""
    image = ksmogn_image(height, width, x, y, target_locs, background, P, m)
    probs = torch.softmax(offset_logits, -1)
    odx = torch.multinomial(probs, image.numel(), replacement=True, generator=generator).reshape(image.shape)
    conc = image / gain
    if generator is None:
        val = torch._standard_gamma(conc) * gain
    else:
        val = torch._standard_gamma(conc, generator=generator) * gain
    val = val.clamp(min=torch.finfo(val.dtype).tiny)
    return val + offset_samples[odx]


def ksmogn_crosstalk_image(height, width, x, y, target_locs, background, P, m, alpha):
    """Crosstalk branch of KSMOGN (ksmogn.py:93-105, 146-165).
    height/width/x/y/m: (..., Q, K); target_locs (..., C, 2); background (..., C); alpha (Q, C)
    -> (..., C, P, P):  image_c = b_c + sum_q alpha_qc sum_k m_qk h_qk N(x_qk + tx_c, y_qk + ty_c; w_qk)."""
    h = height.unsqueeze(-2) * alpha[..., None]  # (..., Q, C, K)
    g = gaussian_spots(h, width.unsqueeze(-2), x.unsqueeze(-2), y.unsqueeze(-2),
                       target_locs.unsqueeze(-3).unsqueeze(-2), P,
                       None if m is None else m.unsqueeze(-2))  # (..., Q, C, K, P, P)
    return background[..., None, None] + g.sum(-5).sum(-3)


def ksmogn_crosstalk_log_prob(value, height, width, x, y, target_locs, background, gain,
                              offset_samples, offset_logits, P, m, alpha):
    """Dense-torch branch (ksmogn.py:222-238) with event shape (C, P, P).  value (..., C, P, P)."""
    image = ksmogn_crosstalk_image(height, width, x, y, target_locs, background, P, m, alpha)
    rate = 1 / gain
    conc = (image / gain).unsqueeze(-1)
    v = value.unsqueeze(-1)
    mask = v > offset_samples
    nv = torch.where(mask, v - offset_samples, torch.ones((), dtype=v.dtype))
    obs = conc * torch.log(rate) + (conc - 1) * torch.log(nv) - rate * nv - torch.lgamma(conc)
    res = torch.logsumexp(obs + offset_logits + torch.log(mask.to(v.dtype)), -1)
    return res.sum((-1, -2, -3))
