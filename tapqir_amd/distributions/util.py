"""
Small host-side probability tables of the cosmos model (API parity with
tapqir/distributions/util.py:15-173).

These are O(K^2) tables and a dense image renderer used by the synthetic-data
generator; the SVI hot path never calls them -- the HIP kernels in
``tapqir_amd/csrc`` evaluate the same closed forms on the device
(``tq_globals.h``: probs_m rows; ``tq_ksmogn.hip``: separable Gaussian spots).
"""

import math

import torch


def gaussian_spots(height, width, x, y, target_locs, P, m=None):
    """(..., K) spot parameters -> (..., K, P, P) images; x runs along the LAST axis
    (columns), y along rows (tapqir/distributions/util.py:46-48, ``indexing="xy"``)."""
    pix = torch.arange(P, dtype=height.dtype, device=height.device)
    inv2v = 0.5 / (width * width)
    norm = height / (2 * math.pi * width * width)
    if m is not None:
        norm = m * norm
    dx = pix - (x + target_locs[..., 0])[..., None]
    dy = pix - (y + target_locs[..., 1])[..., None]
    gx = torch.exp(-dx * dx * inv2v[..., None])  # (..., K, P) columns
    gy = torch.exp(-dy * dy * inv2v[..., None])  # (..., K, P) rows
    return norm[..., None, None] * gy[..., :, None] * gx[..., None, :]


def truncated_poisson_probs(lamda, K):
    """Poisson pmf for k < K, remaining mass at k = K (util.py:67-91)."""
    k = torch.arange(K, dtype=lamda.dtype, device=lamda.device)
    lam = lamda.unsqueeze(-1)
    head = torch.exp(torch.xlogy(k, lam) - lam - torch.lgamma(k + 1))
    return torch.cat([head, 1 - head.sum(-1, keepdim=True)], -1)


def _mean_count_frac(lamda, K):
    tp = truncated_poisson_probs(lamda, K)
    k = torch.arange(K + 1, dtype=lamda.dtype, device=lamda.device)
    return (k * tp).sum(-1) / K


def probs_m(lamda, K):
    """p(m_k = 1 | theta, lamda): lamda.shape + (1+K, K) (util.py:94-130)."""
    out = torch.empty(lamda.shape + (1 + K, K), dtype=lamda.dtype, device=lamda.device)
    if K > 1:
        out[..., 1:, :] = _mean_count_frac(lamda, K - 1)[..., None, None]
    out[..., 0, :] = _mean_count_frac(lamda, K)[..., None]
    idx = torch.arange(K, device=lamda.device)
    out[..., idx + 1, idx] = 1
    return out


def expand_offtarget(probs):
    """probs (..., S+1) -> (..., S+1, 2): [..., 0] off-target law, [..., 1] probs (util.py:133-151)."""
    off = torch.zeros_like(probs)
    off[..., 0] = 1
    return torch.stack([off, probs], dim=-1)


def probs_theta(K, device=torch.device("cpu")):
    """p(theta | z): (2, 1+K) (util.py:154-173)."""
    out = torch.zeros(2, 1 + K, device=device)
    out[0, 0] = 1
    out[1, 1:] = 1 / K
    return out
