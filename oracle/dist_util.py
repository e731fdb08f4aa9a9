"""
Oracle restatement of ``tapqir/distributions/util.py`` (TEST INFRASTRUCTURE ONLY).

Every function cites the reference lines it follows.  Written for clarity, in
float64, with explicit closed forms rather than the reference's tensor idioms.
Pinned by ``tests/golden/util_golden.npz`` (made by importing the reference's
own util.py, see ``tests/golden/make_golden.py``).
"""

import math

import torch


def gaussian_spots(height, width, x, y, target_locs, P, m=None):
    """2-D Gaussian spot images.  Follows tapqir/distributions/util.py:15-64.

    ``height, width, x, y``: (..., K); ``target_locs``: (..., 1, 2) broadcastable
    against (..., K, 2); returns (..., K, P, P) with the LAST axis running along
    x (columns) and the second-to-last along y (rows) -- util.py:46-48 builds the
    pixel grid with ``indexing="xy"``.

    mu_S[row j, col i] = m*h / (2 pi w^2) * exp(-((i - x - tx)^2 + (j - y - ty)^2) / (2 w^2))
    """
    pix = torch.arange(P, dtype=height.dtype, device=height.device)
    cx = (x + target_locs[..., 0])[..., None]  # (..., K, 1)
    cy = (y + target_locs[..., 1])[..., None]
    w = width[..., None]
    log_norm = -w.log() - 0.5 * math.log(2 * math.pi)
    ex = -((pix - cx) ** 2) / (2 * w**2) + log_norm  # (..., K, P) along columns
    ey = -((pix - cy) ** 2) / (2 * w**2) + log_norm  # (..., K, P) along rows
    # util.py:55-61 sums the two exponents before exponentiating; do the same
    g = torch.exp(ey[..., :, None] + ex[..., None, :])
    if m is not None:
        height = m * height
    return height[..., None, None] * g


def truncated_poisson_probs(lamda, K):
    """util.py:67-91.  Returns lamda.shape + (K+1,):
    P(k) = lamda^k e^-lamda / k!  for k < K, and the remaining mass at k = K."""
    cols = []
    for k in range(K):
        # xlogy(k, lamda): 0 * log(lamda) := 0 for k = 0 (util.py:87)
        logp = (k * lamda.log() if k > 0 else torch.zeros_like(lamda)) - lamda - math.lgamma(k + 1)
        cols.append(logp.exp())
    if K > 0:
        head = torch.stack(cols, -1)
        tail = 1 - head.sum(-1, keepdim=True)
        return torch.cat([head, tail], -1)
    return torch.ones(lamda.shape + (1,), dtype=lamda.dtype)


def probs_m(lamda, K):
    """Prior spot-presence probability p(m_k = 1 | theta, lamda); util.py:94-130.

    Returns lamda.shape + (1+K, K):
      row theta = 0      : sum_{l=1..K}   l * TruncPois(l; lamda, K)   / K
      row theta = k + 1  : 1 in column k,
                           sum_{l=1..K-1} l * TruncPois(l; lamda, K-1) / (K-1) elsewhere.
    For K = 1 the "elsewhere" value is 0/0 in the reference (util.py:119-123) but
    every entry is overwritten (127-129); here it is simply never written.
    """
    out = torch.zeros(lamda.shape + (1 + K, K), dtype=lamda.dtype)
    if K > 1:
        tp = truncated_poisson_probs(lamda, K - 1)
        other = sum(l * tp[..., l] for l in range(1, K)) / (K - 1)
        out[..., :, :] = other[..., None, None]
    tp = truncated_poisson_probs(lamda, K)
    out[..., 0, :] = (sum(l * tp[..., l] for l in range(1, K + 1)) / K)[..., None]
    for k in range(K):
        out[..., k + 1, k] = 1
    return out


def expand_offtarget(probs):
    """util.py:133-151: probs (..., S+1) -> (..., S+1, 2); [..., 0] is the
    off-target law [1, 0, ..., 0], [..., 1] is ``probs`` itself."""
    off = torch.zeros_like(probs)
    off[..., 0] = 1
    return torch.stack([off, probs], -1)


def probs_theta(K, dtype=torch.float64):
    """util.py:154-173: p(theta | z): (2, 1+K); z=0 -> [1,0..0]; z=1 -> [0,1/K..1/K]."""
    out = torch.zeros(2, 1 + K, dtype=dtype)
    out[0, 0] = 1
    out[1, 1:] = 1 / K
    return out
