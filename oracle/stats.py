"""
Oracle restatement of ``snr_and_chi2`` (tapqir/utils/stats.py:29-86)  --  TEST INFRASTRUCTURE ONLY (float64 dense torch).

    gaussians = gaussian_spots(height, width, x, y, target_locs, P)            (stats.py:67-74)
    weights   = gaussians / height                                             (77)
    signal    = sum_ij (data - background - offset_mean) * weights             (78-80)
    noise     = sqrt(offset_var + background * gain);  snr = signal / noise    (81-82)
    img_ideal = background + sum_k gaussians                                   (85)
    chi2      = mean_ij (data - img_ideal - offset_mean)^2 / img_ideal         (86-88)

The reference calls it per AOI with height (K, F, Q), target_locs (F, C, 2), data (F, C, P, P): the LAST parameter axis
(Q = C) plays the role of gaussian_spots' spot axis and the first (K) rides along as a batch axis, which
``gaussians.sum(-5)`` then sums (stats.py:85).  PARITY UNPINNED (the reference module imports pyro; its tests hold no
fixture for this function); the Gaussian stack itself comes from the pinned ``oracle.dist_util.gaussian_spots``.
"""

import torch

from .dist_util import gaussian_spots


def snr_and_chi2(data, height, width, x, y, target_locs, background, gain, offset_mean, offset_var, P):
    dt = torch.float64
    data, height, width, x, y = (t.to(dt) for t in (data, height, width, x, y))
    target_locs, background = target_locs.to(dt), background.to(dt)
    gaussians = gaussian_spots(height, width, x, y, target_locs, P)  # (K, F, Q, P, P)
    weights = gaussians / height[..., None, None]
    signal = ((data - background[..., None, None] - offset_mean) * weights).sum(dim=(-2, -1))
    noise = (offset_var + background * gain).sqrt()
    img_ideal = background[..., None, None] + gaussians.sum(-5)
    chi2 = (data - img_ideal - offset_mean) ** 2 / img_ideal
    return signal / noise, chi2.mean(dim=(-1, -2))
