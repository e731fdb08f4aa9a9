"""
``AffineBeta(mean, sample_size, low, high)`` -- Beta shifted to (low, high) and parameterised by its
mean and sample size (drop-in for tapqir/distributions/affine_beta.py:10-59 on top of
pyro.distributions.AffineBeta: ``rsample`` clamps into [low + eps*scale, high - eps*scale]).

The SVI step evaluates this family inside the HIP kernels (tq_site.h: tq_affine_beta_site_terms);
this torch object exists for API parity (priors, credible intervals, user code).
"""

import torch
from torch.distributions import AffineTransform, Beta, TransformedDistribution, constraints


class AffineBeta(TransformedDistribution):
    arg_constraints = {"mean": constraints.dependent, "sample_size": constraints.real,
                       "low": constraints.real, "high": constraints.dependent}

    def __init__(self, mean, sample_size, low, high, validate_args=None):
        mean, sample_size, low, high = (torch.as_tensor(v, dtype=torch.get_default_dtype()) if not torch.is_tensor(v)
                                        else v for v in (mean, sample_size, low, high))
        if bool((low != high).all()):
            concentration1 = sample_size * (mean - low) / (high - low)
            concentration0 = sample_size * (high - mean) / (high - low)
        else:  # degenerate interval (affine_beta.py:38-43)
            low, high = torch.tensor(0.0), torch.tensor(1.0)
            concentration1 = concentration0 = torch.tensor(1.0)
        self.mean_, self.sample_size, self.low, self.high = mean, sample_size, low, high
        self.scale = high - low
        super().__init__(Beta(concentration1, concentration0, validate_args=validate_args),
                         AffineTransform(loc=low, scale=high - low), validate_args=validate_args)

    @property
    def concentration1(self):
        return self.base_dist.concentration1

    @property
    def concentration0(self):
        return self.base_dist.concentration0

    @property
    def mean(self):
        return self.low + self.scale * self.base_dist.mean

    @property
    def variance(self):
        return self.scale**2 * self.base_dist.variance

    def _clamp(self, x):
        eps = torch.finfo(x.dtype).eps * self.scale
        return torch.min(torch.max(x, self.low + eps), self.high - eps)

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return self._clamp(super().sample(sample_shape))

    def rsample(self, sample_shape=torch.Size()):
        return self._clamp(super().rsample(sample_shape))
