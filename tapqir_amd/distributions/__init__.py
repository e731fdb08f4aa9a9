"""Distributions of the drop-in surface (tapqir/distributions/__init__.py:7-14)."""

from tapqir_amd.distributions.affine_beta import AffineBeta
from tapqir_amd.distributions.ksmogn import KSMOGN, KSpotGammaNoise

__all__ = ["AffineBeta", "KSMOGN", "KSpotGammaNoise"]
