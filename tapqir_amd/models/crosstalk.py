"""
``crosstalk`` -- multi-colour time-independent colocalisation model with cross-talk, drop-in for
tapqir/models/crosstalk.py:26-574 (marked EXPERIMENTAL upstream) on AMD MI355X.

Differences from ``cosmos`` (see ``oracle/crosstalk.py`` for the closed form that the kernels evaluate):
one data site per AOI-frame with event shape (C, P, P) whose channel ``c`` sees the spots of every dye ``q``
scaled by ``alpha[q, c]`` (crosstalk.py:262-281), a global Dirichlet site ``alpha`` (80-87, 279-284) with
parameters ``alpha_mean`` (Q, C) / ``alpha_size`` (Q, 1) (429-438), and 2^(K Q) joint spot-presence
combinations in the guide-side enumeration.  Implemented for Q = C = 2 with K <= 2, and for Q = C = 1 (where it
coincides with cosmos; the reference's code indexes dyes and channels alike, crosstalk.py:246-261).
"""

import torch

from tapqir_amd.models.cosmos import _HipTraceEnumELBO, cosmos, initial_values
from tapqir_amd.models.engine import CosmosEngine


def crosstalk_initial_values(eng, data, collective=None):
    """crosstalk.py:424-455 followed by cosmos._init_parameters."""
    v = initial_values(eng, data, collective)
    Q, C = eng.C, eng.C
    conc = torch.ones(Q, C, dtype=torch.float64) + 9 * torch.eye(Q, dtype=torch.float64)
    v["alpha_mean"] = conc / conc.sum(-1, keepdim=True)  # stored through the softmax transform (Appendix B.5)
    v["alpha_size"] = torch.full((Q, 1), 2.0, dtype=torch.float64)
    return v


class crosstalk(cosmos):
    r"""
    **Multi-Color Time-Independent Colocalization Model with Cross-Talk** (crosstalk.py:26-40).

    :param K: Maximum number of spots that can be present in a single image (1 or 2).
    :param Q: Number of fluorescent dyes (= number of channels = 2).
    :param device: Computation device; the SVI step requires ``"cuda"`` (= HIP on ROCm).
    :param dtype: accepted for compatibility; kernels compute in float32.
    :param use_pykeops: accepted and ignored.
    :param priors: Dictionary of parameters of prior distributions.
    """

    name = "crosstalk"

    def __init__(self, S: int = 1, K: int = 2, Q: int = None, device: str = "cpu", dtype: str = "double",
                 use_pykeops: bool = True, priors: dict = None):
        super().__init__(S=S, K=K, Q=Q, device=device, dtype=dtype, use_pykeops=use_pykeops, priors=priors)
        self._global_params = ["gain", "proximity", "lamda", "pi", "alpha"]  # crosstalk.py:62
        self.ci_params = ["alpha", "gain", "pi", "lamda", "proximity", "background", "height", "width", "x", "y"]

    def TraceELBO(self, jit=False):
        return _HipTraceEnumELBO(self, max_plate_nesting=2)  # crosstalk.py:457-464

    def _make_engine(self, engine_cls=None, **kw):
        if self.engine is None:
            self.engine = (engine_cls or CosmosEngine)(self.data, K=self.K, priors=self.priors, device=self.device,
                                                       crosstalk=True, **{**getattr(self, "_engine_kwargs", {}), **kw})
        return self.engine

    def init_parameters(self):
        eng = self._make_engine()
        eng.layout.set_constrained(eng.params, crosstalk_initial_values(eng, self.data, self.collective))
        eng.exp_avg.zero_()
        eng.exp_avg_sq.zero_()
        eng.grad.zero_()
        eng.reset_adam_clock(0)

    # -- posteriors (crosstalk.py:466-574) ------------------------------------------------------------------
    @property
    def compute_probs(self):
        """The sites entering the z / theta posterior (m, x, y, z, theta of each dye, crosstalk.py:473-479)
        factorise over dyes, so the joint normalisation of crosstalk.py:503-526 is the per-dye one of cosmos;
        5 particles (486-488); z_probs has shape (Nt, F, Q) = p(z_q = 1) (467, 521)."""
        if self._probs is None:
            from tapqir_amd.models.posterior import compute_probs

            z, theta = compute_probs(self, particles=5)
            self._probs = (z[..., 1], theta)
        return self._probs

    @property
    def z_map(self) -> torch.Tensor:
        return self.z_probs > 0.5  # crosstalk.py:573-574

    @torch.no_grad()
    def compute_params(self, CI):
        from tapqir_amd.utils.stats import dirichlet_interval

        out = super().compute_params(CI)
        cp = self.engine.layout.constrained(self.engine.params)
        lo, hi, mean = dirichlet_interval(cp["alpha_mean"].detach() * cp["alpha_size"].detach(), CI)
        out["alpha"] = {"LL": lo, "UL": hi, "Mean": mean}
        return out
