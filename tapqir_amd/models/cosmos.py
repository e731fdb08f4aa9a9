"""
``cosmos`` -- multi-colour time-independent colocalisation model, drop-in for
tapqir/models/cosmos.py:28-784 on AMD MI355X.

The probabilistic program of the reference (``model`` 82-327, ``guide`` 329-462 under pyro's
``TraceEnum_ELBO``) is not traced here: its closed-form ELBO is evaluated,
differentiated and optimised by hand-written HIP kernels (``tapqir_amd/csrc``) through the C ABI in
``include/tapqir_hip.h``.  Parameter names, shapes, constraints and initial values are those of
``init_parameters`` / ``_init_parameters`` (464-598); checkpoints use the reference's key set.
"""

import math

import torch

from tapqir_amd.models.engine import DEFAULT_PRIORS, CosmosEngine
from tapqir_amd.models.model import Model


def data_median(images, collective=None, chunk=1 << 26):
    """Per-channel median over ALL pixels of ``images (Nt, F, C, P, P)`` = ``CosmosDataset.median``
    (tapqir/utils/dataset.py:134-138: ``torch.median``, i.e. the lower of the two middle values), computed where the
    images are (the engine's HBM-resident copy).

    Integer-valued, non-negative images -- every real dataset (glimpse_reader.py:376-378 stores integers) and the
    simulator (simulate.py:118-122 floors) -- take an exact counting median: a histogram over the integer values, which
    in an AOI-sharded fit is summed over the ranks (``collective``) so that every rank starts from the median of the whole
    dataset, whatever the sharding.  Other data use ``torch.median`` of the local pixels."""
    C = images.shape[2]
    out = []
    for c in range(C):
        x = images[:, :, c].reshape(-1)
        lo, hi = x.min(), x.max()
        span = torch.stack([lo, -hi]).double().cpu()
        if collective is not None:
            collective.reduce_(span, "min")
        lo_i, hi_i = int(span[0]), int(-span[1])
        integral = bool(float(span[0]) >= 0 and -float(span[1]) < (1 << 24))
        if integral:
            for s in range(0, x.numel(), chunk):
                xs = x[s:s + chunk]
                if not bool((xs == xs.floor()).all()):
                    integral = False
                    break
        if collective is not None:
            integral = not collective.any(not integral)
        if not integral:
            out.append(torch.median(x).double().cpu())
            continue
        hist = torch.zeros(hi_i - lo_i + 1, dtype=torch.int64, device=x.device)
        for s in range(0, x.numel(), chunk):
            hist += torch.bincount(x[s:s + chunk].to(torch.int64) - lo_i, minlength=hist.numel())
        hist = hist.cpu()
        if collective is not None:
            collective.reduce_(hist, "sum")
        k = (int(hist.sum()) - 1) // 2  # rank of the lower median
        out.append(torch.tensor(float(lo_i + int(torch.searchsorted(torch.cumsum(hist, 0), k, right=True))),
                                dtype=torch.float64))
    return torch.stack(out)


def initial_values(eng, data, collective=None):
    """Constrained initial values, cosmos.py:471-598."""
    K, Nt, F, C = eng.K, eng.Nt, eng.F, eng.C
    Q = C
    f = lambda shape, v: torch.full(shape, float(v), dtype=torch.float64)
    # (a streamed data set keeps its images in host memory: the same counting median there)
    med = data_median(eng.images if eng.images is not None else eng.images_host, collective)
    # (the reference takes median - offset.mean as it is; a non-positive value has no `positive` pre-image there)
    bg = (med - data.offset.mean).clamp(min=1e-3)
    return {
        "pi_mean": torch.full((Q, 2), 0.5, dtype=torch.float64),  # ones -> softmax -> uniform (Appendix B.5)
        "pi_size": f((Q, 1), 2), "m_probs": f((K, Nt, F, Q), 0.5),
        "proximity_loc": f((), 0.5), "proximity_size": f((), 100),
        "lamda_loc": f((Q,), 0.5), "lamda_beta": f((Q,), 100),
        "gain_loc": f((), 5), "gain_beta": f((), 100),
        "background_mean_loc": bg.expand(Nt, 1, C), "background_std_loc": f((Nt, 1, C), 1),
        "b_loc": bg.expand(Nt, F, C), "b_beta": f((Nt, F, C), 1),
        "h_loc": f((K, Nt, F, Q), 2000), "h_beta": f((K, Nt, F, Q), 0.001),
        "w_mean": f((K, Nt, F, Q), 1.5), "w_size": f((K, Nt, F, Q), 100),
        "x_mean": f((K, Nt, F, Q), 0), "y_mean": f((K, Nt, F, Q), 0), "size": f((K, Nt, F, Q), 200),
    }


class _HipTraceEnumELBO:
    """Stand-in for ``pyro.infer.TraceEnum_ELBO(max_plate_nesting=3)`` (cosmos.py:600-607): the
    object ``Model.init`` keeps as ``self.elbo``.  ``loss`` / ``loss_and_grads`` evaluate -ELBO of a
    fresh minibatch with the HIP kernels (gradients land in ``engine.grad``)."""

    def __init__(self, model, max_plate_nesting=3):
        self._m = model
        self.max_plate_nesting = max_plate_nesting

    def loss_and_grads(self, model=None, guide=None, *args, **kwargs):
        m = self._m
        eng = m.engine
        ndx, fdx = m._subsample()
        a = eng.make_args(ndx, fdx)
        for name in ("cosmos_sample_globals", "cosmos_sample_locals", "cosmos_elbo_grads", "cosmos_globals_grad"):
            eng.call(name, a)
        return -float(eng.elbo_out.item())

    loss = loss_and_grads
    differentiable_loss = loss_and_grads


class cosmos(Model):
    r"""
    **Multi-Color Time-Independent Colocalization Model** (cosmos.py:28-45).

    :param K: Maximum number of spots that can be present in a single image.
    :param device: Computation device; the SVI step requires ``"cuda"`` (= HIP on ROCm).
    :param dtype: accepted for compatibility ("double"/"float"); kernels compute in float32.
    :param use_pykeops: accepted and ignored (the fused HIP kernel replaces the KeOps reduction).
    :param priors: Dictionary of parameters of prior distributions.
    """

    name = "cosmos"

    def __init__(self, S: int = 1, K: int = 2, Q: int = None, device: str = "cpu", dtype: str = "double",
                 use_pykeops: bool = True, priors: dict = None):
        if S != 1:
            raise NotImplementedError("cosmos supports S=1 molecular state (cosmos.py:251 clamps z to {0,1})")
        priors = dict(DEFAULT_PRIORS) if priors is None else priors
        super().__init__(S=S, K=K, Q=Q, device=device, dtype=dtype, priors=priors)
        self._global_params = ["gain", "proximity", "lamda", "pi"]
        self.use_pykeops = use_pykeops
        self.conv_params = ["-ELBO", "proximity_loc", "gain_loc", "lamda_loc"]
        self.ci_params = ["gain", "pi", "lamda", "proximity", "background", "height", "width", "x", "y"]
        self._subsample_gen = torch.Generator().manual_seed(0)
        self._probs = None
        self.allreduce = None  # set by tapqir_amd.parallel for AOI-sharded data parallelism

    # -- descriptions of the probabilistic program (no tracing happens) ---------------------------
    def model(self):
        """Generative model: see cosmos.py:82-327; evaluated in closed form by the HIP kernels."""
        raise NotImplementedError("the generative model is fused into the HIP step; see DESIGN.md")

    def guide(self):
        """Variational family: see cosmos.py:329-462; sampled by tq_cosmos_sample_* kernels."""
        raise NotImplementedError("the guide is fused into the HIP step; see DESIGN.md")

    def TraceELBO(self, jit=False):
        return _HipTraceEnumELBO(self, max_plate_nesting=3)

    # -- engine / parameters -------------------------------------------------------------------------
    def _make_engine(self, engine_cls=None, **kw):
        if self.engine is None:
            kw = {**getattr(self, "_engine_kwargs", {}), **kw}  # AOI sharding: tapqir_amd.parallel.attach
            self.engine = (engine_cls or CosmosEngine)(self.data, K=self.K, priors=self.priors, device=self.device, **kw)
        return self.engine

    def init_parameters(self):
        """cosmos.py:464-598."""
        eng = self._make_engine()
        eng.layout.set_constrained(eng.params, initial_values(eng, self.data, self.collective))
        eng.exp_avg.zero_()
        eng.exp_avg_sq.zero_()
        eng.grad.zero_()
        eng.reset_adam_clock(0)

    # -- one SVI step ----------------------------------------------------------------------------------
    def _subsample(self):
        """pyro.plate(size, subsample_size): randperm(size)[:subsample_size], identity when equal
        (cosmos.py:194-208; Appendix B.2); ``self.n`` / ``self.f`` fix the subsample (197, 205)."""
        d = self.data
        nb = self.nbatch_size or d.Nt
        fb = self.fbatch_size or d.F
        if self.n is not None:
            ndx = torch.as_tensor(self.n)
        elif nb >= d.Nt:
            ndx = None
        else:
            ndx = torch.randperm(d.Nt, generator=self._subsample_gen)[:nb]
        if self.f is not None:
            fdx = torch.as_tensor(self.f)
        elif fb >= d.F:
            fdx = None
        else:
            fdx = torch.randperm(d.F, generator=self._subsample_gen)[:fb]
        return ndx, fdx

    def step_async(self) -> None:
        """One SVI update, nothing read back (the -ELBO stays on the device)."""
        eng = self.engine
        d = self.data
        if (self.n is None and self.f is None and self.allreduce is None and eng.device.type == "cuda"
                and hasattr(eng, "step_subsampled")
                and eng.step_subsampled(self.nbatch_size or d.Nt, self.fbatch_size or d.F, self._subsample_gen)):
            # the subsample of this step was drawn on the device by the previous launch (CosmosEngine.step_subsampled)
            self._probs = None
            return
        if self.n is None and self.f is None and eng.device.type == "cuda" and hasattr(eng, "draw_subsample") and not eng.streamed:
            # the same two randperm draws as _subsample, made straight into the engine's pinned staging ring
            d = self.data
            ndx, fdx = eng.draw_subsample(self.nbatch_size or d.Nt, self.fbatch_size or d.F, self._subsample_gen)
        else:
            ndx, fdx = self._subsample()
        eng.step(ndx, fdx, allreduce=self.allreduce)
        self._probs = None

    def last_loss(self) -> float:
        self.engine.join()
        return -float(self.engine.elbo_out.item())

    def step(self) -> float:
        """Replaces ``self.svi.step()`` (model.py:212): applies one Adam update and returns the
        minibatch -ELBO; raises ValueError on a non-finite loss so that ``run``'s recovery path
        (model.py:220-232) works."""
        self.step_async()
        loss = self.last_loss()  # the ELBO every rank of a sharded fit reports is the global one: same branch everywhere
        if not math.isfinite(loss):
            raise ValueError(f"Iteration #{getattr(self, 'iter', 0)}. Non-finite loss {loss}")
        return loss

    # -- posteriors (cosmos.py:609-709) ------------------------------------------------------------------
    @property
    def m_probs(self) -> torch.Tensor:
        r"""Posterior spot presence probability :math:`q(m=1)`."""
        return torch.sigmoid(self.engine.named("params")["m_probs"]).detach()

    @property
    def compute_probs(self):
        if self._probs is None:
            from tapqir_amd.models.posterior import compute_probs

            self._probs = compute_probs(self)
        return self._probs

    @property
    def z_probs(self) -> torch.Tensor:
        r"""Probability of there being a target-specific spot :math:`p(z=1)`."""
        return self.compute_probs[0]

    @property
    def theta_probs(self) -> torch.Tensor:
        r"""Posterior target-specific spot probability :math:`q(\theta = k)`."""
        return self.compute_probs[1]

    @property
    def pspecific(self) -> torch.Tensor:
        return self.z_probs

    @property
    def z_map(self) -> torch.Tensor:
        return torch.argmax(self.z_probs, dim=-1)

    def z_sample(self, num_samples):
        """cosmos.py:706-709."""
        return torch.distributions.Categorical(self.params["z_probs"][: self.data.N]).sample((num_samples,))

    @torch.no_grad()
    def compute_params(self, CI):
        """Credible intervals / means of the variational posteriors and the spot probabilities
        (cosmos.py:711-784; same keys, tensors on the CPU)."""
        from tapqir_amd.utils.stats import affine_beta_interval, dirichlet_interval, gamma_interval

        cp = {n: v.detach() for n, v in self.engine.layout.constrained(self.engine.params).items()}
        P, pr = self.data.P, self.priors
        H = (P + 1) / 2
        out = {}

        def put(name, tup):
            out[name] = {"LL": tup[0], "UL": tup[1], "Mean": tup[2]}

        put("gain", gamma_interval(cp["gain_loc"], cp["gain_beta"], CI))
        put("pi", dirichlet_interval(cp["pi_mean"] * cp["pi_size"], CI))
        put("lamda", gamma_interval(cp["lamda_loc"], cp["lamda_beta"], CI))
        put("proximity", affine_beta_interval(cp["proximity_loc"], cp["proximity_size"], 0.0, (P + 1) / math.sqrt(12), CI))
        put("background", gamma_interval(cp["b_loc"], cp["b_beta"], CI))
        put("height", gamma_interval(cp["h_loc"], cp["h_beta"], CI))
        put("width", affine_beta_interval(cp["w_mean"], cp["w_size"], pr["width_min"], pr["width_max"], CI))
        put("x", affine_beta_interval(cp["x_mean"], cp["size"], -H, H, CI))
        put("y", affine_beta_interval(cp["y_mean"], cp["size"], -H, H, CI))
        out["m_probs"] = self.m_probs.cpu()
        out["z_probs"] = self.z_probs.cpu()
        out["theta_probs"] = self.theta_probs.cpu()
        out["z_map"] = self.z_map.cpu()
        out["p_specific"] = out["theta_probs"].sum(0)
        return out


Cosmos = cosmos  # notebooks/part_iii_colab.ipynb:98 imports the capitalised spelling
