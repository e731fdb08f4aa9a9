"""
Oracle restatement of the cosmos SVI step (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows, as text:
  * tapqir/models/cosmos.py:82-327   (model: sites, priors, masks, enumeration)
  * tapqir/models/cosmos.py:329-462  (guide)
  * tapqir/models/cosmos.py:464-598  (parameters, constraints, initial values)
  * tapqir/models/cosmos.py:609-672  (compute_probs)
  * tapqir/models/model.py:153-186   (Adam lr / betas; SVI)
  * tapqir/utils/dataset.py:18-37, 134-151 (offset logits/mean, median, fetch)
and, for what lives in pyro-ppl (>=1.8.5, not installed here), SURVEY.md Appendix A/B:
TraceEnum_ELBO = exact expectation over the guide-enumerated m_k (Dice weights),
exact log-space marginalisation of the model-enumerated z, theta together with the
factors that carry their dims (m_k, x_k, y_k), masks before plate scaling.

PARITY UNPINNED for the ELBO semantics (no Pyro, no reference golden vectors); pinned
pieces: dist_util (reference import), per-site densities (torch.distributions), and the
independent brute-force enumerator ``elbo_bruteforce`` below.

The program is deliberately *dense*: it materialises the (2^K, nb, fb, C, K, P, P)
Gaussian stack and the (..., P, P, O) mixture and uses autograd -- the same tensor program
Pyro would execute -- so it also serves as the CPU baseline of bench.py.
"""

import contextlib
import itertools
import math

import torch
import torch.distributions as D
from torch.distributions import constraints, transform_to
from torch.distributions.utils import probs_to_logits

from .dist_util import expand_offtarget, probs_m, probs_theta
from .ksmogn import ksmogn_log_prob

# Working dtype of the restatement.  float64 is what `tapqir fit` runs (main.py:428) and what every parity test uses;
# bench.py's cpu_baseline leg also times a float32 pass "for information" (BASELINE.md section 3).
_DT = [torch.float64]


@contextlib.contextmanager
def working_dtype(dtype):
    old, _DT[0] = _DT[0], dtype
    try:
        yield
    finally:
        _DT[0] = old


def _t(v):
    """python scalars -> float64 tensors (the reference runs with float64 as default dtype, main.py:428)."""
    return torch.as_tensor(v, dtype=_DT[0])


DEFAULT_PRIORS = {  # tapqir/models/cosmos.py:55-64
    "background_mean_std": 1000.0,
    "background_std_std": 100.0,
    "lamda_rate": 1.0,
    "height_std": 10000.0,
    "width_min": 0.75,
    "width_max": 2.25,
    "proximity_rate": 1.0,
    "gain_std": 50.0,
}


# ----------------------------------------------------------------------------------------
# distributions that live in pyro in the reference
# ----------------------------------------------------------------------------------------
class AffineBeta:
    """tapqir/distributions/affine_beta.py:10-49 on top of pyro.distributions.AffineBeta
    (SURVEY Appendix B.4): Y = low + (high-low) * Beta(c1, c0),
    c1 = size (mean-low)/(high-low), c0 = size (high-mean)/(high-low);
    rsample clamps into [low + eps*scale, high - eps*scale]."""

    def __init__(self, mean, size, low, high):
        self.low, self.scale = low, high - low
        self.c1 = size * (mean - low) / (high - low)
        self.c0 = size * (high - mean) / (high - low)
        self.base = D.Beta(self.c1, self.c0)

    def from_base(self, t):
        y = self.low + self.scale * t
        eps = torch.finfo(y.dtype).eps * self.scale
        return torch.min(torch.max(y, torch.as_tensor(self.low + eps, dtype=y.dtype)),
                         torch.as_tensor(self.low + self.scale - eps, dtype=y.dtype))

    def rsample(self):
        return self.from_base(self.base.rsample())

    def log_prob(self, y):
        t = (y - self.low) / self.scale
        return self.base.log_prob(t) - math.log(self.scale)


class _GammaGiven(torch.autograd.Function):
    """standard Gamma draw supplied from outside; backward = torch's implicit
    reparameterisation gradient (what ``torch._standard_gamma`` registers)."""

    @staticmethod
    def forward(ctx, alpha, g):
        ctx.save_for_backward(alpha, g)
        return g.clone()

    @staticmethod
    def backward(ctx, go):
        alpha, g = ctx.saved_tensors
        return go * torch._standard_gamma_grad(alpha, g), None


class _DirichletGiven(torch.autograd.Function):
    """Dirichlet draw supplied from outside; backward as torch.distributions.dirichlet._Dirichlet_backward."""

    @staticmethod
    def forward(ctx, conc, x):
        ctx.save_for_backward(conc, x)
        return x.clone()

    @staticmethod
    def backward(ctx, go):
        conc, x = ctx.saved_tensors
        total = conc.sum(-1, True).expand_as(conc)
        grad = torch._dirichlet_grad(x, conc, total)
        return grad * (go - (x * go).sum(-1, True)), None


def gamma_from_base(alpha, beta, g):
    """torch.distributions.Gamma.rsample with the standard-gamma draw given."""
    alpha, beta = torch.broadcast_tensors(alpha, beta)
    return _GammaGiven.apply(alpha.contiguous(), g) / beta


def beta_from_base(c1, c0, t):
    """torch.distributions.Beta.rsample (= Dirichlet([c1,c0]).rsample()[...,0]) with the draw given."""
    c1, c0 = torch.broadcast_tensors(c1, c0)
    conc = torch.stack([c1, c0], -1)
    x = torch.stack([t, 1 - t], -1)
    return _DirichletGiven.apply(conc, x)[..., 0]


# ----------------------------------------------------------------------------------------
class OracleData:
    """The slice of tapqir/utils/dataset.py (CosmosDataset/OffsetData) the step needs."""

    def __init__(self, images, xy, is_ontarget, offset_samples, offset_weights, mask=None):
        dt = _DT[0]
        self.images = images.to(dt)  # (Nt, F, C, P, P)
        self.xy = xy.to(dt)  # (Nt, F, C, 2)
        self.is_ontarget = is_ontarget.bool()
        self.mask = torch.ones_like(self.is_ontarget) if mask is None else mask.bool()
        self.offset_samples = offset_samples.to(dt)
        self.offset_weights = offset_weights.to(dt)
        self.Nt, self.F, self.C, self.P = images.shape[:4]

    @property
    def offset_logits(self):  # dataset.py:27-29
        return probs_to_logits(self.offset_weights)

    @property
    def offset_mean(self):  # dataset.py:31-33
        return float((self.offset_samples * self.offset_weights).sum())

    @property
    def median(self):  # dataset.py:134-138
        return torch.stack([torch.median(self.images[..., c, :, :]) for c in range(self.C)])


class CosmosOracle:
    LOCAL_K = ["m_probs", "h_loc", "h_beta", "w_mean", "w_size", "x_mean", "y_mean", "size"]

    def __init__(self, data, K=2, priors=None, eps=None):
        """``eps`` stands for ``torch.finfo(self.dtype).eps`` of the *model* dtype
        (cosmos.py:499-502, 568-593).  The arithmetic here is always float64; pass
        eps=finfo(float32).eps to mirror a float32 build."""
        self.data, self.K, self.S = data, K, 1
        self.Q = data.C
        self.priors = dict(DEFAULT_PRIORS if priors is None else priors)
        self.eps = torch.finfo(torch.float64).eps if eps is None else eps
        self.H = (data.P + 1) / 2
        self.constraints = self._constraints()
        self.params = None

    # -- cosmos.py:464-598 ---------------------------------------------------------------
    def _constraints(self):
        P, e, H = self.data.P, self.eps, self.H
        c = constraints
        return {
            "pi_mean": c.simplex, "pi_size": c.positive, "m_probs": c.unit_interval,
            "proximity_loc": c.interval(0.0, (P + 1) / math.sqrt(12) - e),
            "proximity_size": c.greater_than(2.0),
            "lamda_loc": c.positive, "lamda_beta": c.positive,
            "gain_loc": c.positive, "gain_beta": c.positive,
            "background_mean_loc": c.positive, "background_std_loc": c.positive,
            "b_loc": c.positive, "b_beta": c.positive, "h_loc": c.positive, "h_beta": c.positive,
            "w_mean": c.interval(0.75 + e, 2.25 - e), "w_size": c.greater_than(2.0),
            "x_mean": c.interval(-H + e, H - e), "y_mean": c.interval(-H + e, H - e),
            "size": c.greater_than(2.0),
        }

    def init_values(self):
        d, K, Q = self.data, self.K, self.Q
        f = lambda shape, v: torch.full(shape, float(v), dtype=_DT[0])
        bg = (d.median - d.offset_mean)
        return {
            "pi_mean": torch.ones(Q, 2, dtype=_DT[0]), "pi_size": f((Q, 1), 2),
            "m_probs": f((K, d.Nt, d.F, Q), 0.5),
            "proximity_loc": f((), 0.5), "proximity_size": f((), 100),
            "lamda_loc": f((Q,), 0.5), "lamda_beta": f((Q,), 100),
            "gain_loc": f((), 5), "gain_beta": f((), 100),
            "background_mean_loc": bg.expand(d.Nt, 1, d.C).clone(),
            "background_std_loc": f((d.Nt, 1, d.C), 1),
            "b_loc": bg.expand(d.Nt, d.F, d.C).clone(), "b_beta": f((d.Nt, d.F, d.C), 1),
            "h_loc": f((K, d.Nt, d.F, Q), 2000), "h_beta": f((K, d.Nt, d.F, Q), 0.001),
            "w_mean": f((K, d.Nt, d.F, Q), 1.5), "w_size": f((K, d.Nt, d.F, Q), 100),
            "x_mean": f((K, d.Nt, d.F, Q), 0), "y_mean": f((K, d.Nt, d.F, Q), 0),
            "size": f((K, d.Nt, d.F, Q), 200),
        }

    def init_parameters(self, values=None):
        """pyro.param stores transform_to(constraint).inv(init) as the leaf (Appendix B.5)."""
        values = self.init_values() if values is None else values
        self.params = {}
        for name, v in values.items():
            u = transform_to(self.constraints[name]).inv(v.to(_DT[0]))
            self.params[name] = u.detach().clone().requires_grad_(True)
        return self.params

    def constrained(self, params=None):
        params = self.params if params is None else params
        return {n: transform_to(self.constraints[n])(u) for n, u in params.items()}

    # -- guide, cosmos.py:329-462 -----------------------------------------------------------
    def _guide_dists(self, cp, ndx, fdx):
        d, K = self.data, self.K
        n_, f_ = ndx[:, None], fdx[None, :]
        loc = lambda name: cp[name][:, n_, f_, :]  # (K, nb, fb, Q)
        H, Hs = self.H, (d.P + 1) / math.sqrt(12)
        g = {}
        g["gain"] = D.Gamma(cp["gain_loc"] * cp["gain_beta"], cp["gain_beta"])
        g["pi"] = D.Dirichlet(cp["pi_mean"] * cp["pi_size"])
        g["lamda"] = D.Gamma(cp["lamda_loc"] * cp["lamda_beta"], cp["lamda_beta"])
        g["proximity"] = AffineBeta(cp["proximity_loc"], cp["proximity_size"], 0.0, Hs)
        g["background"] = D.Gamma(cp["b_loc"][n_, f_, :] * cp["b_beta"][n_, f_, :], cp["b_beta"][n_, f_, :])
        g["m_probs"] = loc("m_probs")
        g["height"] = D.Gamma(loc("h_loc") * loc("h_beta"), loc("h_beta"))
        g["width"] = AffineBeta(loc("w_mean"), loc("w_size"), self.priors["width_min"], self.priors["width_max"])
        g["x"] = AffineBeta(loc("x_mean"), loc("size"), -H, H)
        g["y"] = AffineBeta(loc("y_mean"), loc("size"), -H, H)
        return g

    def sample_guide(self, params, ndx, fdx):
        """Native torch rsample of every continuous guide site (differentiable)."""
        g = self._guide_dists(self.constrained(params), ndx, fdx)
        return {k: g[k].rsample() for k in
                ["gain", "pi", "lamda", "proximity", "background", "height", "width", "x", "y"]}

    @staticmethod
    def base_draws(lat, params_constrained_dists):
        """Recover the base draws (standard gammas / unit betas) behind native samples."""
        g = params_constrained_dists
        t = lambda ab, y: ((y - ab.low) / ab.scale)
        return {
            "gain_g": (lat["gain"] * g["gain"].rate).detach(),
            "lamda_g": (lat["lamda"] * g["lamda"].rate).detach(),
            "pi_x": lat["pi"].detach(),
            "proximity_t": t(g["proximity"], lat["proximity"]).detach(),
            "b_g": (lat["background"] * g["background"].rate).detach(),
            "h_g": (lat["height"] * g["height"].rate).detach(),
            "w_t": t(g["width"], lat["width"]).detach(),
            "x_t": t(g["x"], lat["x"]).detach(),
            "y_t": t(g["y"], lat["y"]).detach(),
        }

    def latents_from_base(self, params, ndx, fdx, base):
        """Same reparameterised latents, but with the base draws given (so that a HIP
        step fed the same draws can be compared gradient-for-gradient)."""
        g = self._guide_dists(self.constrained(params), ndx, fdx)
        gam = lambda dist, b: gamma_from_base(dist.concentration, dist.rate, b)
        ab = lambda dist, t: dist.from_base(beta_from_base(dist.c1, dist.c0, t))
        return {
            "gain": gam(g["gain"], base["gain_g"]),
            "lamda": gam(g["lamda"], base["lamda_g"]),
            "pi": _DirichletGiven.apply(g["pi"].concentration, base["pi_x"]),
            "proximity": ab(g["proximity"], base["proximity_t"]),
            "background": gam(g["background"], base["b_g"]),
            "height": gam(g["height"], base["h_g"]),
            "width": ab(g["width"], base["w_t"]),
            "x": ab(g["x"], base["x_t"]),
            "y": ab(g["y"], base["y_t"]),
        }

    # -- ELBO -----------------------------------------------------------
    def m_grid(self):
        """(M=2^K, K) table; combo index mi has bit k = m_k."""
        K = self.K
        return torch.tensor([[(mi >> k) & 1 for k in range(K)] for mi in range(2**K)], dtype=_DT[0])

    def zt_marginal(self, lat, ndx, parts=False):
        """log sum_{z,theta} p(z) p(theta|z) prod_k p(m_k|theta) [p(x_k|theta) p(y_k|theta)]^{m_k}
        -> (M, nb, fb, C); cosmos.py:242-300.  With parts=True also returns the
        (z, theta, M, nb, fb, C) joint (used by compute_probs)."""
        d, K, H = self.data, self.K, self.H
        mg = self.m_grid()  # (M, K)
        on = d.is_ontarget[ndx].long()
        pi_e = expand_offtarget(lat["pi"])  # (Q, 2, 2)
        pz = pi_e[:, :, on].permute(2, 0, 1)  # (nb, Q, 2[z])
        log_pz = D.Categorical(probs=pz).logits.permute(2, 0, 1)[:, :, None, :]  # (z, nb, 1, C)
        log_pt = D.Categorical(probs=probs_theta(K, _DT[0])).logits  # (z, theta)
        pm = probs_m(lat["lamda"], K)  # (Q, 1+K, K)
        size = torch.stack([torch.full_like(lat["proximity"], 2.0), (H / lat["proximity"]) ** 2 - 1], -1)
        joint = log_pz[:, None, None] + log_pt[:, :, None, None, None, None]  # (z, theta, 1, nb, 1, C)
        for k in range(K):
            # Bernoulli(probs_m[q, theta, k]).log_prob(m_k): (theta, M, C)
            bern = D.Bernoulli(probs=pm[:, :, k].T[:, None, :])  # (theta, 1, C)
            lpm = bern.log_prob(mg[:, k][None, :, None].expand(1 + K, -1, self.Q))  # (theta, M, C)
            spec = torch.tensor([1 if th == k + 1 else 0 for th in range(1 + K)])
            sz = size[spec]  # (theta,)
            lpx = AffineBeta(0.0, sz[:, None, None, None], -H, H).log_prob(lat["x"][k][None])  # (theta, nb, fb, C)
            lpy = AffineBeta(0.0, sz[:, None, None, None], -H, H).log_prob(lat["y"][k][None])
            term = lpm[:, :, None, None, :] + mg[:, k][None, :, None, None, None] * (lpx + lpy)[:, None]
            joint = joint + term[None]
        L = torch.logsumexp(joint.flatten(0, 1), 0)
        return (L, joint) if parts else L

    def elbo(self, params, ndx, fdx, lat):
        d, K, H, pr = self.data, self.K, self.H, self.priors
        cp = self.constrained(params)
        g = self._guide_dists(cp, ndx, fdx)
        nb, fb = len(ndx), len(fdx)
        s_n = d.Nt / nb
        s = s_n * d.F / fb
        mask = d.mask[ndx].to(_DT[0])[:, None, None]
        n_, f_ = ndx[:, None], fdx[None, :]

        # global sites (cosmos.py:170-184 / 342-368)
        G = D.HalfNormal(_t(pr["gain_std"])).log_prob(lat["gain"]) - g["gain"].log_prob(lat["gain"])
        G = G + (D.Dirichlet(torch.full((self.Q, 2), 0.5, dtype=_DT[0])).log_prob(lat["pi"])
                 - g["pi"].log_prob(lat["pi"])).sum()
        G = G + (D.Exponential(_t(pr["lamda_rate"])).log_prob(lat["lamda"]) - g["lamda"].log_prob(lat["lamda"])).sum()
        G = G + D.Exponential(_t(pr["proximity_rate"])).log_prob(lat["proximity"]) - g["proximity"].log_prob(lat["proximity"])

        # per-AOI sites (cosmos.py:221-227 / 397-404; Delta guide contributes 0)
        bm = cp["background_mean_loc"][ndx]  # (nb, 1, C)
        bs = cp["background_std_loc"][ndx]
        A = D.HalfNormal(_t(pr["background_mean_std"])).log_prob(bm) + D.HalfNormal(_t(pr["background_std_std"])).log_prob(bs)

        # per-frame sites
        b = lat["background"]
        E = D.Gamma((bm / bs) ** 2, bm / bs**2).log_prob(b) - g["background"].log_prob(b)  # (nb, fb, C)

        mg = self.m_grid()  # (M, K)
        p = g["m_probs"]  # (K, nb, fb, C)
        logq_m = sum(D.Bernoulli(probs=p[k]).log_prob(mg[:, k][:, None, None, None].expand(-1, nb, fb, self.Q))
                     for k in range(K))  # (M, nb, fb, C)
        W = logq_m.exp()

        # spot sites masked by m_k > 0 (cosmos.py:268-300 / 426-462): model - guide, (K, nb, fb, C)
        T = (D.HalfNormal(_t(pr["height_std"])).log_prob(lat["height"])
             + AffineBeta(torch.tensor(1.5, dtype=_DT[0]), 2.0, pr["width_min"], pr["width_max"]).log_prob(lat["width"])
             - g["height"].log_prob(lat["height"]) - g["width"].log_prob(lat["width"])
             - g["x"].log_prob(lat["x"]) - g["y"].log_prob(lat["y"]))

        L = self.zt_marginal(lat, ndx)  # (M, nb, fb, C)

        # data site (cosmos.py:310-327): enumerated m broadcast in front
        obs = d.images[n_, f_]  # (nb, fb, C, P, P)
        xy = d.xy[n_, f_]
        st = lambda v: v.permute(1, 2, 3, 0)  # (nb, fb, C, K)
        ll = ksmogn_log_prob(obs, st(lat["height"]), st(lat["width"]), st(lat["x"]), st(lat["y"]), xy, b,
                             lat["gain"], d.offset_samples, d.offset_logits, d.P,
                             m=mg[:, None, None, None, :])  # (M, nb, fb, C)

        inner = ll + L + (mg[:, :, None, None, None] * T[None]).sum(1) - logq_m
        E = E + (W * inner).sum(0)
        self.last_terms = {"G": G, "A": A, "E": E, "ll": ll, "L": L, "T": T, "W": W}
        return G + s_n * (mask * A).sum() + s * (mask * E).sum()

    # -- one SVI step (model.py:169-183, 212) ---------------------------------------------------
    def make_optim(self, lr=0.005):
        self.optim = {n: torch.optim.Adam([u], lr=lr, betas=(0.9, 0.999)) for n, u in self.params.items()}

    def step(self, ndx, fdx, base=None):
        for u in self.params.values():
            u.grad = None
        lat = self.sample_guide(self.params, ndx, fdx) if base is None else \
            self.latents_from_base(self.params, ndx, fdx, base)
        loss = -self.elbo(self.params, ndx, fdx, lat)
        loss.backward()
        for n, u in self.params.items():
            if u.grad is None:
                u.grad = torch.zeros_like(u)
            self.optim[n].step()
        return float(loss)

    # -- posteriors (cosmos.py:609-672) ----------------------------------------------
    @torch.no_grad()
    def compute_probs(self, ndx, fdx, lat_particles):
        """lat_particles: list of latent dicts (one per particle) for AOIs ndx, frames fdx.
        Returns z_probs (nb, fb, Q, 2), theta_probs (K, nb, fb, Q)."""
        K = self.K
        cp = self.constrained(self.params)
        p = cp["m_probs"][:, ndx[:, None], fdx[None, :], :]
        mg = self.m_grid()
        nb, fb = len(ndx), len(fdx)
        logq_m = sum(D.Bernoulli(probs=p[k]).log_prob(mg[:, k][:, None, None, None].expand(-1, nb, fb, self.Q))
                     for k in range(K))
        zs, ts = 0, 0
        for lat in lat_particles:
            _, joint = self.zt_marginal(lat, ndx, parts=True)  # (z, theta, M, nb, fb, C)
            logp = joint - torch.logsumexp(joint.flatten(0, 1), 0)
            res = torch.logsumexp(logp + logq_m, 2)  # (z, theta, nb, fb, C)
            zs = zs + torch.logsumexp(res, 1).exp()  # (z, nb, fb, C)
            ts = ts + torch.logsumexp(res, 0)[1:].exp()  # (K, nb, fb, C)
        n = len(lat_particles)
        return (zs / n).permute(1, 2, 3, 0), ts / n


# ----------------------------------------------------------------------------------------
# independent brute-force enumerator (python scalars; tiny inputs only)
# ----------------------------------------------------------------------------------------
def _lg(x):
    return math.lgamma(x)


def _gamma_lp(v, a, r):
    return a * math.log(r) + (a - 1) * math.log(v) - r * v - _lg(a)


def _beta_lp(t, c1, c0):
    return (c1 - 1) * math.log(t) + (c0 - 1) * math.log(1 - t) + _lg(c1 + c0) - _lg(c1) - _lg(c0)


def _abeta_lp(y, mean, size, low, high):
    sc = high - low
    return _beta_lp((y - low) / sc, size * (mean - low) / sc, size * (high - mean) / sc) - math.log(sc)


def _halfnormal_lp(v, s):
    return math.log(2) - 0.5 * math.log(2 * math.pi) - math.log(s) - v * v / (2 * s * s)


def elbo_bruteforce(oracle, params, ndx, fdx, lat):
    """Sum the published joint (cosmos.py:139-167) over every (z, theta) for every guide
    assignment m with explicit python loops and ``math`` scalars; no torch.distributions,
    no broadcasting.  Probabilities that are exactly zero are treated as zero
    (not eps-clamped) -- the difference is ~1e-16 relative."""
    from .ksmogn import ksmogn_log_prob_bruteforce

    d, K, H, pr = oracle.data, oracle.K, oracle.H, oracle.priors
    cp = {n: v.detach() for n, v in oracle.constrained(params).items()}
    lat = {n: v.detach() for n, v in lat.items()}
    nb, fb, C = len(ndx), len(fdx), d.C
    s_n = d.Nt / nb
    s = s_n * d.F / fb
    Hs = (d.P + 1) / math.sqrt(12)
    f = float

    gain, prox = f(lat["gain"]), f(lat["proximity"])
    al, be = f(cp["gain_loc"] * cp["gain_beta"]), f(cp["gain_beta"])
    tot = _halfnormal_lp(gain, pr["gain_std"]) - _gamma_lp(gain, al, be)
    tot += -pr["proximity_rate"] * prox + math.log(pr["proximity_rate"]) \
        - _abeta_lp(prox, f(cp["proximity_loc"]), f(cp["proximity_size"]), 0.0, Hs)
    for q in range(C):
        lam = f(lat["lamda"][q])
        tot += math.log(pr["lamda_rate"]) - pr["lamda_rate"] * lam \
            - _gamma_lp(lam, f(cp["lamda_loc"][q] * cp["lamda_beta"][q]), f(cp["lamda_beta"][q]))
        p0, p1 = f(lat["pi"][q, 0]), f(lat["pi"][q, 1])
        c0, c1 = f(cp["pi_mean"][q, 0] * cp["pi_size"][q, 0]), f(cp["pi_mean"][q, 1] * cp["pi_size"][q, 0])
        tot += (_lg(1.0) - 2 * _lg(0.5) - 0.5 * math.log(p0) - 0.5 * math.log(p1))
        tot -= (_lg(c0 + c1) - _lg(c0) - _lg(c1) + (c0 - 1) * math.log(p0) + (c1 - 1) * math.log(p1))

    size_spec = (H / prox) ** 2 - 1
    for a, n in enumerate(ndx.tolist()):
        if not bool(d.mask[n]):
            continue
        on = bool(d.is_ontarget[n])
        for c in range(C):
            lam = f(lat["lamda"][c])
            pmat = probs_m(torch.tensor(lam, dtype=torch.float64), K).tolist()  # (1+K, K)
            rho = f(lat["pi"][c, 1]) if on else 0.0
            bm, bs = f(cp["background_mean_loc"][n, 0, c]), f(cp["background_std_loc"][n, 0, c])
            tot += s_n * (_halfnormal_lp(bm, pr["background_mean_std"]) + _halfnormal_lp(bs, pr["background_std_std"]))
            for bi, fr in enumerate(fdx.tolist()):
                b = f(lat["background"][a, bi, c])
                E = _gamma_lp(b, (bm / bs) ** 2, bm / bs**2) \
                    - _gamma_lp(b, f(cp["b_loc"][n, fr, c] * cp["b_beta"][n, fr, c]), f(cp["b_beta"][n, fr, c]))
                hs = [f(lat["height"][k, a, bi, c]) for k in range(K)]
                ws = [f(lat["width"][k, a, bi, c]) for k in range(K)]
                xs = [f(lat["x"][k, a, bi, c]) for k in range(K)]
                ys = [f(lat["y"][k, a, bi, c]) for k in range(K)]
                qm = [f(cp["m_probs"][k, n, fr, c]) for k in range(K)]
                for m in itertools.product([0, 1], repeat=K):
                    Wm = 1.0
                    for k in range(K):
                        Wm *= qm[k] if m[k] else 1 - qm[k]
                    if Wm == 0.0:
                        continue
                    # sum over z, theta of the model factors that depend on them
                    acc = 0.0
                    for z in (0, 1):
                        pz = (rho if z == 1 else 1 - rho)
                        for th in range(K + 1):
                            pth = (1.0 if th == 0 else 0.0) if z == 0 else (0.0 if th == 0 else 1.0 / K)
                            term = pz * pth
                            if term == 0.0:
                                continue
                            for k in range(K):
                                pk = pmat[th][k]
                                term *= pk if m[k] else 1 - pk
                                if m[k]:
                                    sz = size_spec if th == k + 1 else 2.0
                                    term *= math.exp(_abeta_lp(xs[k], 0.0, sz, -H, H) + _abeta_lp(ys[k], 0.0, sz, -H, H))
                            acc += term
                    inner = math.log(acc)
                    inner += ksmogn_log_prob_bruteforce(
                        d.images[n, fr, c], _t(hs), _t(ws), _t(xs), _t(ys),
                        d.xy[n, fr, c], b, gain, d.offset_samples, d.offset_weights, d.P, m=torch.tensor(m))
                    for k in range(K):
                        if m[k]:
                            inner += _halfnormal_lp(hs[k], pr["height_std"]) - math.log(pr["width_max"] - pr["width_min"])
                            inner -= _gamma_lp(hs[k], f(cp["h_loc"][k, n, fr, c] * cp["h_beta"][k, n, fr, c]), f(cp["h_beta"][k, n, fr, c]))
                            inner -= _abeta_lp(ws[k], f(cp["w_mean"][k, n, fr, c]), f(cp["w_size"][k, n, fr, c]), pr["width_min"], pr["width_max"])
                            inner -= _abeta_lp(xs[k], f(cp["x_mean"][k, n, fr, c]), f(cp["size"][k, n, fr, c]), -H, H)
                            inner -= _abeta_lp(ys[k], f(cp["y_mean"][k, n, fr, c]), f(cp["size"][k, n, fr, c]), -H, H)
                        inner -= math.log(qm[k] if m[k] else 1 - qm[k])
                    E += Wm * inner
                tot += s * E
    return tot
