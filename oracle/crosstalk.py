"""
Oracle restatement of ``tapqir/models/crosstalk.py`` (TEST INFRASTRUCTURE ONLY; float64 dense torch).

The crosstalk model is the cosmos model with
  * a global Dirichlet site ``alpha`` (Q, C): fraction of dye q's signal seen in channel c
    (crosstalk.py:82-87 prior ``Dirichlet(1 + 9 I)``, 279-284 guide ``Dirichlet(alpha_mean * alpha_size)``,
    429-438 parameters),
  * ONE data site per AOI-frame with event shape (C, P, P) whose image in channel c adds the spots of every dye,
    ``image_c = b_c + sum_q alpha_qc sum_k m_qk h_qk N(.)`` (crosstalk.py:262-278, ksmogn.py:93-105),
  * plates aois (-2) / frames (-1) only; backgrounds are ``to_event(1)`` over channels (crosstalk.py:132-156),
  * per-dye enumerated z_q, theta_q (model) and m_kq (guide): 2^(K Q) joint spot-presence combinations.
The z/theta marginal, the spot-site terms and the background terms are the per-dye / per-channel ones of
cosmos (they factorise over q); only the likelihood couples the dyes.  PARITY UNPINNED like the cosmos
ELBO (no Pyro here); checked against ``elbo_bruteforce_crosstalk`` below.
"""

import itertools
import math

import torch
import torch.distributions as D
from torch.distributions import constraints

from .cosmos import (AffineBeta, CosmosOracle, _DirichletGiven, _abeta_lp, _gamma_lp, _halfnormal_lp, _lg, _t)
from .dist_util import probs_m
from .ksmogn import ksmogn_crosstalk_log_prob, ksmogn_log_prob_bruteforce


class CrosstalkOracle(CosmosOracle):
    def __init__(self, data, K=2, priors=None, eps=None):
        super().__init__(data, K=K, priors=priors, eps=eps)
        assert self.Q == data.C, "crosstalk.py indexes dyes and channels with the same default (Q = data.C)"

    def alpha_prior_conc(self):  # crosstalk.py:84-86
        return torch.ones(self.Q, self.data.C, dtype=torch.float64) + 9 * torch.eye(self.Q, dtype=torch.float64)

    def _constraints(self):
        c = super()._constraints()
        c["alpha_mean"] = constraints.simplex
        c["alpha_size"] = constraints.positive
        return c

    def init_values(self):  # crosstalk.py:424-455
        v = super().init_values()
        v["alpha_mean"] = self.alpha_prior_conc()
        v["alpha_size"] = torch.full((self.Q, 1), 2.0, dtype=torch.float64)
        return v

    def _guide_dists(self, cp, ndx, fdx):
        g = super()._guide_dists(cp, ndx, fdx)
        g["alpha"] = D.Dirichlet(cp["alpha_mean"] * cp["alpha_size"])
        return g

    def sample_guide(self, params, ndx, fdx):
        lat = super().sample_guide(params, ndx, fdx)
        lat["alpha"] = self._guide_dists(self.constrained(params), ndx, fdx)["alpha"].rsample()
        return lat

    @staticmethod
    def base_draws(lat, params_constrained_dists):
        b = CosmosOracle.base_draws(lat, params_constrained_dists)
        b["alpha_x"] = lat["alpha"].detach()
        return b

    def latents_from_base(self, params, ndx, fdx, base):
        lat = super().latents_from_base(params, ndx, fdx, base)
        g = self._guide_dists(self.constrained(params), ndx, fdx)
        lat["alpha"] = _DirichletGiven.apply(g["alpha"].concentration, base["alpha_x"])
        return lat

    def joint_grid(self):
        """(Mj = 2^(K Q), Q, K): joint combination index has bit (q K + k) = m_qk."""
        K, Q = self.K, self.Q
        return torch.tensor([[[(mj >> (q * K + k)) & 1 for k in range(K)] for q in range(Q)]
                             for mj in range(2 ** (K * Q))], dtype=torch.float64)

    def elbo(self, params, ndx, fdx, lat):
        d, K, Q, pr = self.data, self.K, self.Q, self.priors
        cp = self.constrained(params)
        g = self._guide_dists(cp, ndx, fdx)
        nb, fb = len(ndx), len(fdx)
        s_n = d.Nt / nb
        s = s_n * d.F / fb
        mask = d.mask[ndx].to(torch.float64)[:, None]
        n_, f_ = ndx[:, None], fdx[None, :]

        # global sites (crosstalk.py:80-103 / 268-305)
        G = D.HalfNormal(_t(pr["gain_std"])).log_prob(lat["gain"]) - g["gain"].log_prob(lat["gain"])
        G = G + (D.Dirichlet(self.alpha_prior_conc()).log_prob(lat["alpha"]) - g["alpha"].log_prob(lat["alpha"])).sum()
        G = G + (D.Dirichlet(torch.full((Q, 2), 0.5, dtype=torch.float64)).log_prob(lat["pi"])
                 - g["pi"].log_prob(lat["pi"])).sum()
        G = G + (D.Exponential(_t(pr["lamda_rate"])).log_prob(lat["lamda"]) - g["lamda"].log_prob(lat["lamda"])).sum()
        G = G + D.Exponential(_t(pr["proximity_rate"])).log_prob(lat["proximity"]) - g["proximity"].log_prob(lat["proximity"])

        bm = cp["background_mean_loc"][ndx]  # (nb, 1, C)
        bs = cp["background_std_loc"][ndx]
        A = (D.HalfNormal(_t(pr["background_mean_std"])).log_prob(bm)
             + D.HalfNormal(_t(pr["background_std_std"])).log_prob(bs)).sum(-1)  # to_event(1): (nb, 1)

        b = lat["background"]
        E = (D.Gamma((bm / bs) ** 2, bm / bs**2).log_prob(b) - g["background"].log_prob(b)).sum(-1)  # (nb, fb)

        mgj = self.joint_grid()  # (Mj, Q, K)
        p = g["m_probs"]  # (K, nb, fb, Q)
        logq = sum(D.Bernoulli(probs=p[k, :, :, q]).log_prob(mgj[:, q, k][:, None, None].expand(-1, nb, fb))
                   for q in range(Q) for k in range(K))  # (Mj, nb, fb)
        W = logq.exp()

        T = (D.HalfNormal(_t(pr["height_std"])).log_prob(lat["height"])
             + AffineBeta(torch.tensor(1.5, dtype=torch.float64), 2.0, pr["width_min"], pr["width_max"]).log_prob(lat["width"])
             - g["height"].log_prob(lat["height"]) - g["width"].log_prob(lat["width"])
             - g["x"].log_prob(lat["x"]) - g["y"].log_prob(lat["y"]))  # (K, nb, fb, Q)

        L = self.zt_marginal(lat, ndx)  # (2^K, nb, fb, Q): per-dye marginal over (z_q, theta_q)
        idx = (mgj * (2.0 ** torch.arange(K, dtype=torch.float64))).sum(-1).long()  # (Mj, Q): per-dye combination
        Lj = sum(L[idx[:, q], :, :, q] for q in range(Q))  # (Mj, nb, fb)
        Tj = sum(mgj[:, q, k][:, None, None] * T[k, :, :, q] for q in range(Q) for k in range(K))

        obs = d.images[n_, f_]  # (nb, fb, C, P, P)
        xy = d.xy[n_, f_]  # (nb, fb, C, 2)
        st = lambda v: v.permute(1, 2, 3, 0)  # (nb, fb, Q, K)
        ll = ksmogn_crosstalk_log_prob(obs, st(lat["height"]), st(lat["width"]), st(lat["x"]), st(lat["y"]), xy, b,
                                       lat["gain"], d.offset_samples, d.offset_logits, d.P,
                                       mgj[:, None, None], lat["alpha"])  # (Mj, nb, fb)
        inner = ll + Lj + Tj - logq
        E = E + (W * inner).sum(0)
        self.last_terms = {"G": G, "A": A, "E": E, "ll": ll, "L": L, "T": T, "W": W}
        return G + s_n * (mask * A).sum() + s * (mask * E).sum()


def elbo_bruteforce_crosstalk(oracle, params, ndx, fdx, lat):
    """Explicit python loops over every joint guide assignment m and every per-dye (z_q, theta_q)."""
    d, K, Q, H, pr = oracle.data, oracle.K, oracle.Q, oracle.H, oracle.priors
    cp = {n: v.detach() for n, v in oracle.constrained(params).items()}
    lat = {n: v.detach() for n, v in lat.items()}
    nb, fb, C = len(ndx), len(fdx), d.C
    s_n = d.Nt / nb
    s = s_n * d.F / fb
    Hs = (d.P + 1) / math.sqrt(12)
    f = float

    gain, prox = f(lat["gain"]), f(lat["proximity"])
    tot = _halfnormal_lp(gain, pr["gain_std"]) - _gamma_lp(gain, f(cp["gain_loc"] * cp["gain_beta"]), f(cp["gain_beta"]))
    tot += -pr["proximity_rate"] * prox + math.log(pr["proximity_rate"]) \
        - _abeta_lp(prox, f(cp["proximity_loc"]), f(cp["proximity_size"]), 0.0, Hs)
    conc = oracle.alpha_prior_conc()
    for q in range(Q):
        lam = f(lat["lamda"][q])
        tot += math.log(pr["lamda_rate"]) - pr["lamda_rate"] * lam \
            - _gamma_lp(lam, f(cp["lamda_loc"][q] * cp["lamda_beta"][q]), f(cp["lamda_beta"][q]))
        p0, p1 = f(lat["pi"][q, 0]), f(lat["pi"][q, 1])
        c0, c1 = f(cp["pi_mean"][q, 0] * cp["pi_size"][q, 0]), f(cp["pi_mean"][q, 1] * cp["pi_size"][q, 0])
        tot += (_lg(1.0) - 2 * _lg(0.5) - 0.5 * math.log(p0) - 0.5 * math.log(p1))
        tot -= (_lg(c0 + c1) - _lg(c0) - _lg(c1) + (c0 - 1) * math.log(p0) + (c1 - 1) * math.log(p1))
        # alpha row q: Dirichlet prior and guide over C components
        a_pr = [f(conc[q, c]) for c in range(C)]
        a_gu = [f(cp["alpha_mean"][q, c] * cp["alpha_size"][q, 0]) for c in range(C)]
        xs_ = [f(lat["alpha"][q, c]) for c in range(C)]
        tot += _lg(sum(a_pr)) - sum(_lg(v) for v in a_pr) + sum((a_pr[c] - 1) * math.log(xs_[c]) for c in range(C))
        tot -= _lg(sum(a_gu)) - sum(_lg(v) for v in a_gu) + sum((a_gu[c] - 1) * math.log(xs_[c]) for c in range(C))

    size_spec = (H / prox) ** 2 - 1
    for a, n in enumerate(ndx.tolist()):
        if not bool(d.mask[n]):
            continue
        on = bool(d.is_ontarget[n])
        for c in range(C):
            bm, bs = f(cp["background_mean_loc"][n, 0, c]), f(cp["background_std_loc"][n, 0, c])
            tot += s_n * (_halfnormal_lp(bm, pr["background_mean_std"]) + _halfnormal_lp(bs, pr["background_std_std"]))
        for bi, fr in enumerate(fdx.tolist()):
            E = 0.0
            bvals = []
            for c in range(C):
                bm, bs = f(cp["background_mean_loc"][n, 0, c]), f(cp["background_std_loc"][n, 0, c])
                b = f(lat["background"][a, bi, c])
                bvals.append(b)
                E += _gamma_lp(b, (bm / bs) ** 2, bm / bs**2) \
                    - _gamma_lp(b, f(cp["b_loc"][n, fr, c] * cp["b_beta"][n, fr, c]), f(cp["b_beta"][n, fr, c]))
            hs = [[f(lat["height"][k, a, bi, q]) for k in range(K)] for q in range(Q)]
            ws = [[f(lat["width"][k, a, bi, q]) for k in range(K)] for q in range(Q)]
            xs = [[f(lat["x"][k, a, bi, q]) for k in range(K)] for q in range(Q)]
            ys = [[f(lat["y"][k, a, bi, q]) for k in range(K)] for q in range(Q)]
            qm = [[f(cp["m_probs"][k, n, fr, q]) for k in range(K)] for q in range(Q)]
            for mflat in itertools.product([0, 1], repeat=K * Q):
                m = [[mflat[q * K + k] for k in range(K)] for q in range(Q)]
                Wm = 1.0
                for q in range(Q):
                    for k in range(K):
                        Wm *= qm[q][k] if m[q][k] else 1 - qm[q][k]
                if Wm == 0.0:
                    continue
                inner = 0.0
                for q in range(Q):
                    lam = f(lat["lamda"][q])
                    pmat = probs_m(torch.tensor(lam, dtype=torch.float64), K).tolist()
                    rho = f(lat["pi"][q, 1]) if on else 0.0
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
                                term *= pk if m[q][k] else 1 - pk
                                if m[q][k]:
                                    sz = size_spec if th == k + 1 else 2.0
                                    term *= math.exp(_abeta_lp(xs[q][k], 0.0, sz, -H, H) + _abeta_lp(ys[q][k], 0.0, sz, -H, H))
                            acc += term
                    inner += math.log(acc)
                    for k in range(K):
                        if m[q][k]:
                            inner += _halfnormal_lp(hs[q][k], pr["height_std"]) - math.log(pr["width_max"] - pr["width_min"])
                            inner -= _gamma_lp(hs[q][k], f(cp["h_loc"][k, n, fr, q] * cp["h_beta"][k, n, fr, q]), f(cp["h_beta"][k, n, fr, q]))
                            inner -= _abeta_lp(ws[q][k], f(cp["w_mean"][k, n, fr, q]), f(cp["w_size"][k, n, fr, q]), pr["width_min"], pr["width_max"])
                            inner -= _abeta_lp(xs[q][k], f(cp["x_mean"][k, n, fr, q]), f(cp["size"][k, n, fr, q]), -H, H)
                            inner -= _abeta_lp(ys[q][k], f(cp["y_mean"][k, n, fr, q]), f(cp["size"][k, n, fr, q]), -H, H)
                        inner -= math.log(qm[q][k] if m[q][k] else 1 - qm[q][k])
                # data: channel c sees every dye's spots scaled by alpha_qc; one K*Q-spot image per channel
                for c in range(C):
                    hh = [f(lat["alpha"][q, c]) * hs[q][k] for q in range(Q) for k in range(K)]
                    ww = [ws[q][k] for q in range(Q) for k in range(K)]
                    xx = [xs[q][k] for q in range(Q) for k in range(K)]
                    yy = [ys[q][k] for q in range(Q) for k in range(K)]
                    inner += ksmogn_log_prob_bruteforce(
                        d.images[n, fr, c], _t(hh), _t(ww), _t(xx), _t(yy), d.xy[n, fr, c], bvals[c], gain,
                        d.offset_samples, d.offset_weights, d.P, m=torch.tensor(list(mflat)))
                E += Wm * inner
            tot += s * E
    return tot
