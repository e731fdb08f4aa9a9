"""
Flat storage of the cosmos variational parameters.

The reference keeps one tensor per ``pyro.param`` name in Pyro's global store
(tapqir/models/cosmos.py:471-598).  Here all *unconstrained* values live in ONE flat float32
buffer (and so do the gradient and the two Adam moments), so that the dense Adam update of
model.py:169-171 is a single streaming kernel, and every named parameter is a zero-copy view
of that buffer with exactly the reference's shape:

    [ local   (8K+2) x U ]  m_probs, h_loc, h_beta, w_mean, w_size, x_mean, y_mean, size : (K, Nt, F, Q)
                            b_loc, b_beta : (Nt, F, C)
    [ per-AOI 2 x Nt*C   ]  background_mean_loc, background_std_loc : (Nt, 1, C)
    [ global  4 + 5Q     ]  gain_loc, gain_beta, proximity_loc, proximity_size : ()
                            lamda_loc, lamda_beta : (Q,)   pi_mean : (Q, 2)   pi_size : (Q, 1)
(the C ABI states the same layout in include/tapqir_hip.h).  The crosstalk model
(tapqir/models/crosstalk.py:429-438) appends  alpha_mean : (Q, C)  and  alpha_size : (Q, 1)  to the global block.
"""

import math

import torch
from torch.distributions import constraints, transform_to

LOCAL_K_NAMES = ["m_probs", "h_loc", "h_beta", "w_mean", "w_size", "x_mean", "y_mean", "size"]


class ParamLayout:
    def __init__(self, Nt, F, C, K, P, eps, crosstalk=False):
        self.Nt, self.F, self.C, self.K, self.P, self.Q = Nt, F, C, K, P, C
        self.crosstalk = bool(crosstalk)
        self.U = Nt * F * C
        self.n_local = (8 * K + 2) * self.U
        self.n_aoi = 2 * Nt * C
        self.n_global = 4 + 5 * C + (C * C + C if crosstalk else 0)  # alpha_mean (Q, C), alpha_size (Q, 1)
        self.total = self.n_local + self.n_aoi + self.n_global
        self.eps = eps

    # name -> (offset, shape)
    def slots(self):
        K, U, Nt, F, C, Q = self.K, self.U, self.Nt, self.F, self.C, self.Q
        s = {}
        for j, name in enumerate(LOCAL_K_NAMES):
            s[name] = (j * K * U, (K, Nt, F, Q))
        s["b_loc"] = (8 * K * U, (Nt, F, C))
        s["b_beta"] = ((8 * K + 1) * U, (Nt, F, C))
        a = self.n_local
        s["background_mean_loc"] = (a, (Nt, 1, C))
        s["background_std_loc"] = (a + Nt * C, (Nt, 1, C))
        g = self.n_local + self.n_aoi
        s["gain_loc"] = (g + 0, ())
        s["gain_beta"] = (g + 1, ())
        s["proximity_loc"] = (g + 2, ())
        s["proximity_size"] = (g + 3, ())
        s["lamda_loc"] = (g + 4, (Q,))
        s["lamda_beta"] = (g + 4 + Q, (Q,))
        s["pi_mean"] = (g + 4 + 2 * Q, (Q, 2))
        s["pi_size"] = (g + 4 + 4 * Q, (Q, 1))
        if self.crosstalk:
            s["alpha_mean"] = (g + 4 + 5 * Q, (Q, C))
            s["alpha_size"] = (g + 4 + 5 * Q + Q * C, (Q, 1))
        return s

    def views(self, flat):
        """dict name -> view of ``flat`` with the reference's parameter shape."""
        out = {}
        for name, (off, shape) in self.slots().items():
            n = int(math.prod(shape)) if shape else 1
            out[name] = flat[off:off + n].view(shape)
        return out

    def constraints(self):
        """cosmos.py:471-598; eps = finfo(model dtype).eps."""
        P, e = self.P, self.eps
        H = (P + 1) / 2
        c = constraints
        cons = {
            "pi_mean": c.simplex, "pi_size": c.positive, "m_probs": c.unit_interval,
            "proximity_loc": c.interval(0.0, (P + 1) / math.sqrt(12) - e),
            "proximity_size": c.greater_than(2.0),
            "lamda_loc": c.positive, "lamda_beta": c.positive, "gain_loc": c.positive, "gain_beta": c.positive,
            "background_mean_loc": c.positive, "background_std_loc": c.positive,
            "b_loc": c.positive, "b_beta": c.positive, "h_loc": c.positive, "h_beta": c.positive,
            "w_mean": c.interval(0.75 + e, 2.25 - e), "w_size": c.greater_than(2.0),
            "x_mean": c.interval(-H + e, H - e), "y_mean": c.interval(-H + e, H - e),
            "size": c.greater_than(2.0),
        }
        if self.crosstalk:  # crosstalk.py:429-438
            cons["alpha_mean"] = c.simplex
            cons["alpha_size"] = c.positive
        return cons

    def constrained(self, flat, names=None):
        """dict name -> constrained value; ``names`` restricts it (the convergence check of every checkpoint needs
        the few global parameters, not the transform of every local one)."""
        cons = self.constraints()
        return {n: transform_to(cons[n])(v) for n, v in self.views(flat).items() if names is None or n in names}

    def set_constrained(self, flat, values):
        """Write constrained values (dict name -> tensor) as unconstrained leaves (pyro.param
        stores ``transform_to(constraint).inv(init)``)."""
        cons = self.constraints()
        views = self.views(flat)
        for n, v in values.items():
            u = transform_to(cons[n]).inv(torch.as_tensor(v, dtype=torch.float64).expand(views[n].shape))
            views[n].copy_(u.to(flat.dtype))
