"""
``KSMOGN`` -- K-Spots Marginalised-Offset Gamma-Noise image distribution (drop-in for
tapqir/distributions/ksmogn.py:21-238, cosmos branch), evaluated by the fused HIP kernel
``tq_ksmogn_log_prob`` (include/tapqir_hip.h) instead of KeOps / dense torch.

    mu[j, i] = background + sum_k m_k h_k N(i; x_k + tx, w_k) N(j; y_k + ty, w_k)
    p(D | mu, g) = sum_o w_o Gamma(D - delta_o; mu / g, 1 / g),  D > delta_o

``log_prob`` is differentiable w.r.t. height, width, x, y, background and gain (the kernel's fused
backward).  There is no CPU path: tensors must live on the HIP device.

With ``alpha`` (Q, C) the distribution is the crosstalk branch (ksmogn.py:93-105, 119-144, 161-165): parameters have
shape (..., Q, K), ``background`` (..., C), ``target_locs`` (..., C, 2), the event shape is (C, P, P) and
``image_c = background_c + sum_q alpha[q, c] sum_k m_qk h_qk N(.)``; evaluated by ``tq_ksmogn_crosstalk_log_prob``
(Q = C = 2, K <= 2), differentiable w.r.t. ``alpha`` as well.
"""

import ctypes as C

import torch
from torch.distributions import constraints
from torch.distributions.distribution import Distribution

from tapqir_amd import _lib
from tapqir_amd.distributions.util import gaussian_spots
from tapqir_amd.exceptions import HipExtensionError


def _launch(value, h, w, x, y, xy, b, gain, offs, logits, P, K, gout=None):
    """value (B,P,P); h,w,x,y (K,B); xy (B,2); b (B,) -> ll (2^K, B) [+ grads]."""
    if value.device.type != "cuda":
        raise HipExtensionError("KSMOGN.log_prob runs on the HIP device only (no CPU fallback)")
    lib = _lib.load()
    B, M = value.shape[0], 1 << K
    dev, f32 = value.device, torch.float32
    ll = torch.empty(M, B, dtype=f32, device=dev)
    a = _lib.KsmognArgs()
    p = _lib.ptr
    a.images, a.images_il, a.pixstats, a.xy = p(value), None, None, p(xy)
    a.background, a.height, a.width, a.x, a.y, a.gain = p(b), p(h), p(w), p(x), p(y), p(gain)
    a.offset_samples, a.offset_logits = p(offs), p(logits)
    a.ll = p(ll)
    grads = None
    if gout is not None:
        grads = {n: torch.empty(K, B, dtype=f32, device=dev) for n in ("h", "w", "x", "y")}
        grads["b"] = torch.empty(B, dtype=f32, device=dev)
        grads["g"] = torch.empty(B, dtype=f32, device=dev)
        a.gout = p(gout)
        a.g_background, a.g_gain = p(grads["b"]), p(grads["g"])
        a.g_height, a.g_width, a.g_x, a.g_y = p(grads["h"]), p(grads["w"]), p(grads["x"]), p(grads["y"])
    a.nb, a.fb, a.C, a.F, a.P, a.K, a.O = B, 1, 1, 1, P, K, offs.numel()
    a.nb_full, a.il_min_units, a.scale, a.m_kstride, a.stats_stride = B, 1 << 30, 1.0, B, B
    _lib.check(lib.tq_ksmogn_log_prob(C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
               "tq_ksmogn_log_prob")
    return ll, grads


class _KsmognLogProb(torch.autograd.Function):
    """The kernel evaluates every spot-presence combination of a unit; ``combo`` (B,) picks the one that the
    caller's 0/1 presence indicators select (bit k = m_k)."""

    @staticmethod
    def forward(ctx, value, h, w, x, y, xy, b, gain, offs, logits, combo, P):
        K = h.shape[0]
        ll, _ = _launch(value, h, w, x, y, xy, b, gain, offs, logits, P, K)
        ctx.save_for_backward(value, h, w, x, y, xy, b, gain, offs, logits, combo)
        ctx.P = P
        return ll.gather(0, combo[None])[0]

    @staticmethod
    def backward(ctx, go):
        value, h, w, x, y, xy, b, gain, offs, logits, combo = ctx.saved_tensors
        K, B = h.shape
        gout = torch.zeros(1 << K, B, dtype=torch.float32, device=value.device)
        gout.scatter_(0, combo[None], go[None].to(torch.float32))
        _, g = _launch(value, h, w, x, y, xy, b, gain, offs, logits, ctx.P, K, gout=gout)
        return (None, g["h"], g["w"], g["x"], g["y"], None, g["b"], g["g"].sum().reshape(gain.shape), None, None,
                None, None)


def _launch_xt(value, h, w, x, y, xy, b, gain, alpha, offs, logits, P, K, gout=None):
    """value (Bg,2,P,P); h,w,x,y (K, Bg*2) [unit g*2+q]; xy (Bg,2,2); b (Bg*2,) -> ll_joint (2^(2K), Bg) [+ grads]."""
    if value.device.type != "cuda":
        raise HipExtensionError("KSMOGN.log_prob runs on the HIP device only (no CPU fallback)")
    lib = _lib.load()
    Bg, B, MJ = value.shape[0], value.shape[0] * 2, 1 << (2 * K)
    dev, f32 = value.device, torch.float32
    ll = torch.empty(MJ, Bg, dtype=f32, device=dev)
    a = _lib.XtalkArgs()
    p = _lib.ptr
    a.images, a.xy = p(value), p(xy)
    a.background, a.height, a.width, a.x, a.y, a.gain, a.alpha = p(b), p(h), p(w), p(x), p(y), p(gain), p(alpha)
    a.offset_samples, a.offset_logits = p(offs), p(logits)
    a.ll_joint = p(ll)
    grads = None
    if gout is not None:
        grads = {n: torch.empty(K, B, dtype=f32, device=dev) for n in ("h", "w", "x", "y")}
        grads["b"] = torch.empty(B, dtype=f32, device=dev)
        grads["g"] = torch.empty(B, dtype=f32, device=dev)
        grads["alpha"] = torch.empty(2, B, dtype=f32, device=dev)
        a.gout = p(gout)
        a.g_background, a.g_gain, a.g_alpha = p(grads["b"]), p(grads["g"]), p(grads["alpha"])
        a.g_height, a.g_width, a.g_x, a.g_y = p(grads["h"]), p(grads["w"]), p(grads["x"]), p(grads["y"])
    a.nb, a.fb, a.C, a.F, a.P, a.K, a.O = Bg, 1, 2, 1, P, K, offs.numel()
    a.nb_full, a.il_min_units, a.scale, a.m_kstride = Bg, 1 << 30, 1.0, B
    _lib.check(lib.tq_ksmogn_crosstalk_log_prob(C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
               "tq_ksmogn_crosstalk_log_prob")
    return ll, grads


class _KsmognXtLogProb(torch.autograd.Function):
    """Crosstalk branch: the kernel evaluates the 2^(QK) joint combinations; ``combo`` (Bg,) selects one (bit q*K+k)."""

    @staticmethod
    def forward(ctx, value, h, w, x, y, xy, b, gain, alpha, offs, logits, combo, P):
        K = h.shape[0]
        ll, _ = _launch_xt(value, h, w, x, y, xy, b, gain, alpha, offs, logits, P, K)
        ctx.save_for_backward(value, h, w, x, y, xy, b, gain, alpha, offs, logits, combo)
        ctx.P = P
        return ll.gather(0, combo[None])[0]

    @staticmethod
    def backward(ctx, go):
        value, h, w, x, y, xy, b, gain, alpha, offs, logits, combo = ctx.saved_tensors
        K, Bg = h.shape[0], value.shape[0]
        gout = torch.zeros(1 << (2 * K), Bg, dtype=torch.float32, device=value.device)
        gout.scatter_(0, combo[None], go[None].to(torch.float32))
        _, g = _launch_xt(value, h, w, x, y, xy, b, gain, alpha, offs, logits, ctx.P, K, gout=gout)
        g_alpha = g["alpha"].reshape(2, Bg, 2).sum(1)  # [q][g*2 + c] -> (Q, C)
        return (None, g["h"], g["w"], g["x"], g["y"], None, g["b"], g["g"].sum().reshape(gain.shape), g_alpha, None,
                None, None, None)


class KSMOGN(Distribution):
    r"""Same constructor as tapqir.distributions.KSMOGN (ksmogn.py:70-86); ``use_pykeops`` is accepted and ignored."""

    arg_constraints = {}
    support = constraints.positive

    def __init__(self, height, width, x, y, target_locs, background, gain, offset_samples, offset_logits, P: int,
                 m=None, alpha=None, use_pykeops: bool = True, validate_args=None):
        self.height, self.width, self.x, self.y = height, width, x, y
        self.target_locs, self.m = target_locs, m
        self.background_ = background
        self.gain = gain
        self.alpha = alpha
        self.offset_samples, self.offset_logits, self.P = offset_samples, offset_logits, P
        batch_shape = torch.broadcast_shapes(height.shape, width.shape, x.shape, y.shape)
        if m is not None:
            batch_shape = torch.broadcast_shapes(batch_shape, m.shape)
        self.K = batch_shape[-1]
        if alpha is None:
            batch_shape = torch.broadcast_shapes(batch_shape[:-1], background.shape, target_locs.shape[:-1])
            event_shape = torch.Size([P, P])
        else:  # ksmogn.py:119-144: drop the K and Q dims; channels move into the event
            self.Q, self.C = alpha.shape[-2], alpha.shape[-1]
            if (self.Q, self.C) != (2, 2) or self.K > 2:
                raise NotImplementedError("the cross-talk branch is implemented for Q = C = 2 and K <= 2")
            batch_shape = torch.broadcast_shapes(batch_shape[:-2], background.shape[:-1], target_locs.shape[:-2])
            event_shape = torch.Size([self.C, P, P])
        super().__init__(batch_shape, event_shape, validate_args=False)

    # -- dense helpers (data generation only) -----------------------------------------------------------
    @property
    def image(self):
        if self.alpha is not None:  # ksmogn.py:93-105, 161-165
            h = self.height.unsqueeze(-2) * self.alpha[..., None]
            m = None if self.m is None else self.m.unsqueeze(-2)
            g = gaussian_spots(h, self.width.unsqueeze(-2), self.x.unsqueeze(-2), self.y.unsqueeze(-2),
                               self.target_locs.unsqueeze(-3).unsqueeze(-2), self.P, m)  # (..., Q, C, K, P, P)
            return self.background_[..., None, None] + g.sum(-5).sum(-3)
        g = gaussian_spots(self.height, self.width, self.x, self.y, self.target_locs.unsqueeze(-2), self.P, self.m)
        return self.background_[..., None, None] + g.sum(-3)

    def rsample(self, sample_shape=torch.Size(), seed=None):
        """ksmogn.py:171-185: per pixel ``Gamma(image / gain, 1 / gain) + offset_samples[odx]``,
        ``odx ~ Categorical(offset_logits)``; drawn by ``tq_ksmogn_rsample`` (Philox + Marsaglia-Tsang on the device:
        the law of torch's sampler, not its bit stream).  ``seed`` defaults to a draw from torch's global generator, so
        ``torch.manual_seed`` makes simulations reproducible."""
        dev = self.height.device
        if dev.type != "cuda":
            raise HipExtensionError("KSMOGN.rsample runs on the HIP device only (no CPU fallback)")
        f32 = torch.float32
        P, K = self.P, self.K
        shape = self._extended_shape(sample_shape)
        lead = shape[: len(shape) - len(self.event_shape)]  # sample_shape + batch_shape
        n = int(torch.Size(lead).numel())
        m = torch.ones((), device=dev) if self.m is None else self.m
        if self.alpha is None:
            # one render unit per batch element, K spots
            ex = lambda t: t.to(f32).expand(lead + (K,)).reshape(n, K).t().contiguous()
            h = ex(self.height * m)
            w, x, y = ex(self.width), ex(self.x), ex(self.y)
            xy = self.target_locs.to(f32).expand(lead + (2,)).reshape(n, 2).contiguous()
            b = self.background_.to(f32).expand(lead).reshape(n).contiguous()
            B, KK = n, K
        else:
            # crosstalk: one render unit per (batch element, channel c) with the Q K spots of every dye, heights times alpha_qc
            Q, Cc = self.Q, self.C
            hq = (self.height * m).to(f32).expand(lead + (Q, K))  # (..., Q, K)
            he = hq.unsqueeze(-3) * self.alpha.to(f32).transpose(-1, -2)[..., None]  # (..., C, Q, K)
            rep = lambda t: t.to(f32).expand(lead + (Q, K)).unsqueeze(-3).expand(lead + (Cc, Q, K))
            flat = lambda t: t.reshape(n * Cc, Q * K).t().contiguous()
            h, w, x, y = flat(he), flat(rep(self.width)), flat(rep(self.x)), flat(rep(self.y))
            xy = self.target_locs.to(f32).expand(lead + (Cc, 2)).reshape(n * Cc, 2).contiguous()
            b = self.background_.to(f32).expand(lead + (Cc,)).reshape(n * Cc).contiguous()
            B, KK = n * Cc, Q * K
        out = torch.empty(B, P, P, dtype=f32, device=dev)
        gain = self.gain.to(f32).reshape(-1)[:1].contiguous()
        offs, logits = self.offset_samples.to(f32).contiguous(), self.offset_logits.to(f32).contiguous()
        a = _lib.RsampleArgs()
        p = _lib.ptr
        a.height, a.width, a.x, a.y, a.xy, a.background, a.gain = p(h), p(w), p(x), p(y), p(xy), p(b), p(gain)
        a.offset_samples, a.offset_logits, a.out = p(offs), p(logits), p(out)
        a.B, a.P, a.K, a.O = B, P, KK, offs.numel()
        a.seed = int(torch.randint(0, 2**62, (1,)).item()) if seed is None else int(seed)
        _lib.check(_lib.load().tq_ksmogn_rsample(C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   "tq_ksmogn_rsample")
        return out.reshape(shape)

    sample = rsample

    # -- the fused kernel ---------------------------------------------------------------------------------
    def _log_prob_crosstalk(self, value):
        Q, Cc, K, P = self.Q, self.C, self.K, self.P
        bs = torch.broadcast_shapes(self.batch_shape, value.shape[:-3])
        f32 = torch.float32
        Bg = int(torch.Size(bs).numel())
        # (..., Q, K) -> [K][g*Q + q]
        ex = lambda t: t.to(f32).expand(bs + (Q, K)).reshape(Bg, Q, K).permute(2, 0, 1).reshape(K, Bg * Q).contiguous()
        if self.m is None:
            combo = torch.full((Bg,), (1 << (Q * K)) - 1, dtype=torch.int64, device=value.device)
        else:
            bits = (self.m.expand(bs + (Q, K)).reshape(Bg, Q * K) > 0).to(torch.int64)  # bit q*K + k
            combo = (bits << torch.arange(Q * K, device=value.device)[None, :]).sum(1)
        val = value.to(f32).expand(bs + (Cc, P, P)).reshape(Bg, Cc, P, P).contiguous()
        xy = self.target_locs.to(f32).expand(bs + (Cc, 2)).reshape(Bg, Cc, 2).contiguous()
        b = self.background_.to(f32).expand(bs + (Cc,)).reshape(Bg * Cc).contiguous()
        gain = self.gain.to(f32).reshape(-1)[:1].contiguous()
        out = _KsmognXtLogProb.apply(val, ex(self.height), ex(self.width), ex(self.x), ex(self.y), xy, b, gain,
                                     self.alpha.to(f32).contiguous(), self.offset_samples.to(f32).contiguous(),
                                     self.offset_logits.to(f32).contiguous(), combo, P)
        return out.reshape(bs)

    def log_prob(self, value):
        if self.alpha is not None:
            return self._log_prob_crosstalk(value)
        bs = torch.broadcast_shapes(self.batch_shape, value.shape[:-2])
        f32 = torch.float32
        ex = lambda t: t.to(f32).expand(bs + (self.K,)).reshape(-1, self.K).t().contiguous()
        B = int(torch.Size(bs).numel())
        if self.m is None:
            combo = torch.full((B,), (1 << self.K) - 1, dtype=torch.int64, device=value.device)
        else:  # presence indicators are 0/1 (Bernoulli draws or enumerated values, cosmos.py:262-267)
            bits = (ex(self.m) > 0).to(torch.int64)  # (K, B)
            combo = (bits << torch.arange(self.K, device=value.device)[:, None]).sum(0)
        val = value.to(f32).expand(bs + (self.P, self.P)).reshape(B, self.P, self.P).contiguous()
        xy = self.target_locs.to(f32).expand(bs + (2,)).reshape(B, 2).contiguous()
        b = self.background_.to(f32).expand(bs).reshape(B).contiguous()
        gain = self.gain.to(f32).reshape(-1)[:1].contiguous()
        out = _KsmognLogProb.apply(val, ex(self.height), ex(self.width), ex(self.x), ex(self.y), xy, b, gain,
                                   self.offset_samples.to(f32).contiguous(), self.offset_logits.to(f32).contiguous(),
                                   combo, self.P)
        return out.reshape(bs)


KSpotGammaNoise = KSMOGN  # the spelling used in BASELINE.json's north_star
