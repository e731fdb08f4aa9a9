// tq_pixel.h -- per-pixel arithmetic of the KSMOGN likelihood (host+device inline).
//
// Reference semantics: tapqir/distributions/ksmogn.py:187-238
//   log p(D | alpha, beta) = alpha ln beta - lgamma(alpha)
//                            + LSE_o [ ln w_o + (alpha-1) ln(D-delta_o) - beta (D-delta_o) ],  D > delta_o
// with alpha = mu / g, beta = 1 / g, mu = background + sum_k m_k spot_k.
//
// fp32 formulation.  Substituting Stirling/Binet for lgamma removes the O(alpha ln alpha)
// cancelling terms: with v = D - delta, rho = v / mu, phi(rho) = ln rho + 1 - rho <= 0,
//   ln w + alpha ln beta - lgamma(alpha) + (alpha-1) ln v - beta v
//     = ln w - ln v + alpha phi(rho) + (1/2) ln alpha - ln sqrt(2 pi) - S(alpha)
// and the derivatives needed by the ELBO gradient are
//   d/dalpha = E_o[ln rho_o] + 1/(2 alpha) - S'(alpha)           (= ln beta - digamma(alpha) + E_o[ln v_o])
//   d/dbeta  = alpha / beta - E_o[v_o]
// where E_o is the softmax over offsets of the bracketed terms.  Gain enters through both:
//   d/dg = -(1/g) * [ alpha (d/dalpha + 1) - E_o[v_o] / g ].
#pragma once
#include "tq_math.h"

// quantities that depend on the combination (alpha) but not on the offset
struct TqComboPix {
  float alpha, lnalpha, ralpha;  // alpha, ln alpha, 1/alpha
  float lmu, rmu;                // ln mu, 1/mu
};

TQ_HD void tq_combo_prepare(float mu, float rg, float g, float ln_g, TqComboPix* c) {
  c->rmu = TQ_FRCP(mu);
  c->lmu = TQ_FLOG(mu);
  c->alpha = mu * rg;
  c->lnalpha = c->lmu - ln_g;
  c->ralpha = g * c->rmu;
}

// Single-offset fast path (all offset samples identical after host-side merging).
//   v = D - delta (> 0), lv = ln v, lw = ln weight.
// Outputs: lp = log-density, da = d lp / d alpha, gq = alpha (da + 1) - v / g  (gain term, see above)
TQ_HD void tq_pix_one_offset(float v, float lv, float lw, float mu, float rg, float g, float ln_g, float* lp,
                             float* da, float* gq) {
  const float rmu = TQ_FRCP(mu);
  const float rho = v * rmu;
  const float lrho = TQ_FLOG(rho);
  const float phi = lrho + 1.0f - rho;
  const float alpha = mu * rg;
  const float ralpha = g * rmu;
  const float lnalpha = lv - ln_g - lrho;
  float S, dS;
  tq_binet(alpha, lnalpha, ralpha, &S, &dS);
  *lp = lw - lv + alpha * phi + 0.5f * lnalpha - TQ_LN_SQRT_2PI - S;
  const float d = lrho + 0.5f * ralpha - dS;
  *da = d;
  *gq = alpha * (phi + 0.5f * ralpha - dS);  // = alpha (d + 1 - rho)
}

// Online log-sum-exp accumulator over offsets for one combination.
struct TqLse {
  float m, s, sl, sv;  // running max, sum e, sum e * ln rho_o, sum e * v_o
};
TQ_HD void tq_lse_init(TqLse* a) {
  a->m = -INFINITY;
  a->s = 0.0f;
  a->sl = 0.0f;
  a->sv = 0.0f;
}
// one (offset, combination) term:  t = (ln w_o - ln v_o) + alpha * phi(rho_o)
TQ_HD void tq_lse_push(TqLse* a, const TqComboPix& c, float v, float lv, float lwl) {
  const float rho = v * c.rmu;
  const float lrho = lv - c.lmu;
  const float t = lwl + c.alpha * (lrho + 1.0f - rho);
  const float e = TQ_FEXP(-fabsf(t - a->m));  // exp(-inf) = 0 covers the first push
  if (t > a->m) {
    a->s = a->s * e + 1.0f;
    a->sl = a->sl * e + lrho;
    a->sv = a->sv * e + v;
    a->m = t;
  } else {
    a->s += e;
    a->sl += e * lrho;
    a->sv += e * v;
  }
}
TQ_HD void tq_lse_finish(const TqLse& a, const TqComboPix& c, float rg, float* lp, float* da, float* gq) {
  if (a.s == 0.0f) {  // every offset masked (D <= min offset): log 0
    *lp = -INFINITY;
    *da = 0.0f;
    *gq = 0.0f;
    return;
  }
  float S, dS;
  tq_binet(c.alpha, c.lnalpha, c.ralpha, &S, &dS);
  const float rs = TQ_FRCP(a.s);
  *lp = a.m + TQ_FLOG(a.s) + 0.5f * c.lnalpha - TQ_LN_SQRT_2PI - S;
  const float d = a.sl * rs + 0.5f * c.ralpha - dS;
  *da = d;
  *gq = c.alpha * (d + 1.0f) - a.sv * rs * rg;
}
