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
//   v = D - delta (> 0), lv = ln v.  The combination-independent part of the log-density,
//   ln w - ln v - ln sqrt(2 pi), is added once per pixel by the caller; the rest is split as
//     log p = (1/g) * [mu phi(v/mu)]  +  [ (1/2) ln alpha - S(alpha) ]
//   so that the caller accumulates sum mu*phi and applies 1/g once per unit.
//   Outputs: mphi = mu*phi, rest = (1/2) ln alpha - S(alpha), da = d log p / d alpha.
// The gain term needs no per-combination work here: with one offset,
//   alpha (da + 1) - v/g = (mu da + mu - v) / g, and the pixel sums of mu*da, mu and v are
//   linear in sums the kernel accumulates anyway (see tq_ksmogn.hip).
// FAST: the caller guarantees alpha >= TQ_FAST_ALPHA (alpha >= background / gain for every
// combination); S = 1/(12a), S' = -1/(12a^2) are then exact to 7e-7 / 2e-7 absolute per pixel.
#define TQ_FAST_ALPHA 16.0f
template <bool FAST>
TQ_HD void tq_pix_one_offset(float v, float lv_minus_lng, float mu, float rg, float g, float* mphi, float* rest,
                             float* da) {
  const float rmu = TQ_FRCP(mu);
  const float rho = v * rmu;
  const float lrho = TQ_FLOG(rho);
  *mphi = mu * ((lrho + 1.0f) - rho);
  const float ralpha = g * rmu;
  const float lnalpha = lv_minus_lng - lrho;
  if (FAST) {
    *rest = 0.5f * lnalpha - ralpha * (1.0f / 12.0f);
    *da = lrho + ralpha * (0.5f + ralpha * (1.0f / 12.0f));
  } else {
    float S, dS;
    tq_binet(mu * rg, lnalpha, ralpha, &S, &dS);
    *rest = 0.5f * lnalpha - S;
    *da = lrho + (0.5f * ralpha - dS);
  }
}

// The all-spots-absent combination has mu = background for every pixel of a unit, so its
// alpha-dependent terms are per-unit constants (single-offset path only).
struct TqCombo0 {
  float alpha, lnb, rb;  // b/g, ln b, 1/b
  float c_lp;            // (1/2) ln alpha - S(alpha)
  float c_da;            // 1/(2 alpha) - S'(alpha)
};
TQ_HD void tq_combo0_prepare(float b, float rg, float g, float ln_g, TqCombo0* c) {
  c->alpha = b * rg;
  c->lnb = TQ_FLOG(b);
  c->rb = TQ_FRCP(b);
  const float lna = c->lnb - ln_g, ra = g * c->rb;
  float S, dS;
  tq_binet(c->alpha, lna, ra, &S, &dS);
  c->c_lp = 0.5f * lna - S;
  c->c_da = 0.5f * ra - dS;
}
// (per pixel only phi is needed: sum_pix [alpha phi + c_lp] = alpha * sum phi + npix * c_lp)
TQ_HD void tq_pix_combo0(const TqCombo0& c, float v, float lv, float* phi, float* da) {
  const float lrho = lv - c.lnb;
  *phi = (lrho + 1.0f) - v * c.rb;
  *da = lrho + c.c_da;
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
template <bool FAST>
TQ_HD void tq_lse_finish(const TqLse& a, const TqComboPix& c, float rg, float* lp, float* da, float* gq) {
  if (a.s == 0.0f) {  // every offset masked (D <= min offset): log 0
    *lp = -INFINITY;
    *da = 0.0f;
    *gq = 0.0f;
    return;
  }
  float S, dS;
  if (FAST) tq_binet_fast(c.ralpha, &S, &dS);
  else tq_binet(c.alpha, c.lnalpha, c.ralpha, &S, &dS);
  const float rs = TQ_FRCP(a.s);
  *lp = a.m + TQ_FLOG(a.s) + 0.5f * c.lnalpha - TQ_LN_SQRT_2PI - S;
  const float d = a.sl * rs + 0.5f * c.ralpha - dS;
  *da = d;
  *gq = c.alpha * (d + 1.0f) - a.sv * rs * rg;
}
