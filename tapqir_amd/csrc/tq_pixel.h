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

// ---- single-offset path (all offset samples identical after host-side merging) -------------------
// With one offset delta, v = D - delta is a property of the DATA.  Write the per-pixel log-density as
//   log p = [ln w - ln sqrt(2pi) - ln v] + (1/g) [mu ln(v/mu) + mu - v] + (1/2) [ln v - ln g - ln(v/mu)] - S(alpha)
// Summed over the pixels of a unit, everything except  sum mu*ln(v/mu),  sum ln(v/mu),  sum S(alpha)  and
// sum mu  is a function of the per-unit data statistics
//   S_v = sum_pix v ,   S_lv = sum_pix ln v        (tq_image_stats, computed once per dataset)
// so the pixel loop evaluates, per combination with at least one spot,
//   rcp(mu), l2 = log2(v/mu), r = g/mu             -> running sums of mu*l2, l2, S
//   da = d log p / d alpha = ln2*l2 + r/2 - S'(alpha)
// and NOTHING for the spot-free combination (mu = background for every pixel):
//   sum mu*ln(v/mu) = b (S_lv - n ln b),  sum ln(v/mu) = S_lv - n ln b,  sum_pix da = S_lv - n ln b + n c_da.
// The gain term alpha (da + 1) - v/g = (mu da + mu - v)/g is linear in sums the kernel has anyway.
// FAST: the caller guarantees alpha >= TQ_FAST_ALPHA for every combination (alpha >= background /
// gain); two Binet terms, S = 1/(12a) - 1/(360a^3), S' = -1/(12a^2) + 1/(120a^4), are then exact to
// 2.4e-8 / 1.5e-8 absolute per pixel.  (8 rather than a larger cut: typical data has background/gain
// around 20 with +-30 % spread across draws, and one unit below the cut sends its whole wave to the
// general loop.)
#define TQ_FAST_ALPHA 8.0f
template <bool FAST>
TQ_HD void tq_pix_one_offset(float v, float mu, float g, float rg, float ln_g, float* l2, float* S, float* da) {
  const float rmu = TQ_FRCP(mu);
  const float ralpha = g * rmu;
#if defined(__HIP_DEVICE_COMPILE__)
  const float lg2 = __builtin_amdgcn_logf(v * rmu);
#else
  const float lg2 = log2f(v * rmu);
#endif
  *l2 = lg2;
  if (FAST) {
    const float r2 = ralpha * ralpha;
    *S = ralpha * (1.0f / 12.0f - r2 * (1.0f / 360.0f));
    *da = lg2 * TQ_LN2 + (0.5f * ralpha + r2 * (1.0f / 12.0f - r2 * (1.0f / 120.0f)));
  } else {
    float S_, dS_;
    tq_binet(mu * rg, TQ_FLOG(mu) - ln_g, ralpha, &S_, &dS_);
    *S = S_;
    *da = lg2 * TQ_LN2 + (0.5f * ralpha - dS_);
  }
}

// per-unit constants of the spot-free combination
struct TqCombo0 {
  float alpha, lnb;  // b/g, ln b
  float S;           // S(alpha)
  float c_da;        // 1/(2 alpha) - S'(alpha)
};
TQ_HD void tq_combo0_prepare(float b, float rg, float g, float ln_g, TqCombo0* c) {
  c->alpha = b * rg;
  c->lnb = TQ_FLOG(b);
  const float lna = c->lnb - ln_g, ra = g * TQ_FRCP(b);
  float S, dS;
  tq_binet(c->alpha, lna, ra, &S, &dS);
  c->S = S;
  c->c_da = 0.5f * ra - dS;
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
