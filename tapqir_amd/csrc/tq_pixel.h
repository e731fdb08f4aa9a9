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

// ---- general path: a histogram of offsets ---------------------------------------------------------
// For one pixel D and one combination (alpha = mu/g, beta = 1/g), with v_o = D - delta_o > 0:
//   log p = alpha ln beta - lgamma(alpha) + ln sum_o w_o v_o^(alpha-1) exp(-beta v_o).
// All exponents are taken relative to a reference v_s INSIDE the valid range [v_lo, v_hi] at the maximiser
// of the (concave, or decreasing when alpha < 1) function (alpha-1) ln v - beta v, v* = mu - g:
//   v_s = clamp(mu - g, v_lo, v_hi),   v_hi = D - min_o delta_o,   v_lo = smallest positive v_o
// so that the terms
//   t_o = exp2[ (alpha-1) (log2(v_o / v_hi) - log2(v_s / v_hi)) - (beta / ln 2) (v_hi - v_s) + db_o ],
//   db_o = log2(w_o / w_max) + (beta / ln 2) (delta_o - delta_min)        (a per-offset constant)
// = (w_o / w_max) f(v_o) / f(v_s) with f(v) = v^(alpha-1) exp(-beta v) neither overflow (f(v_o) <= f(v_s)) nor all
// underflow, and
//   log p = [ -ln v_s + alpha phi(v_s / mu) + (1/2) ln alpha - ln sqrt(2 pi) - S(alpha) ]     (Binet form, see above)
//           + ln w_max + ln sum_o t_o.
// (Until round 3 the constant -beta (v_hi - v_s) stood outside the sum: exp2(db_o) alone reaches 2^(beta (delta_max -
// delta_min) / ln 2), past the range of float32 once (delta_max - delta_min) / gain > 88 -- a 260-count histogram with the
// gain fitted down to 2.9: d/d gain = -inf for a few units, every parameter NaN two steps later.)
// Per (offset, combination) that is one fma, one add, one exp2 and three accumulations
// (sum t, sum t log2(v_o / v_hi), sum t (delta_o - delta_min)); the log2 of v_o is shared by the combinations.
//   d/dalpha = E_t[ln(v_o / mu)] + 1/(2 alpha) - S'(alpha),   E_t[v_o] = v_hi - E_t[delta_o - delta_min].
struct TqOffsetInfo {
  float dmin, dmax;      // smallest / largest offset sample
  float lw2max, lw2min;  // log2 of the largest / smallest weight
};
TQ_HD void tq_offset_info(const float* samples, const float* logits, int O, TqOffsetInfo* h) {
  float lo = samples[0], hi = samples[0], lw = logits[0], lwn = logits[0];
  for (int o = 1; o < O; ++o) {
    lo = fminf(lo, samples[o]);
    hi = fmaxf(hi, samples[o]);
    lw = fmaxf(lw, logits[o]);
    lwn = fminf(lwn, logits[o]);
  }
  h->dmin = lo;
  h->dmax = hi;
  h->lw2max = lw * TQ_LOG2E;
  h->lw2min = lwn * TQ_LOG2E;
}

// reference point of one combination: returns v_s, writes a = alpha - 1 and c = -a log2(v_s / v_hi)
TQ_HD float tq_mo_reference(float mu, float g, float rg, float vlo, float vhi, float rvhi, float* a, float* c) {
  const float vs = fminf(fmaxf(mu - g, vlo), vhi);
  *a = mu * rg - 1.0f;
  *c = -(*a) * TQ_FLOG2(vs * rvhi) - (rg * TQ_LOG2E) * (vhi - vs);
  return vs;
}

// S0 = sum t, S1 = sum t log2(v_o / v_hi), S2 = sum t (delta_o - delta_min)  ->  log p, d/dalpha, gain term
TQ_HD void tq_mo_finish(bool fast, float mu, float vs, float vhi, float S0, float S1, float S2, const TqOffsetInfo& h,
                        float g, float rg, float ln_g, float* lp, float* da, float* gq) {
  const float rmu = TQ_FRCP(mu);
  const float alpha = mu * rg, ralpha = g * rmu;
  const float lnalpha = TQ_FLOG(mu) - ln_g;
  float S, dS;
  if (fast) tq_binet_fast(ralpha, &S, &dS);  // alpha >= TQ_FAST_ALPHA (the caller's test is wave-uniform on the device)
  else tq_binet(alpha, lnalpha, ralpha, &S, &dS);
  const float rho = vs * rmu;
  const float lrho = TQ_FLOG(rho);
  const float rs = TQ_FRCP(S0);
  *lp = (alpha * (lrho + 1.0f - rho) - (lrho + TQ_FLOG(mu))) + 0.5f * lnalpha - TQ_LN_SQRT_2PI - S
        + h.lw2max * TQ_LN2 + TQ_LN2 * TQ_FLOG2(S0);
  const float d = TQ_LN2 * (S1 * rs + TQ_FLOG2(vhi * rmu)) + 0.5f * ralpha - dS;
  *da = d;
  *gq = alpha * (d + 1.0f) - (vhi - S2 * rs) * rg;
}

// One pixel, every combination (scalar form; the packed kernel in tq_ksmogn.hip repeats it on float2).
template <int M, bool BWD, bool FAST>
TQ_HD void tq_pix_multi_offset(float D, const float* mu, const float* samples, const float* logits, int O,
                               const TqOffsetInfo& h, float g, float rg, float ln_g, float* lp, float* da, float* gq) {
  const float vhi = D - h.dmin;
  if (!(vhi > 0.0f)) {  // every offset at or above the pixel (ksmogn.py:226): log 0, no gradient
    for (int mi = 0; mi < M; ++mi) {
      lp[mi] = -INFINITY;
      da[mi] = 0.0f;
      gq[mi] = 0.0f;
    }
    return;
  }
  float vlo = D - h.dmax;
  if (!(vlo > 0.0f)) {  // some offsets are masked: the smallest valid v
    vlo = vhi;
    for (int o = 0; o < O; ++o) {
      const float v = D - samples[o];
      if (v > 0.0f) vlo = fminf(vlo, v);
    }
  }
  const float rvhi = TQ_FRCP(vhi);
  const float beta2 = rg * TQ_LOG2E;
  float a[M], c[M], vs[M], S0[M], S1[M], S2[M];
  for (int mi = 0; mi < M; ++mi) {
    vs[mi] = tq_mo_reference(mu[mi], g, rg, vlo, vhi, rvhi, &a[mi], &c[mi]);
    S0[mi] = S1[mi] = S2[mi] = 0.0f;
  }
  for (int o = 0; o < O; ++o) {
    const float v = D - samples[o];
    if (v > 0.0f) {
      const float dl = TQ_FLOG2(v * rvhi);
      const float dd = samples[o] - h.dmin;
      const float db = (logits[o] * TQ_LOG2E - h.lw2max) + beta2 * dd;
      for (int mi = 0; mi < M; ++mi) {
        const float t = TQ_FEXP2((a[mi] * dl + c[mi]) + db);
        S0[mi] += t;
        if (BWD) {
          S1[mi] += t * dl;
          S2[mi] += t * dd;
        }
      }
    }
  }
  for (int mi = 0; mi < M; ++mi) tq_mo_finish(FAST, mu[mi], vs[mi], vhi, S0[mi], S1[mi], S2[mi], h, g, rg, ln_g, &lp[mi], &da[mi], &gq[mi]);
}

// The same with the per-offset constants taken from a table tab[4 o] = {delta_o, delta_o - delta_min, log2 w_o - log2 w_max +
// beta (delta_o - delta_min) log2 e, unused} (16 bytes per offset: one LDS read) (the expressions of the loop above, formed once per workgroup: the 16-lane kernel keeps
// the table in LDS -- as two scalar loads from global memory per offset and pixel, the offset loop of a minibatch step
// with a 50-value histogram spent most of its time waiting for them).
template <int M, bool BWD, bool FAST>
TQ_HD void tq_pix_multi_offset_tab(float D, const float* mu, const float* tab, int O, const TqOffsetInfo& h, float g, float rg,
                                   float ln_g, float* lp, float* da, float* gq) {
  const float vhi = D - h.dmin;
  if (!(vhi > 0.0f)) {  // every offset at or above the pixel (ksmogn.py:226): log 0, no gradient
    for (int mi = 0; mi < M; ++mi) {
      lp[mi] = -INFINITY;
      da[mi] = 0.0f;
      gq[mi] = 0.0f;
    }
    return;
  }
  float vlo = D - h.dmax;
  if (!(vlo > 0.0f)) {  // some offsets are masked: the smallest valid v
    vlo = vhi;
    for (int o = 0; o < O; ++o) {
      const float v = D - tab[4 * o];
      if (v > 0.0f) vlo = fminf(vlo, v);
    }
  }
  const float rvhi = TQ_FRCP(vhi);
  float a[M], c[M], vs[M], S0[M], S1[M], S2[M];
  for (int mi = 0; mi < M; ++mi) {
    vs[mi] = tq_mo_reference(mu[mi], g, rg, vlo, vhi, rvhi, &a[mi], &c[mi]);
    S0[mi] = S1[mi] = S2[mi] = 0.0f;
  }
  // branch-free: an offset at or above the pixel contributes t = 0 (its logarithm is taken of a clamped argument and
  // selected away), so that the loads and transcendentals of consecutive offsets overlap
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (M % 2 == 0) {
    // the combinations in pairs on packed (two results per lane and issue slot) instructions: per offset and pair one
    // v_pk_fma for the exponent, two exp2, and -- with the backward -- two v_pk_fma and one v_pk_add for the three sums
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a2[M / 2], c2[M / 2], T0[M / 2], T1[M / 2], T2[M / 2];
    for (int p = 0; p < M / 2; ++p) {
      a2[p] = (f2){a[2 * p], a[2 * p + 1]};
      c2[p] = (f2){c[2 * p], c[2 * p + 1]};
      T0[p] = T1[p] = T2[p] = (f2){0.0f, 0.0f};
    }
    if (__builtin_amdgcn_ballot_w64(!(D - h.dmax > 0.0f)) == 0) {
      // no pixel of this wave reaches down to the largest offset (every real camera image: the offsets are its dark level):
      // no masks
#pragma unroll 4
      for (int o = 0; o < O; ++o) {
        const float4 e4 = *reinterpret_cast<const float4*>(tab + 4 * o);
        const float dl = TQ_FLOG2((D - e4.x) * rvhi);
        const f2 dl2 = (f2){dl, dl}, dd2 = (f2){e4.y, e4.y}, db2 = (f2){e4.z, e4.z};
        for (int p = 0; p < M / 2; ++p) {
          const f2 ex = (a2[p] * dl2 + c2[p]) + db2;
          const f2 t = (f2){TQ_FEXP2(ex.x), TQ_FEXP2(ex.y)};
          T0[p] += t;
          if (BWD) {
            T1[p] += t * dl2;
            T2[p] += t * dd2;
          }
        }
      }
    } else
#pragma unroll 4
    for (int o = 0; o < O; ++o) {
      const float4 e4 = *reinterpret_cast<const float4*>(tab + 4 * o);
      const float v = D - e4.x;
      const bool on = v > 0.0f;
      const float dl = TQ_FLOG2((on ? v : vhi) * rvhi);
      const float dd = e4.y;
      const float dbe = on ? e4.z : -INFINITY;  // exp2(-inf) = 0: a masked offset drops out of all three sums (dl is finite)
      const f2 dl2 = (f2){dl, dl}, dd2 = (f2){dd, dd}, db2 = (f2){dbe, dbe};
      for (int p = 0; p < M / 2; ++p) {
        const f2 ex = (a2[p] * dl2 + c2[p]) + db2;
        const f2 t = (f2){TQ_FEXP2(ex.x), TQ_FEXP2(ex.y)};
        T0[p] += t;
        if (BWD) {
          T1[p] += t * dl2;
          T2[p] += t * dd2;
        }
      }
    }
    for (int p = 0; p < M / 2; ++p) {
      S0[2 * p] = T0[p].x; S0[2 * p + 1] = T0[p].y;
      S1[2 * p] = T1[p].x; S1[2 * p + 1] = T1[p].y;
      S2[2 * p] = T2[p].x; S2[2 * p + 1] = T2[p].y;
    }
  } else
#endif
#pragma unroll 4
  for (int o = 0; o < O; ++o) {
    const float v = D - tab[4 * o];
    const bool on = v > 0.0f;
    const float dl = TQ_FLOG2((on ? v : vhi) * rvhi);
    const float dd = tab[4 * o + 1];
    const float db = tab[4 * o + 2];
    for (int mi = 0; mi < M; ++mi) {
      const float e = TQ_FEXP2((a[mi] * dl + c[mi]) + db);
      const float t = on ? e : 0.0f;
      S0[mi] += t;
      if (BWD) {
        S1[mi] += t * dl;
        S2[mi] += t * dd;
      }
    }
  }
  for (int mi = 0; mi < M; ++mi) tq_mo_finish(FAST, mu[mi], vs[mi], vhi, S0[mi], S1[mi], S2[mi], h, g, rg, ln_g, &lp[mi], &da[mi], &gq[mi]);
}

// One pixel, every combination, ONE offset (no data statistics needed: used by the crosstalk kernel, whose
// combinations all carry spots in every channel).  Same Binet form as above with S0 = 1:
//   log p = [ln w - ln sqrt(2 pi) - ln v] + alpha phi(v / mu) + (1/2) ln alpha - S(alpha).
template <int M, bool BWD, bool FAST>
TQ_HD void tq_pix_single_offset(float D, const float* mu, float off, float lw, float g, float rg, float ln_g, float* lp,
                                float* da, float* gq) {
  const float v = D - off;
  if (!(v > 0.0f)) {
    for (int mi = 0; mi < M; ++mi) {
      lp[mi] = -INFINITY;
      da[mi] = 0.0f;
      gq[mi] = 0.0f;
    }
    return;
  }
  const float lnv = TQ_FLOG(v);
  const float c0 = lw - TQ_LN_SQRT_2PI - lnv;
  const float vg = v * rg;
  for (int mi = 0; mi < M; ++mi) {
    const float r = TQ_FRCP(mu[mi]);
    const float rho = v * r;
    const float lnrho = TQ_LN2 * TQ_FLOG2(rho);
    const float alpha = mu[mi] * rg, ralpha = g * r;
    const float lnalpha = lnv - lnrho - ln_g;
    float S, dS;
    if (FAST) tq_binet_fast(ralpha, &S, &dS);
    else tq_binet(alpha, lnalpha, ralpha, &S, &dS);
    lp[mi] = c0 + alpha * (lnrho + 1.0f - rho) + 0.5f * lnalpha - S;
    if (BWD) {
      const float d = lnrho + 0.5f * ralpha - dS;
      da[mi] = d;
      gq[mi] = alpha * (d + 1.0f) - vg;
    }
  }
}
