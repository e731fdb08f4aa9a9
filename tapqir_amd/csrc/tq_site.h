// tq_site.h -- everything in one SVI step of the cosmos model that is per unit (AOI n, frame f,
// channel c) but NOT per pixel (host+device inline; used by tq_cosmos.hip and tests/hostcheck).
//
// Reference semantics (tapqir/models/cosmos.py:82-462 under pyro TraceEnum_ELBO; SURVEY.md
// Appendix A.3).  For one unit, with Dice weights W(m) = prod_k q(m_k), m in {0,1}^K:
//
//   E_u = log Gamma(b; (mu_b/sigma_b)^2, mu_b/sigma_b^2) - log q(b)
//       + sum_m W(m) { ll(m) + L(m) + sum_k m_k T_k - sum_k log q(m_k) }
//   T_k = log HalfNormal(h_k; height_std) + log U(w_k; w_min, w_max)
//         - log q(h_k) - log q(w_k) - log q(x_k) - log q(y_k)
//   L(m) = log[ (1-rho) A_0(m) + (rho/K) sum_k' A_k'(m) ]            (z, theta summed out)
//   A_theta(m) = prod_k Bern(m_k; pm[theta,k]) [p(x_k|theta) p(y_k|theta)]^{m_k}
//
// and the gradient of w_u * E_u (w_u = plate scale * AOI mask) with respect to the unit's
// 8K+2 unconstrained variational parameters, the AOI's background_mean/std parameters and the
// global latents (rho, a(lamda), c(lamda), c_s(proximity)); the pixel part (ll(m) and its
// pathwise gradients) comes from tq_ksmogn.hip.
//
// Reparameterisation: Gamma(alpha=loc*beta, beta) draws are g/beta with g ~ Gamma(alpha,1);
// AffineBeta draws are low + scale * t, t ~ Beta(c1,c0), clamped to [low+eps*scale,
// high-eps*scale] (pyro AffineBeta.rsample).  d g/d alpha and d t/d c are the implicit
// gradients of tq_math.h, exactly the functions torch.autograd uses for the reference.
#pragma once
#include "tq_math.h"

#define TQ_MAXK 4
#define TQ_MAXQ 4

// number of per-unit unconstrained local parameters: 8 per spot + 2
#define TQ_NLOCAL(K) (8 * (K) + 2)
// row order of the local parameter block [TQ_NLOCAL][U]  (spot rows are k-major within a name)
enum { TQ_P_MPROBS = 0, TQ_P_HLOC = 1, TQ_P_HBETA = 2, TQ_P_WMEAN = 3, TQ_P_WSIZE = 4, TQ_P_XMEAN = 5, TQ_P_YMEAN = 6, TQ_P_SIZE = 7 };
#define TQ_ROW(name, k, K) ((name) * (K) + (k))
#define TQ_ROW_BLOC(K) (8 * (K))
#define TQ_ROW_BBETA(K) (8 * (K) + 1)

// per-step global latents and tables derived from them (written by the globals kernel)
struct TqGlobals {
  float gain, proximity;
  float cs;       // per-axis concentration of the target-specific position prior: ((H/sigma)^2 - 1)/2
  // The symmetric Beta(cs, cs) density in t is written against 4 t (1-t) (<= 1, = 1 at the centre):
  //   ln Beta(t; cs, cs) = (cs-1) ln[4 t (1-t)] - lnBp,   lnBp = ln B(cs,cs) + 2 (cs-1) ln 2  (= O(ln cs))
  // so the O(cs) terms (cs-1) ln[t(1-t)] and ln B(cs,cs), which cancel, are never formed in fp32.
  float lnB_s;    // lnBp
  float dlnB_s;   // d lnBp / d cs = 2 psi(cs) - 2 psi(2 cs) + 2 ln 2  (= O(1/cs))
  float lamda[TQ_MAXQ];
  float rho[TQ_MAXQ];       // pi_q[1]
  float a[TQ_MAXQ];         // probs_m[q, theta=0, k]
  float c[TQ_MAXQ];         // probs_m[q, theta=k'+1, k != k']   (K >= 2)
  float alpha[2][2];        // crosstalk model only: alpha[q][c], fraction of dye q's signal in channel c
};

struct TqSiteConsts {
  float H;            // (P+1)/2
  float eps;          // finfo(model dtype).eps used in the interval bounds and the rsample clamp
  float w_lo, w_hi;   // priors width_min / width_max
  float height_std;
  float bg_mean_std, bg_std_std;  // only used by the AOI kernel
};

// ln(1 + u) with the rounding of 1 + u compensated (|u| < 1/2): one hardware log, one reciprocal
TQ_HD float tq_log1p_small(float u) {
  const float w = 1.0f + u;
  return TQ_FLOG(w) - ((w - 1.0f) - u) * TQ_FRCP(w);
}

// ---- log-densities with derivatives ---------------------------------------------------------------
// Gamma(v; alpha = loc*beta, rate = beta) in the cancellation-free form (see tq_pixel.h):
//   = -ln v + alpha phi(v/loc) + (1/2) ln alpha - ln sqrt(2pi) - S(alpha)
TQ_HD void tq_gamma_logpdf(float v, float loc, float beta, float* lp, float* d_v, float* d_alpha, float* d_beta) {
  const float alpha = loc * beta;
  const float rloc = TQ_FRCP(loc);
  const float rho = v * rloc;
  const float lnv = TQ_FLOG(v);
  // ln(v / loc): log1p near the mean (no cancellation in phi), difference of logs far from it (a draw many orders of
  // magnitude below loc, as Gamma draws with concentration < 1 are, must not round (v - loc) / loc to -1)
  const float lrho = (fabsf(v - loc) < 0.5f * loc) ? tq_log1p_small((v - loc) * rloc) : lnv - TQ_FLOG(loc);
  const float lna = TQ_FLOG(alpha), ra = TQ_FRCP(alpha);
  float S, dS;
  tq_binet(alpha, lna, ra, &S, &dS);
  *lp = -lnv + alpha * (lrho + 1.0f - rho) + 0.5f * lna - TQ_LN_SQRT_2PI - S;
  *d_v = (alpha - 1.0f) * TQ_FRCP(v) - beta;
  *d_alpha = lrho + 0.5f * ra - dS;  // = ln beta + ln v - digamma(alpha)
  *d_beta = loc - v;                 // = alpha / beta - v
}

// Beta(t; c1, c0)
TQ_HD void tq_beta_logpdf(float t, float c1, float c0, float* lp, float* d_t, float* d_c1, float* d_c0) {
  const float T = c1 + c0;
  *d_t = (c1 - 1.0f) * TQ_FRCP(t) - (c0 - 1.0f) * TQ_FRCP(1.0f - t);
  if (c1 >= 8.0f && c0 >= 8.0f) {
    // Binet form throughout: lgamma(a) = (a - 1/2) ln a - a + ln sqrt(2 pi) + S(a), psi(a) = ln a - 1/(2a) + S'(a).
    // d lp / d c1 = ln t - psi(c1) + psi(T) is O(|t - mean| / mean + 1/c) while its three terms are O(ln c): the
    // logarithms combine to ln(t T / c1) = log1p((t T - c1) / c1), and likewise for c0 with (1-t) T - c0 = -(t T - c1)
    const float r1 = TQ_FRCP(c1), r0 = TQ_FRCP(c0), rT = TQ_FRCP(T);
    float S1, dS1, S0, dS0, ST, dST;
    tq_binet_series<float>(c1, r1, &S1, &dS1);
    tq_binet_series<float>(c0, r0, &S0, &dS0);
    tq_binet_series<float>(T, rT, &ST, &dST);
    const float num = t * T - c1;
    const float u1 = tq_log1p_small(num * r1), u0 = tq_log1p_small(-num * r0);  // ln(t T / c1), ln((1-t) T / c0)
    // (c1-1) ln t + (c0-1) ln(1-t) + lgamma(T) - lgamma(c1) - lgamma(c0) with ln t = u1 + ln(c1/T) etc.: the O(c ln c)
    // terms cancel analytically, leaving (c1-1) u1 + (c0-1) u0 + (3/2) ln T - (1/2)(ln c1 + ln c0) - ln sqrt(2 pi) + S terms
    *lp = ((c1 - 1.0f) * u1 + (c0 - 1.0f) * u0) + (1.5f * TQ_FLOG(T) - 0.5f * (TQ_FLOG(c1) + TQ_FLOG(c0))) - TQ_LN_SQRT_2PI +
          (ST - S1 - S0);
    *d_c1 = u1 + 0.5f * (r1 - rT) + (dST - dS1);
    *d_c0 = u0 + 0.5f * (r0 - rT) + (dST - dS0);
  } else {
    const float lt = TQ_FLOG(t), l1t = TQ_FLOG(1.0f - t);
    float lg1, dg1, lg0, dg0, lgt, dgt;
    tq_lgamma_digamma(c1, &lg1, &dg1);
    tq_lgamma_digamma(c0, &lg0, &dg0);
    tq_lgamma_digamma(T, &lgt, &dgt);
    *lp = (c1 - 1.0f) * lt + (c0 - 1.0f) * l1t + lgt - lg1 - lg0;
    *d_c1 = lt - dg1 + dgt;
    *d_c0 = l1t - dg0 + dgt;
  }
}

// ---- guide-site terms ----------------------------------------------------------------------------
// Evaluated once per (unit, site) by the sampling kernel right after the draw (everything here
// depends on the site's own parameters and draw only):
//   Gamma(alpha = loc*beta, beta), draw v = g/beta:
//     s[0] = log q(v)  s[1] = d lq/d v  s[2] = d lq/d alpha  s[3] = d lq/d beta  s[4] = d v/d alpha (implicit)
//   AffineBeta(mean, size, low, high), draw y = low + sc*t (clamped):
//     s[0] = log q(y)  s[1] = d lq/d y  s[2] = d lq/d c1  s[3] = d lq/d c0  s[4] = d y/d c1  s[5] = d y/d c0
#define TQ_NSITE_TERMS 6
// What the sampling kernel hands to the per-unit phase through the step workspace (`site`, TQ_NSITE_STORED rows): the terms
// that cost transcendental series or regime code -- log q, d lq/d alpha (d c1, d c0), the implicit gradients.  d lq/d v,
// d lq/d beta (Gamma) and d lq/d y (AffineBeta) are one or two reciprocals from the draw and the parameters the per-unit
// phase holds anyway, and are rebuilt there (48 B per unit less written by one launch and read by the next at K = 2).
//   stored rows: 0 = s[0], 1 = s[2], 2 = s[4], and for AffineBeta sites 3 = s[3], 4 = s[5]
#define TQ_NSITE_STORED 5
TQ_HD void tq_gamma_terms_expand(const float* c, float v, float loc, float beta, float* s) {
  s[0] = c[0];
  s[1] = (loc * beta - 1.0f) * TQ_FRCP(v) - beta;  // d lq / d v   (tq_gamma_logpdf)
  s[2] = c[1];
  s[3] = loc - v;                                  // d lq / d beta
  s[4] = c[2];
  s[5] = 0.0f;
}
TQ_HD void tq_affine_beta_terms_expand(const float* c, float y, float mean, float size, float low, float high, float* s) {
  const float rsc = TQ_FRCP(high - low);
  const float t = (y - low) * rsc;
  const float c1 = size * (mean - low) * rsc;
  const float c0 = size * (high - mean) * rsc;
  s[0] = c[0];
  s[1] = ((c1 - 1.0f) * TQ_FRCP(t) - (c0 - 1.0f) * TQ_FRCP(1.0f - t)) * rsc;  // d lq / d y   (tq_beta_logpdf, tq_affine_beta_site_terms)
  s[2] = c[1];
  s[3] = c[3];
  s[4] = c[2];
  s[5] = c[4];
}

TQ_HD void tq_gamma_site_terms(float v, float loc, float beta, float* s) {
  float lq, d_v, d_alpha, d_beta;
  tq_gamma_logpdf(v, loc, beta, &lq, &d_v, &d_alpha, &d_beta);
  s[0] = lq;
  s[1] = d_v;
  s[2] = d_alpha;
  s[3] = d_beta;
#ifdef TQ_DIAG_NO_GAMMAGRAD  // (diagnostic builds: scripts/gpu_site_diag.sh)
  s[4] = v;
#else
  s[4] = tq_std_gamma_grad(loc * beta, v * beta) * TQ_FRCP(beta);
#endif
  s[5] = 0.0f;
}

// dd_given: the two tq_dirichlet_grad values of the draw, dd[0] = grad(t, c1, size), dd[1] = grad(1 - t, c0, size), if the
// caller has evaluated them already (the sampling kernels' regime compaction); nullptr: evaluated here
TQ_HD void tq_affine_beta_site_terms(float y, float mean, float size, float low, float high, float eps, float* s,
                                     const float* dd_given = nullptr) {
  const float sc = high - low;
  const float rsc = TQ_FRCP(sc);
  const float t = (y - low) * rsc;
  const float c1 = size * (mean - low) * rsc;
  const float c0 = size * (high - mean) * rsc;
  float lq, d_t, d_c1, d_c0;
#ifdef TQ_DIAG_NO_BETALP
  lq = t; d_t = c1; d_c1 = c0; d_c0 = t;
#else
  tq_beta_logpdf(t, c1, c0, &lq, &d_t, &d_c1, &d_c0);
#endif
  s[0] = lq - TQ_FLOG(sc);
  s[1] = d_t * rsc;
  s[2] = d_c1;
  s[3] = d_c0;
  // pathwise: y = low + sc * t unless clamped by rsample
  const bool clamped = (y <= low + eps * sc) || (y >= high - eps * sc);
  float dd[2] = {0.0f, 0.0f};
  if (dd_given) {
    dd[0] = dd_given[0];
    dd[1] = dd_given[1];
  } else
#ifdef TQ_DIAG_NO_BETAGRAD
  if (false) {
#else
  if (!clamped) {
#endif
    double ga, gb;
    if (tq_beta_grad_pair_mid((double)t, (double)c1, (double)size - (double)c1, &ga, &gb)) {
      dd[0] = (float)ga;  // common case: both directions in the saddle-point regime, evaluated together
      dd[1] = (float)gb;
    } else {
      tq_beta_grad_pair_rest(t, c1, c0, size, dd);  // the other regimes, each run once for both directions
    }
  }
  s[4] = sc * dd[0] * (1.0f - t);
  s[5] = -sc * dd[1] * t;
}

// chain rule of one Gamma site: e_v = d objective / d v from everything else, wq = weight of -log q
TQ_HD void tq_gamma_site_chain(const float* s, float v, float loc, float beta, float e_v, float wq, float* d_loc_u,
                               float* d_beta_u) {
  const float alpha = loc * beta;
  // d lq/d v = (alpha - 1)/v - beta reaches 3e37 when a draw of a Gamma with alpha < 1 lands on the clamp at the smallest
  // normal number (height of an absent spot in a converged fit: seen once in ~7000 minibatch steps); times the plate
  // scale Nt F / (nb fb) ~ 80 it overflows, and inf * (dv/dalpha ~ 1e-36) - inf made the height parameters NaN.  The
  // products are formed so that the large and the small factor meet first: s[1] s[4] = O(1), s[1] v = (alpha - 1) - beta v.
  const float g_alpha = (e_v * s[4] - wq * (s[1] * s[4])) - wq * s[2];
  const float s1v = (alpha - 1.0f) - beta * v;
  const float g_beta_direct = (wq * s1v - e_v * v) * TQ_FRCP(beta) - wq * s[3];
  *d_loc_u = g_alpha * alpha;                          // alpha = loc*beta; loc = exp(u_loc)
  *d_beta_u = g_alpha * alpha + g_beta_direct * beta;  // beta = exp(u_beta)
}

// chain rule of one AffineBeta site -> d objective / d mean, d size (constrained parameters)
TQ_HD void tq_affine_beta_site_chain(const float* s, float mean, float size, float low, float high, float e_y,
                                     float wq, float* d_mean, float* d_size) {
  const float rsc = TQ_FRCP(high - low);
  const float ey_tot = e_y - wq * s[1];
  const float g_c1 = ey_tot * s[4] - wq * s[2];
  const float g_c0 = ey_tot * s[5] - wq * s[3];
  *d_mean = (g_c1 - g_c0) * size * rsc;
  *d_size = (g_c1 * (mean - low) + g_c0 * (high - mean)) * rsc;
}

// ---- inputs / outputs of the per-unit routine ---------------------------------------------------
template <int K>
struct TqUnitIn {
  float u[TQ_NLOCAL(K)];   // unconstrained local parameters, row order above
  float u_bml, u_bsl;      // unconstrained background_mean_loc / background_std_loc of the AOI
  float b, h[K], w[K], x[K], y[K];  // latent draws
  float ll[1 << K];        // unweighted pixel log-likelihood per combination
  float gb, gh[K], gw[K], gx[K], gy[K];  // pathwise pixel gradients, already times w_u W(m)
  float sb[3];                              // stored guide-site terms of b (TQ_NSITE_STORED rows; Gamma sites: three)
  float sh[K][3], sw[K][TQ_NSITE_STORED], sx[K][TQ_NSITE_STORED], sy[K][TQ_NSITE_STORED];
  float wu;                // plate scale * mask
  int on;                  // is_ontarget
  int q;                   // dye / channel index
};

template <int K>
struct TqUnitOut {
  float g[TQ_NLOCAL(K)];   // d (w_u E_u) / d unconstrained local parameters
  float g_bml, g_bsl;      // contribution to d/d u_background_mean_loc, d/d u_background_std_loc
  float d_rho, d_a, d_c, d_cs;  // d (w_u E_u) / d global tables
  float elbo;              // w_u E_u
};

template <int K>
TQ_HD void tq_cosmos_unit(const TqUnitIn<K>& in, const TqGlobals& G, const TqSiteConsts& C, TqUnitOut<K>* out) {
  constexpr int M = 1 << K;
  const float wu = in.wu;
  const float H = C.H, eps = C.eps;
  const int q = in.q;

  // ---- q(m_k) ------------------------------------------------------------------------------------
  float p1[K], p0[K], lp1[K], lp0[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float uk = in.u[TQ_ROW(TQ_P_MPROBS, k, K)];
    p1[k] = tq_sigmoid(uk);
    p0[k] = tq_sigmoid(-uk);
    lp1[k] = -tq_softplus(-uk);
    lp0[k] = -tq_softplus(uk);
  }

  // ---- z / theta marginal L(m) and its derivatives -----------------------------------------------
  const float rho = in.on ? G.rho[q] : 0.0f;
  const float a = G.a[q], c = G.c[q];
  const float ln_a = TQ_FLOG(a), ln_1ma = TQ_FLOG(1.0f - a);
  const float ln_c = (K > 1) ? TQ_FLOG(c) : 0.0f, ln_1mc = (K > 1) ? TQ_FLOG(1.0f - c) : 0.0f;
  const float ln_1mrho = TQ_FLOG(1.0f - rho);
  const float ln_rhoK = in.on ? TQ_FLOG(rho * (1.0f / (float)K)) : -INFINITY;
  const float r2H = TQ_FRCP(2.0f * H);
  const float lu = 2.0f * TQ_FLOG(r2H);  // log uniform density of (x, y) on (-H, H)^2
  const float ra_ = TQ_FRCP(a), r1a_ = TQ_FRCP(1.0f - a);
  const float rc_ = (K > 1) ? TQ_FRCP(c) : 0.0f, r1c_ = (K > 1) ? TQ_FRCP(1.0f - c) : 0.0f;
  const float rrho_ = in.on ? TQ_FRCP(rho) : 0.0f, r1rho_ = TQ_FRCP(1.0f - rho);
  const float cs = G.cs;
  float tsum[K], sxy[K], dsx[K], dsy[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float tx = (in.x[k] + H) * r2H, ty = (in.y[k] + H) * r2H;
    const float ex = 2.0f * tx - 1.0f, ey = 2.0f * ty - 1.0f;
    tsum[k] = tq_log1p_fast(-ex * ex) + tq_log1p_fast(-ey * ey);  // ln[4 tx (1-tx)] + ln[4 ty (1-ty)]
    sxy[k] = (cs - 1.0f) * tsum[k] - 2.0f * G.lnB_s + lu;
    dsx[k] = (cs - 1.0f) * (TQ_FRCP(tx) - TQ_FRCP(1.0f - tx)) * r2H;
    dsy[k] = (cs - 1.0f) * (TQ_FRCP(ty) - TQ_FRCP(1.0f - ty)) * r2H;
  }

  float Lm[M], Wm[M];
  float d_rho = 0.0f, d_a = 0.0f, d_c = 0.0f, dS[K];
#pragma unroll
  for (int k = 0; k < K; ++k) dS[k] = 0.0f;
#pragma unroll
  for (int mi = 0; mi < M; ++mi) {
    int n1 = 0;
    float w = 1.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      n1 += (mi >> k) & 1;
      w *= ((mi >> k) & 1) ? p1[k] : p0[k];
    }
    Wm[mi] = w;
    const float fn1 = (float)n1, fn0 = (float)(K - n1);
    float T[K + 1];
    T[0] = ln_1mrho + fn1 * ln_a + fn0 * ln_1ma + fn1 * lu;
    float mx = T[0];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (((mi >> k) & 1) && in.on) {
        T[k + 1] = ln_rhoK + sxy[k] + (fn1 - 1.0f) * lu + fn0 * ln_1mc;
        if (K > 1 && n1 > 1) T[k + 1] += (fn1 - 1.0f) * ln_c;
      } else {
        T[k + 1] = -INFINITY;
      }
      mx = fmaxf(mx, T[k + 1]);
    }
    float se = 0.0f, r[K + 1];
#pragma unroll
    for (int th = 0; th <= K; ++th) {
      r[th] = TQ_FEXP(T[th] - mx);
      se += r[th];
    }
    Lm[mi] = mx + TQ_FLOG(se);
    const float rse = TQ_FRCP(se);
#pragma unroll
    for (int th = 0; th <= K; ++th) r[th] *= rse;
    // derivatives of L(m), weighted by the Dice weight
    d_a += w * r[0] * (fn1 * ra_ - fn0 * r1a_);
    float rsum = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      rsum += r[k + 1];
      dS[k] += w * r[k + 1];
    }
    if (K > 1) d_c += w * rsum * ((fn1 - 1.0f) * rc_ - fn0 * r1c_);
    if (in.on) d_rho += w * (-r[0] * r1rho_ + rsum * rrho_);
  }

  // ---- per-spot continuous sites (densities / implicit gradients precomputed per site) -------------
  float Tk[K];
  const float w_sc = C.w_hi - C.w_lo;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float wq = wu * p1[k];
    // height: prior HalfNormal(height_std), guide Gamma(h_loc*h_beta, h_beta)
    const float hl = TQ_FEXP(in.u[TQ_ROW(TQ_P_HLOC, k, K)]), hb = TQ_FEXP(in.u[TQ_ROW(TQ_P_HBETA, k, K)]);
    const float hs = C.height_std, rhs2 = TQ_FRCP(hs * hs);
    const float lp_h = TQ_LN2 - TQ_FLOG(hs) - TQ_LN_SQRT_2PI - 0.5f * in.h[k] * in.h[k] * rhs2;
    const float e_h = in.gh[k] + wq * (-in.h[k] * rhs2);
    {
      float s[TQ_NSITE_TERMS];
      tq_gamma_terms_expand(in.sh[k], in.h[k], hl, hb, s);
      tq_gamma_site_chain(s, in.h[k], hl, hb, e_h, wq, &out->g[TQ_ROW(TQ_P_HLOC, k, K)], &out->g[TQ_ROW(TQ_P_HBETA, k, K)]);
    }
    float t_k = lp_h - in.sh[k][0];
    // width: prior uniform on (w_lo, w_hi), guide AffineBeta(w_mean, w_size, w_lo, w_hi)
    {
      const float sg = tq_sigmoid(in.u[TQ_ROW(TQ_P_WMEAN, k, K)]);
      const float lo = C.w_lo + eps, hi = C.w_hi - eps;
      const float mean = lo + (hi - lo) * sg;
      const float ex = TQ_FEXP(in.u[TQ_ROW(TQ_P_WSIZE, k, K)]);
      float d_mean, d_size;
      float s[TQ_NSITE_TERMS];
      tq_affine_beta_terms_expand(in.sw[k], in.w[k], mean, 2.0f + ex, C.w_lo, C.w_hi, s);
      tq_affine_beta_site_chain(s, mean, 2.0f + ex, C.w_lo, C.w_hi, in.gw[k], wq, &d_mean, &d_size);
      out->g[TQ_ROW(TQ_P_WMEAN, k, K)] = d_mean * (hi - lo) * sg * (1.0f - sg);
      out->g[TQ_ROW(TQ_P_WSIZE, k, K)] = d_size * ex;
      t_k += -TQ_FLOG(w_sc) - in.sw[k][0];
    }
    // x, y: guide AffineBeta(mean, size, -H, H); model-side dependence through L(m)
    {
      const float ex = TQ_FEXP(in.u[TQ_ROW(TQ_P_SIZE, k, K)]);
      const float size = 2.0f + ex;
      const float lo = -H + eps, hi = H - eps;
      const float sgx = tq_sigmoid(in.u[TQ_ROW(TQ_P_XMEAN, k, K)]);
      const float sgy = tq_sigmoid(in.u[TQ_ROW(TQ_P_YMEAN, k, K)]);
      float d_mean, d_size_x, d_size_y;
      const float e_x = in.gx[k] + wu * dS[k] * dsx[k];
      const float e_y = in.gy[k] + wu * dS[k] * dsy[k];
      float s[TQ_NSITE_TERMS];
      tq_affine_beta_terms_expand(in.sx[k], in.x[k], lo + (hi - lo) * sgx, size, -H, H, s);
      tq_affine_beta_site_chain(s, lo + (hi - lo) * sgx, size, -H, H, e_x, wq, &d_mean, &d_size_x);
      out->g[TQ_ROW(TQ_P_XMEAN, k, K)] = d_mean * (hi - lo) * sgx * (1.0f - sgx);
      tq_affine_beta_terms_expand(in.sy[k], in.y[k], lo + (hi - lo) * sgy, size, -H, H, s);
      tq_affine_beta_site_chain(s, lo + (hi - lo) * sgy, size, -H, H, e_y, wq, &d_mean, &d_size_y);
      out->g[TQ_ROW(TQ_P_YMEAN, k, K)] = d_mean * (hi - lo) * sgy * (1.0f - sgy);
      t_k -= in.sx[k][0] + in.sy[k][0];
      out->g[TQ_ROW(TQ_P_SIZE, k, K)] = (d_size_x + d_size_y) * ex;
    }
    Tk[k] = t_k;
  }

  // ---- Dice expectation over m and the gradient w.r.t. the m_probs logits --------------------------
  float Esum = 0.0f, gm[K];
#pragma unroll
  for (int k = 0; k < K; ++k) gm[k] = 0.0f;
#pragma unroll
  for (int mi = 0; mi < M; ++mi) {
    float inner = in.ll[mi] + Lm[mi];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int mk = (mi >> k) & 1;
      inner += mk ? (Tk[k] - lp1[k]) : -lp0[k];
    }
    const float wi = Wm[mi] * inner;
    // W(m) = 0 with inner = -inf (impossible image under m that the guide also excludes) must not give NaN
    const float term = (Wm[mi] == 0.0f) ? 0.0f : wi;
    Esum += term;
#pragma unroll
    for (int k = 0; k < K; ++k) gm[k] += term * ((float)((mi >> k) & 1) - p1[k]);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) out->g[TQ_ROW(TQ_P_MPROBS, k, K)] = wu * gm[k];

  // ---- background: prior Gamma((mu_b/sigma_b)^2, mu_b/sigma_b^2), guide Gamma(b_loc*b_beta, b_beta) ---
  {
    const float mub = TQ_FEXP(in.u_bml), sgb = TQ_FEXP(in.u_bsl);
    const float r0 = mub * TQ_FRCP(sgb * sgb);
    float lp_b, d_v, d_alpha, d_beta;
    tq_gamma_logpdf(in.b, mub, r0, &lp_b, &d_v, &d_alpha, &d_beta);  // loc = a0/r0 = mu_b
    const float a0 = mub * r0;
    // a0 = mu^2/sigma^2, r0 = mu/sigma^2 ; d/d ln mu and d/d ln sigma
    out->g_bml = wu * (d_alpha * 2.0f * a0 + d_beta * r0);
    out->g_bsl = wu * (d_alpha * -2.0f * a0 + d_beta * -2.0f * r0);
    const float bl = TQ_FEXP(in.u[TQ_ROW_BLOC(K)]), bb = TQ_FEXP(in.u[TQ_ROW_BBETA(K)]);
    const float e_b = in.gb + wu * d_v;
    float s[TQ_NSITE_TERMS];
    tq_gamma_terms_expand(in.sb, in.b, bl, bb, s);
    tq_gamma_site_chain(s, in.b, bl, bb, e_b, wu, &out->g[TQ_ROW_BLOC(K)], &out->g[TQ_ROW_BBETA(K)]);
    Esum += lp_b - in.sb[0];
  }

  // ---- global-table partials ---------------------------------------------------------------------
  float d_cs = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) d_cs += dS[k] * (tsum[k] - 2.0f * G.dlnB_s);
  out->d_rho = wu * d_rho;
  out->d_a = wu * d_a;
  out->d_c = wu * d_c;
  out->d_cs = wu * d_cs;
  out->elbo = wu * Esum;
}

// ---- posterior responsibilities (cosmos.compute_probs, cosmos.py:609-672) ---------------------------
// R[k] = sum_m prod_k q(m_k) * r(z = 1, theta = k+1 | m) for one unit and one joint draw of (pi, lamda,
// proximity, x, y).  Same marginal as L(m) above.
template <int K>
TQ_HD void tq_zt_responsibilities(const float* x, const float* y, const float* u_mprobs, const TqGlobals& G, int q,
                                  float H, float* R) {
  constexpr int M = 1 << K;
  const float rho = G.rho[q], a = G.a[q], c = G.c[q];
  const float ln_a = logf(a), ln_1ma = log1pf(-a);
  const float ln_c = (K > 1) ? logf(c) : 0.0f, ln_1mc = (K > 1) ? log1pf(-c) : 0.0f;
  const float ln_1mrho = log1pf(-rho), ln_rhoK = logf(rho / (float)K);
  const float lu = -2.0f * logf(2.0f * H);
  float sxy[K], p1[K], p0[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float ex = x[k] / H, ey = y[k] / H;  // 2t - 1
    sxy[k] = (G.cs - 1.0f) * (log1pf(-ex * ex) + log1pf(-ey * ey)) - 2.0f * G.lnB_s + lu;
    p1[k] = tq_sigmoid(u_mprobs[k]);
    p0[k] = tq_sigmoid(-u_mprobs[k]);
    R[k] = 0.0f;
  }
#pragma unroll
  for (int mi = 0; mi < M; ++mi) {
    int n1 = 0;
    float w = 1.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      n1 += (mi >> k) & 1;
      w *= ((mi >> k) & 1) ? p1[k] : p0[k];
    }
    const float fn1 = (float)n1, fn0 = (float)(K - n1);
    float T[K + 1];
    T[0] = ln_1mrho + fn1 * ln_a + fn0 * ln_1ma + fn1 * lu;
    float mx = T[0];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if ((mi >> k) & 1) {
        T[k + 1] = ln_rhoK + sxy[k] + (fn1 - 1.0f) * lu + fn0 * ln_1mc;
        if (K > 1 && n1 > 1) T[k + 1] += (fn1 - 1.0f) * ln_c;
      } else {
        T[k + 1] = -INFINITY;
      }
      mx = fmaxf(mx, T[k + 1]);
    }
    float se = 0.0f, r[K + 1];
#pragma unroll
    for (int th = 0; th <= K; ++th) {
      r[th] = expf(T[th] - mx);
      se += r[th];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) R[k] += w * r[k + 1] / se;
  }
}

// ---- per-AOI prior terms (cosmos.py:221-227): HalfNormal(mu_b; s1) + HalfNormal(sigma_b; s2) ------
TQ_HD void tq_cosmos_aoi(float u_bml, float u_bsl, float w_n, const TqSiteConsts& C, float* elbo, float* g_bml,
                         float* g_bsl) {
  const float mub = expf(u_bml), sgb = expf(u_bsl);
  const float s1 = C.bg_mean_std, s2 = C.bg_std_std;
  const float lp = 2.0f * (TQ_LN2 - TQ_LN_SQRT_2PI) - logf(s1) - logf(s2) - mub * mub / (2.0f * s1 * s1) -
                   sgb * sgb / (2.0f * s2 * s2);
  *elbo = w_n * lp;
  *g_bml = w_n * (-mub / (s1 * s1)) * mub;
  *g_bsl = w_n * (-sgb / (s2 * s2)) * sgb;
}
