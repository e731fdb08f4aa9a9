// tq_globals.h -- the global sites of the cosmos model (gain, pi, lamda, proximity): guide draws,
// the tables every unit needs (probs_m rows, target-specific position prior), and the gradient
// of the global part of the ELBO.  Host+device inline, double precision (a handful of scalars
// per step; runs on one lane).
//
// Reference: tapqir/models/cosmos.py:170-191 (model), 342-368 (guide), 471-528 (parameters);
// tapqir/distributions/util.py:67-130 (truncated_poisson_probs, probs_m).
//
// Unconstrained global parameter vector (NG = 4 + 5Q):
//   [0] gain_loc  [1] gain_beta  [2] proximity_loc  [3] proximity_size
//   [4      .. 4+Q)   lamda_loc[q]      [4+Q  .. 4+2Q)  lamda_beta[q]
//   [4+2Q   .. 4+4Q)  pi_mean[q][0..1]  [4+4Q .. 4+5Q)  pi_size[q]
#pragma once
#include "tq_math.h"
#include "tq_site.h"

#define TQ_NGLOBAL(Q) (4 + 5 * (Q))
// per-rank partial sums that cross GPUs in ONE all-reduce: [gain, cs, elbo, (rho, a, c) x Q]
#define TQ_NGSUM(Q) (3 + 3 * (Q))
enum { TQ_GS_GAIN = 0, TQ_GS_CS = 1, TQ_GS_ELBO = 2, TQ_GS_Q0 = 3 };

struct TqGlobalBase {  // base draws behind the global latents
  double gain_g;
  double prox_t;
  double lamda_g[TQ_MAXQ];
  double pi_x[TQ_MAXQ][2];
};

struct TqGlobalConsts {
  int K, P, Q;
  double eps;
  double gain_std, lamda_rate, proximity_rate;
};

// fraction of occupied spot slots under the truncated Poisson: sum_l l TP(l; lam, K) / K, and d/dlam
TQ_HD void tq_mean_frac(double lam, int K, double* val, double* dval) {
  // pois(l) for l = 0..K-1
  double pois = exp(-lam), prev = 0.0;
  double head = 0.0, dhead = 0.0;     // sum_{l<K} pois(l) and its derivative
  double first = 0.0, dfirst = 0.0;   // sum_{1<=l<K} l pois(l) and its derivative
  for (int l = 0; l < K; ++l) {
    const double dp = prev - pois;  // d pois(l) / d lam = pois(l-1) - pois(l)
    head += pois;
    dhead += dp;
    first += l * pois;
    dfirst += l * dp;
    prev = pois;
    pois *= lam / (double)(l + 1);
  }
  *val = (first + K * (1.0 - head)) / K;
  *dval = (dfirst - K * dhead) / K;
}

TQ_HD double tq_sigmoid_d(double u) { return u >= 0 ? 1.0 / (1.0 + exp(-u)) : exp(u) / (1.0 + exp(u)); }

struct TqGlobalParams {  // constrained view of the unconstrained vector
  double gain_loc, gain_beta, prox_loc, prox_size, prox_sg, prox_ex;
  double lamda_loc[TQ_MAXQ], lamda_beta[TQ_MAXQ];
  double pi_mean[TQ_MAXQ][2], pi_size[TQ_MAXQ];
};

TQ_HD void tq_globals_constrain(const float* u, const TqGlobalConsts& C, TqGlobalParams* p) {
  const int Q = C.Q;
  const double Hs = (C.P + 1) / sqrt(12.0);
  p->gain_loc = exp((double)u[0]);
  p->gain_beta = exp((double)u[1]);
  p->prox_sg = tq_sigmoid_d(u[2]);
  p->prox_loc = (Hs - C.eps) * p->prox_sg;  // interval(0, Hs - eps)
  p->prox_ex = exp((double)u[3]);
  p->prox_size = 2.0 + p->prox_ex;
  for (int q = 0; q < Q; ++q) {
    p->lamda_loc[q] = exp((double)u[4 + q]);
    p->lamda_beta[q] = exp((double)u[4 + Q + q]);
    const double u0 = u[4 + 2 * Q + 2 * q], u1 = u[4 + 2 * Q + 2 * q + 1];
    const double mx = u0 > u1 ? u0 : u1;
    const double e0 = exp(u0 - mx), e1 = exp(u1 - mx);
    p->pi_mean[q][0] = e0 / (e0 + e1);
    p->pi_mean[q][1] = e1 / (e0 + e1);
    p->pi_size[q] = exp((double)u[4 + 4 * Q + q]);
  }
}

// Global guide sites are numbered s = 0: gain, 1: proximity, 2+q: lamda_q, 2+Q+q: pi_q.  Each site
// has its own Philox stream, draws its own base variates and fills its own fields of TqGlobals, so
// the device runs one site per wave (tq_cosmos.hip) and the host loops over s.
#define TQ_NGSITES(Q) (2 + 2 * (Q))

TQ_HD double tq_clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Draw (if `draw`) the base variates of site s (cosmos.py:342-368) and derive its latents/tables.
TQ_HD void tq_globals_sample_site(int s, const TqGlobalParams& p, const TqGlobalConsts& C, uint64_t seed, uint32_t step,
                                  int draw, TqGlobalBase* b, TqGlobals* G) {
  const double Hs = (C.P + 1) / sqrt(12.0), H = (C.P + 1) / 2.0;
  const double tiny = 1.1754943508222875e-38;
  const int Q = C.Q;
  TqPhilox ph;
  tq_philox_init(&ph, seed, step, /*site=*/0xFFF, /*elem=*/(uint64_t)s);
  if (s == 0) {
    if (draw) b->gain_g = tq_sample_std_gamma(&ph, (float)(p.gain_loc * p.gain_beta));
    double gain = b->gain_g / p.gain_beta;
    if (gain < tiny) gain = tiny;
    G->gain = (float)gain;
  } else if (s == 1) {
    if (draw) {
      const double c1 = p.prox_size * p.prox_loc / Hs, c0 = p.prox_size - c1;
      const double g1 = tq_sample_std_gamma(&ph, (float)c1), g0 = tq_sample_std_gamma(&ph, (float)c0);
      b->prox_t = g1 / (g1 + g0);
    }
    const double sigma = tq_clampd(Hs * b->prox_t, C.eps * Hs, Hs - C.eps * Hs);
    const double cs = 0.5 * ((H / sigma) * (H / sigma) - 1.0);
    double lg1, dg1, lg2, dg2;
    tq_lgamma_digamma_d(cs, &lg1, &dg1);
    tq_lgamma_digamma_d(2.0 * cs, &lg2, &dg2);
    G->proximity = (float)sigma;
    G->cs = (float)cs;
    G->lnB_s = (float)(2.0 * lg1 - lg2 + 2.0 * (cs - 1.0) * 0.69314718055994530942);
    G->dlnB_s = (float)(2.0 * dg1 - 2.0 * dg2 + 2.0 * 0.69314718055994530942);
  } else if (s < 2 + Q) {
    const int q = s - 2;
    if (draw) b->lamda_g[q] = tq_sample_std_gamma(&ph, (float)(p.lamda_loc[q] * p.lamda_beta[q]));
    double lam = b->lamda_g[q] / p.lamda_beta[q];
    if (lam < tiny) lam = tiny;
    double a, da, c = 0.5, dc = 0.0;
    tq_mean_frac(lam, C.K, &a, &da);
    if (C.K > 1) tq_mean_frac(lam, C.K - 1, &c, &dc);
    G->lamda[q] = (float)lam;
    G->a[q] = (float)a;
    G->c[q] = (float)c;
  } else {
    const int q = s - 2 - Q;
    if (draw) {
      const double g0 = tq_sample_std_gamma(&ph, (float)(p.pi_mean[q][0] * p.pi_size[q]));
      const double g1 = tq_sample_std_gamma(&ph, (float)(p.pi_mean[q][1] * p.pi_size[q]));
      b->pi_x[q][0] = g0 / (g0 + g1);
      b->pi_x[q][1] = g1 / (g0 + g1);
    }
    G->rho[q] = (float)b->pi_x[q][1];
  }
}

// double-precision Gamma / Beta log-densities with derivatives
TQ_HD void tq_gamma_logpdf_d(double v, double alpha, double beta, double* lp, double* d_v, double* d_alpha, double* d_beta) {
  double lg, dg;
  tq_lgamma_digamma_d(alpha, &lg, &dg);
  *lp = alpha * log(beta) + (alpha - 1.0) * log(v) - beta * v - lg;
  *d_v = (alpha - 1.0) / v - beta;
  *d_alpha = log(beta) + log(v) - dg;
  *d_beta = alpha / beta - v;
}

// Gamma(loc*beta, beta) guide site: returns lq; d objective / d u_loc, d u_beta (loc, beta = exp(u))
TQ_HD double tq_gamma_site_d(double v, double g_base, double loc, double beta, double e_v, double* d_loc_u, double* d_beta_u) {
  const double alpha = loc * beta;
  double lq, d_v, d_alpha, d_beta;
  tq_gamma_logpdf_d(v, alpha, beta, &lq, &d_v, &d_alpha, &d_beta);
  const double ev_tot = e_v - d_v;
  const double dv_dalpha = (double)tq_std_gamma_grad((float)alpha, (float)g_base) / beta;
  const double g_alpha = ev_tot * dv_dalpha - d_alpha;
  const double g_beta = ev_tot * (-v / beta) - d_beta;
  *d_loc_u = g_alpha * alpha;
  *d_beta_u = g_alpha * alpha + g_beta * beta;
  return lq;
}

// Gradient of the ELBO w.r.t. the unconstrained parameters of global site s, given the cross-unit sums
//   gsum[TQ_GS_GAIN] = d/d gain, gsum[TQ_GS_CS] = d/d cs, gsum[TQ_GS_Q0+3q..] = d/d (rho_q, a_q, c_q)
// of the local part.  Writes the site's entries of g_u and returns the site's part of the ELBO
// (model - guide log-density).
TQ_HD double tq_globals_grad_site(int s, const TqGlobalParams& p, const TqGlobalBase& b, const TqGlobals& G,
                                  const TqGlobalConsts& C, const double* gsum, double* g_u) {
  const int Q = C.Q;
  const double Hs = (C.P + 1) / sqrt(12.0), H = (C.P + 1) / 2.0;
  if (s == 0) {  // ---- gain: prior HalfNormal(gain_std) ----
    const double g = G.gain, sd = C.gain_std;
    const double lp = log(2.0) - log(sd) - 0.91893853320467274178 - g * g / (2 * sd * sd);
    const double e = gsum[TQ_GS_GAIN] - g / (sd * sd);
    const double lq = tq_gamma_site_d(g, b.gain_g, p.gain_loc, p.gain_beta, e, &g_u[0], &g_u[1]);
    return lp - lq;
  }
  if (s == 1) {  // ---- proximity: prior Exponential(rate); guide AffineBeta(loc, size, 0, Hs) ----
    const double sg = G.proximity, rate = C.proximity_rate;
    const double lp = log(rate) - rate * sg;
    const double dcs_dsigma = -H * H / (sg * sg * sg);
    const double t = sg / Hs;
    const double size = p.prox_size, c1 = size * p.prox_loc / Hs, c0 = size - c1;
    double lg1, dg1, lg0, dg0, lgt, dgt;
    tq_lgamma_digamma_d(c1, &lg1, &dg1);
    tq_lgamma_digamma_d(c0, &lg0, &dg0);
    tq_lgamma_digamma_d(size, &lgt, &dgt);
    const double lq = (c1 - 1) * log(t) + (c0 - 1) * log1p(-t) + lgt - lg1 - lg0 - log(Hs);
    const double dlq_dsg = ((c1 - 1) / t - (c0 - 1) / (1 - t)) / Hs;
    const double e = gsum[TQ_GS_CS] * dcs_dsigma - rate - dlq_dsg;
    const bool clamped = (sg <= C.eps * Hs) || (sg >= Hs - C.eps * Hs);
    double dd[2];
#pragma nounroll
    for (int j = 0; j < 2; ++j)
      dd[j] = clamped ? 0.0
                      : (double)tq_dirichlet_grad((float)(j ? 1 - b.prox_t : b.prox_t), (float)(j ? c0 : c1), (float)size);
    const double dy1 = Hs * dd[0] * (1 - b.prox_t), dy0 = -Hs * dd[1] * b.prox_t;
    const double g_c1 = e * dy1 - (log(t) - dg1 + dgt);
    const double g_c0 = e * dy0 - (log1p(-t) - dg0 + dgt);
    const double d_loc = (g_c1 - g_c0) * size / Hs;
    const double d_size = g_c1 * p.prox_loc / Hs + g_c0 * (Hs - p.prox_loc) / Hs;
    g_u[2] = d_loc * (Hs - C.eps) * p.prox_sg * (1 - p.prox_sg);
    g_u[3] = d_size * p.prox_ex;
    return lp - lq;
  }
  if (s < 2 + Q) {  // ---- lamda_q: prior Exponential(rate) ----
    const int q = s - 2;
    const double lam = G.lamda[q], rate = C.lamda_rate;
    double a, da, c = 0.5, dc = 0.0;
    tq_mean_frac(lam, C.K, &a, &da);
    if (C.K > 1) tq_mean_frac(lam, C.K - 1, &c, &dc);
    const double e = gsum[TQ_GS_Q0 + 3 * q + 1] * da + gsum[TQ_GS_Q0 + 3 * q + 2] * dc - rate;
    const double lq = tq_gamma_site_d(lam, b.lamda_g[q], p.lamda_loc[q], p.lamda_beta[q], e, &g_u[4 + q], &g_u[4 + Q + q]);
    return log(rate) - rate * lam - lq;
  }
  {  // ---- pi_q: prior Dirichlet(1/2, 1/2); guide Dirichlet(pi_mean * pi_size) ----
    const int q = s - 2 - Q;
    const double x[2] = {b.pi_x[q][0], b.pi_x[q][1]};
    const double ps = p.pi_size[q];
    const double c[2] = {p.pi_mean[q][0] * ps, p.pi_mean[q][1] * ps};
    const double tot = c[0] + c[1];
    double lg0, dg0, lg1, dg1, lgt, dgt, lgh, dgh;
    tq_lgamma_digamma_d(c[0], &lg0, &dg0);
    tq_lgamma_digamma_d(c[1], &lg1, &dg1);
    tq_lgamma_digamma_d(tot, &lgt, &dgt);
    tq_lgamma_digamma_d(0.5, &lgh, &dgh);
    const double lp = -2.0 * lgh - 0.5 * log(x[0]) - 0.5 * log(x[1]);  // lgamma(1) = 0
    const double lq = lgt - lg0 - lg1 + (c[0] - 1) * log(x[0]) + (c[1] - 1) * log(x[1]);
    const double go0 = (-0.5 - (c[0] - 1)) / x[0];
    const double go1 = gsum[TQ_GS_Q0 + 3 * q + 0] + (-0.5 - (c[1] - 1)) / x[1];
    const double dot = x[0] * go0 + x[1] * go1;
    double dgr[2];
#pragma nounroll
    for (int j = 0; j < 2; ++j) dgr[j] = (double)tq_dirichlet_grad((float)x[j], (float)c[j], (float)tot);
    const double g_c0 = dgr[0] * (go0 - dot) - (dgt - dg0 + log(x[0]));
    const double g_c1 = dgr[1] * (go1 - dot) - (dgt - dg1 + log(x[1]));
    const double d_ps = g_c0 * p.pi_mean[q][0] + g_c1 * p.pi_mean[q][1];
    const double d_m0 = g_c0 * ps, d_m1 = g_c1 * ps;
    const double mdot = p.pi_mean[q][0] * d_m0 + p.pi_mean[q][1] * d_m1;
    g_u[4 + 2 * Q + 2 * q + 0] = p.pi_mean[q][0] * (d_m0 - mdot);
    g_u[4 + 2 * Q + 2 * q + 1] = p.pi_mean[q][1] * (d_m1 - mdot);
    g_u[4 + 4 * Q + q] = d_ps * ps;
    return lp - lq;
  }
}
