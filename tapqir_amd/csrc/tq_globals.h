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

//
// The crosstalk model (tapqir/models/crosstalk.py:80-87, 279-284, 429-438; Q = C = 2) appends
//   [4+5Q .. 4+7Q)  alpha_mean[q][0..1]   [4+7Q .. 4+8Q)  alpha_size[q]
#define TQ_NGLOBAL(Q) (4 + 5 * (Q))
#define TQ_NGLOBAL_X(Q, xt) (4 + 5 * (Q) + ((xt) ? 3 * (Q) : 0))
// per-rank partial sums that cross GPUs in ONE all-reduce: [gain, cs, elbo, (rho, a, c) x Q] (+ d/d alpha[q][c])
#define TQ_NGSUM(Q) (3 + 3 * (Q))
#define TQ_NGSUM_X(Q, xt) (3 + 3 * (Q) + ((xt) ? (Q) * (Q) : 0))
#define TQ_GS_ALPHA0(Q) (3 + 3 * (Q))
enum { TQ_GS_GAIN = 0, TQ_GS_CS = 1, TQ_GS_ELBO = 2, TQ_GS_Q0 = 3 };

struct TqGlobalBase {  // base draws behind the global latents
  double gain_g;
  double prox_t;
  double lamda_g[TQ_MAXQ];
  double pi_x[TQ_MAXQ][2];
  double alpha_x[2][2];  // crosstalk model only
};

struct TqGlobalConsts {
  int K, P, Q;
  int xt;  // crosstalk model: Q more sites (alpha_q)
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

// Constrained parameters of ONE global site (scalars only: no runtime-indexed local arrays, so the
// single-lane global kernels need no scratch memory).
//   gain / lamda_q : loc, beta                 (Gamma(loc*beta, beta))
//   proximity      : loc, size, sg, ex         (AffineBeta(loc, size, 0, Hs); sg = sigmoid(u), ex = exp(u))
//   pi_q           : m0, m1, size              (Dirichlet([m0, m1] * size))
struct TqGlobalSite {
  double loc, beta, size, sg, ex, m0, m1;
};

// Global guide sites are numbered s = 0: gain, 1: proximity, 2+q: lamda_q, 2+Q+q: pi_q.  Each site
// has its own Philox stream, draws its own base variates and fills its own fields of TqGlobals, so
// the device runs one site per wave (tq_cosmos.hip) and the host loops over s.
#define TQ_NGSITES(Q) (2 + 2 * (Q))
#define TQ_NGSITES_X(Q, xt) (2 + 2 * (Q) + ((xt) ? (Q) : 0))   // crosstalk: 2+2Q+q = alpha_q

TQ_HD void tq_globals_constrain_site(const float* u, const TqGlobalConsts& C, int s, TqGlobalSite* p) {
  const int Q = C.Q;
  const double Hs = (C.P + 1) / sqrt(12.0);
  p->loc = p->beta = p->size = p->sg = p->ex = p->m0 = p->m1 = 0.0;
  if (s == 0) {
    p->loc = exp((double)u[0]);
    p->beta = exp((double)u[1]);
  } else if (s == 1) {
    p->sg = tq_sigmoid_d(u[2]);
    p->loc = (Hs - C.eps) * p->sg;  // interval(0, Hs - eps)
    p->ex = exp((double)u[3]);
    p->size = 2.0 + p->ex;
  } else if (s < 2 + Q) {
    const int q = s - 2;
    p->loc = exp((double)u[4 + q]);
    p->beta = exp((double)u[4 + Q + q]);
  } else {
    // pi_q, or alpha_q of the crosstalk model: softmax mean of 2 components and a positive size
    const bool is_alpha = s >= 2 + 2 * Q;
    const int q = is_alpha ? s - 2 - 2 * Q : s - 2 - Q;
    const int im = is_alpha ? 4 + 5 * Q + 2 * q : 4 + 2 * Q + 2 * q;
    const int is = is_alpha ? 4 + 7 * Q + q : 4 + 4 * Q + q;
    const double u0 = u[im], u1 = u[im + 1];
    const double mx = u0 > u1 ? u0 : u1;
    const double e0 = exp(u0 - mx), e1 = exp(u1 - mx);
    p->m0 = e0 / (e0 + e1);
    p->m1 = e1 / (e0 + e1);
    p->size = exp((double)u[is]);
  }
}

TQ_HD double tq_clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Draw (if `draw`) the base variates of site s (cosmos.py:342-368) and derive its latents/tables.
TQ_HD void tq_globals_sample_site(int s, const TqGlobalSite& p, const TqGlobalConsts& C, uint64_t seed, uint32_t step,
                                  int draw, TqGlobalBase* b, TqGlobals* G) {
  const double Hs = (C.P + 1) / sqrt(12.0), H = (C.P + 1) / 2.0;
  const double tiny = 1.1754943508222875e-38;
  const int Q = C.Q;
  TqPhilox ph;
  tq_philox_init(&ph, seed, step, /*site=*/0xFFF, /*elem=*/(uint64_t)s);
  if (s == 0) {
    if (draw) b->gain_g = tq_sample_std_gamma(&ph, (float)(p.loc * p.beta));
    double gain = b->gain_g / p.beta;
    if (gain < tiny) gain = tiny;
    G->gain = (float)gain;
  } else if (s == 1) {
    if (draw) {
      const double c1 = p.size * p.loc / Hs, c0 = p.size - c1;
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
    if (draw) b->lamda_g[q] = tq_sample_std_gamma(&ph, (float)(p.loc * p.beta));
    double lam = b->lamda_g[q] / p.beta;
    if (lam < tiny) lam = tiny;
    double a, da, c = 0.5, dc = 0.0;
    tq_mean_frac(lam, C.K, &a, &da);
    if (C.K > 1) tq_mean_frac(lam, C.K - 1, &c, &dc);
    G->lamda[q] = (float)lam;
    G->a[q] = (float)a;
    G->c[q] = (float)c;
  } else {
    const bool is_alpha = s >= 2 + 2 * Q;
    const int q = is_alpha ? s - 2 - 2 * Q : s - 2 - Q;
    double* x = is_alpha ? b->alpha_x[q] : b->pi_x[q];
    if (draw) {
      const double g0 = tq_sample_std_gamma(&ph, (float)(p.m0 * p.size));
      const double g1 = tq_sample_std_gamma(&ph, (float)(p.m1 * p.size));
      x[0] = g0 / (g0 + g1);
      x[1] = g1 / (g0 + g1);
    }
    if (is_alpha) {
      G->alpha[q][0] = (float)x[0];
      G->alpha[q][1] = (float)x[1];
    } else {
      G->rho[q] = (float)x[1];
    }
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
// of the local part.  Writes the site's entries of the global gradient vector `g` and returns the
// site's part of the ELBO (model - guide log-density).
TQ_HD double tq_globals_grad_site(int s, const TqGlobalSite& p, const TqGlobalBase& b, const TqGlobals& G,
                                  const TqGlobalConsts& C, const double* gsum, float* g) {
  const int Q = C.Q;
  const double Hs = (C.P + 1) / sqrt(12.0), H = (C.P + 1) / 2.0;
  if (s == 0 || (s >= 2 && s < 2 + Q)) {
    // ---- Gamma sites: gain (prior HalfNormal(gain_std)) and lamda_q (prior Exponential(rate)) ----
    double v, gb, e, lp;
    int i_loc, i_beta;
    if (s == 0) {
      v = G.gain;
      gb = b.gain_g;
      const double sd = C.gain_std;
      lp = log(2.0) - log(sd) - 0.91893853320467274178 - v * v / (2 * sd * sd);
      e = gsum[TQ_GS_GAIN] - v / (sd * sd);
      i_loc = 0; i_beta = 1;
    } else {
      const int q = s - 2;
      v = G.lamda[q];
      gb = b.lamda_g[q];
      const double rate = C.lamda_rate;
      double a, da, c = 0.5, dc = 0.0;
      tq_mean_frac(v, C.K, &a, &da);
      if (C.K > 1) tq_mean_frac(v, C.K - 1, &c, &dc);
      lp = log(rate) - rate * v;
      e = gsum[TQ_GS_Q0 + 3 * q + 1] * da + gsum[TQ_GS_Q0 + 3 * q + 2] * dc - rate;
      i_loc = 4 + q; i_beta = 4 + Q + q;
    }
    double d_loc, d_beta;
    const double lq = tq_gamma_site_d(v, gb, p.loc, p.beta, e, &d_loc, &d_beta);
    g[i_loc] = (float)d_loc;
    g[i_beta] = (float)d_beta;
    return lp - lq;
  }
  // ---- Beta / Dirichlet sites: proximity (AffineBeta on (0, Hs)) and pi_q (2-component Dirichlet) ----
  double x0, x1, c0, c1, go0, go1, lp, sc;  // x0 = first component; (c0, c1) concentrations of (x0, x1)
  const bool is_alpha = s >= 2 + 2 * Q;
  if (s == 1) {
    const double sg = G.proximity, rate = C.proximity_rate;
    lp = log(rate) - rate * sg;
    // write the AffineBeta as a Dirichlet over (t, 1-t), t = sigma / Hs: objective derivative w.r.t. t
    c0 = p.size * p.loc / Hs;
    c1 = p.size - c0;
    x0 = sg / Hs;
    const double dcs_dsigma = -H * H / (sg * sg * sg);
    const bool clamped = (sg <= C.eps * Hs) || (sg >= Hs - C.eps * Hs);
    go0 = clamped ? 0.0 : (gsum[TQ_GS_CS] * dcs_dsigma - rate) * Hs;  // pathwise part only
    go1 = 0.0;
    sc = Hs;
    x1 = 1.0 - x0;
  } else if (is_alpha) {
    // alpha_q: prior Dirichlet(1 + 9 [q == c]) (crosstalk.py:82-87); likelihood derivative from the cross-unit sums
    const int q = s - 2 - 2 * Q;
    x0 = b.alpha_x[q][0];
    x1 = b.alpha_x[q][1];
    c0 = p.m0 * p.size;
    c1 = p.m1 * p.size;
    const double a0 = q == 0 ? 10.0 : 1.0, a1 = q == 1 ? 10.0 : 1.0;
    lp = 2.30258509299404568402 + (a0 - 1.0) * log(x0) + (a1 - 1.0) * log(x1);  // lgamma(11) - lgamma(10) - lgamma(1) = ln 10
    go0 = gsum[TQ_GS_ALPHA0(Q) + 2 * q + 0] + (a0 - 1.0) / x0;
    go1 = gsum[TQ_GS_ALPHA0(Q) + 2 * q + 1] + (a1 - 1.0) / x1;
    sc = 1.0;
  } else {
    const int q = s - 2 - Q;
    x0 = b.pi_x[q][0];
    x1 = b.pi_x[q][1];
    c0 = p.m0 * p.size;
    c1 = p.m1 * p.size;
    lp = -2.0 * 0.57236494292470008707 - 0.5 * log(x0) - 0.5 * log(1.0 - x0);  // Dirichlet(1/2,1/2): lgamma(1/2) = ln sqrt(pi)
    go0 = -0.5 / x0;
    go1 = gsum[TQ_GS_Q0 + 3 * q + 0] - 0.5 / b.pi_x[q][1];
    sc = 1.0;
  }
  const double tot = c0 + c1;
  double lg0, dg0, lg1, dg1, lgt, dgt;
  tq_lgamma_digamma_d(c0, &lg0, &dg0);
  tq_lgamma_digamma_d(c1, &lg1, &dg1);
  tq_lgamma_digamma_d(tot, &lgt, &dgt);
  const double lq = lgt - lg0 - lg1 + (c0 - 1) * log(x0) + (c1 - 1) * log(x1) - log(sc);
  // total derivative of the objective w.r.t. the draw = outside part - d lq / d x
  const double e0 = go0 - (c0 - 1) / x0, e1 = go1 - (c1 - 1) / x1;
  const double dot = x0 * e0 + x1 * e1;
  double dgr[2];
#pragma nounroll
  for (int j = 0; j < 2; ++j) dgr[j] = (double)tq_dirichlet_grad((float)(j ? x1 : x0), (float)(j ? c1 : c0), (float)tot);
  const bool frozen = (s == 1) && (go0 == 0.0) && ((G.proximity <= C.eps * Hs) || (G.proximity >= Hs - C.eps * Hs));
  const double g_c0 = (frozen ? 0.0 : dgr[0] * (e0 - dot)) - (dgt - dg0 + log(x0));
  const double g_c1 = (frozen ? 0.0 : dgr[1] * (e1 - dot)) - (dgt - dg1 + log(x1));
  if (s == 1) {
    // c0 = size*loc/Hs, c1 = size*(Hs-loc)/Hs
    const double d_loc = (g_c0 - g_c1) * p.size / Hs;
    const double d_size = g_c0 * p.loc / Hs + g_c1 * (Hs - p.loc) / Hs;
    g[2] = (float)(d_loc * (Hs - C.eps) * p.sg * (1 - p.sg));
    g[3] = (float)(d_size * p.ex);
  } else {
    const int q = is_alpha ? s - 2 - 2 * Q : s - 2 - Q;
    const int im = is_alpha ? 4 + 5 * Q + 2 * q : 4 + 2 * Q + 2 * q;
    const int is = is_alpha ? 4 + 7 * Q + q : 4 + 4 * Q + q;
    const double d_ps = g_c0 * p.m0 + g_c1 * p.m1;
    const double d_m0 = g_c0 * p.size, d_m1 = g_c1 * p.size;
    const double mdot = p.m0 * d_m0 + p.m1 * d_m1;
    g[im + 0] = (float)(p.m0 * (d_m0 - mdot));
    g[im + 1] = (float)(p.m1 * (d_m1 - mdot));
    g[is] = (float)(d_ps * p.size);
  }
  return lp - lq;
}
