// tq_bodies.h -- per-work-item bodies of the cosmos step kernels (host+device inline).
// The __global__ wrappers in tq_cosmos.hip call these with the work-item index; the CPU test
// harness (tests/hostcheck) calls the same bodies in plain loops on host memory.
#pragma once
#include "../../include/tapqir_hip.h"
#include "tq_globals.h"
#include "tq_site.h"

struct TqUnitIdx {
  int n, c;
  int64_t u;      // dataset unit index (n*F + f)*C + c
  uint64_t elem;  // GLOBAL unit index (AOI sharding: n counted from the first AOI of rank 0): RNG stream id
};

// Minibatch position i -> dataset unit.  32-bit arithmetic (B < 2^31 is checked by the callers' hosts; a 64-bit
// integer division costs ~100 VALU instructions per lane) and no division by the frame count for contiguous batches.
TQ_HD TqUnitIdx tq_decode_unit(const tq_cosmos_args& a, int64_t i) {
  TqUnitIdx r;
  const uint32_t iu = (uint32_t)i, C = (uint32_t)a.C;
  uint32_t ab = iu, c = 0;
  if (C > 1) {
    ab = iu / C;
    c = iu - ab * C;
  }
  r.c = (int)c;
  if (a.ndx == nullptr && a.fdx == nullptr && a.fb == a.F) {  // contiguous in frames and AOIs: unit i IS dataset unit i
    r.u = i;
    r.n = (int)(ab / (uint32_t)a.F);
  } else {
    const uint32_t fb = (uint32_t)a.fb;
    const uint32_t ai = ab / fb, bi = ab - ai * fb;
    r.n = a.ndx ? a.ndx[ai] : (int)ai;
    const int f = a.fdx ? a.fdx[bi] : (int)bi;
    r.u = ((int64_t)r.n * a.F + f) * a.C + c;
  }
  r.elem = (uint64_t)r.u + (uint64_t)a.n_offset * (uint64_t)a.F * (uint64_t)a.C;
  return r;
}

TQ_HD int64_t tq_num_units(const tq_cosmos_args& a) { return (int64_t)a.Nt * a.F * a.C; }
TQ_HD int64_t tq_batch_units(const tq_cosmos_args& a) { return (int64_t)a.nb * a.fb * a.C; }
TQ_HD int64_t tq_aoi_base(const tq_cosmos_args& a) { return (int64_t)TQ_NLOCAL(a.K) * tq_num_units(a); }
TQ_HD int64_t tq_global_base(const tq_cosmos_args& a) { return tq_aoi_base(a) + 2 * (int64_t)a.Nt * a.C; }
TQ_HD int tq_num_gsum(const tq_cosmos_args& a) { return TQ_NGSUM_X(a.C, a.crosstalk); }
TQ_HD int tq_num_gsites(const tq_cosmos_args& a) { return TQ_NGSITES_X(a.C, a.crosstalk); }
TQ_HD int64_t tq_num_params(const tq_cosmos_args& a) { return tq_global_base(a) + TQ_NGLOBAL_X(a.C, a.crosstalk); }

TQ_HD TqGlobalConsts tq_global_consts(const tq_cosmos_args& a) {
  TqGlobalConsts c;
  c.K = a.K; c.P = a.P; c.Q = a.C;
  c.xt = a.crosstalk;
  c.eps = a.eps;
  c.gain_std = a.gain_std; c.lamda_rate = a.lamda_rate; c.proximity_rate = a.proximity_rate;
  return c;
}

// ---- global sites: draw + tables, one work item per site s in [0, 2+2Q) ---------------------------------
TQ_HD void tq_body_sample_globals(const tq_cosmos_args& a, int s) {
  const TqGlobalConsts C = tq_global_consts(a);
  TqGlobalSite p;
  tq_globals_constrain_site(a.params + tq_global_base(a), C, s, &p);
  tq_globals_sample_site(s, p, C, a.seed, a.step, a.draw_globals, (TqGlobalBase*)a.gbase, (TqGlobals*)a.globals);
}

// Streaming accesses (device: non-temporal loads / stores): data that is touched once per step and is not wanted in the
// L2 -- the Adam moments of the per-unit phase of a FULL-BATCH step (57 MB in, 57 MB out at c2; 0.2381 -> 0.2306 ms per
// step on one box, 0.2394 -> 0.2384 on another).  Not in minibatch steps, whose per-unit phase reads what the replay of
// the same launch has just written (53.5 -> 55.7 us with the hint).
#if defined(__HIP_DEVICE_COMPILE__)
#define TQ_LOAD_STREAM(p) __builtin_nontemporal_load(p)
#define TQ_STORE_STREAM(v, p) __builtin_nontemporal_store(v, p)
// a value another workgroup of the SAME launch has published (release at device scope + flag): a device-scope load goes past
// the CU's vector cache and a stale line in this XCD's L2 -- without the acquire fence that invalidates both for everybody
#define TQ_LOAD_COHERENT(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define TQ_LOAD_STREAM(p) (*(p))
#define TQ_STORE_STREAM(v, p) (*(p) = (v))
#define TQ_LOAD_COHERENT(p) (*(p))
#endif

// ---- local guide sites: work item (site, i), site in [0, 1+4K): b, h[k], w[k], x[k], y[k]; row t = site * B + i ---
// Draws the latent (or takes it from `lat` when draw_locals == 0) and evaluates the site's guide
// terms (tq_site.h: TQ_NSITE_TERMS per site).  Lanes of a wave share the site kind (site-major
// order), so the Gamma and Beta code paths do not diverge inside a wave.
// The body in three parts (the sampling kernels put a workgroup-wide regime compaction between the draw and the terms of
// the AffineBeta sites, tq_cosmos.hip: tq_site_beta_compact):
//   tq_site_draw   parameters -> draw (or the given value), everything the terms need
//   tq_site_terms  log q, its derivatives, implicit gradients (dd: the pair of tq_dirichlet_grad values if the caller
//                  has them already)
//   tq_site_store
struct TqSiteDraw {
  float val;
  float p0, p1;  // Gamma: loc, beta;  AffineBeta: mean, size
  float lo, hi;  // AffineBeta
  int64_t t;     // row of the site in lat / site
};
TQ_HD TqSiteDraw tq_site_draw(const tq_cosmos_args& a, int site, int64_t i) {
  const int K = a.K;
  const int64_t B = tq_batch_units(a), U = tq_num_units(a);
  TqSiteDraw d;
  d.t = (int64_t)site * B + i;
  const TqUnitIdx ix = tq_decode_unit(a, i);
  const float* P = a.params;
  TqPhilox s;
  tq_philox_init(&s, a.seed, a.step, (uint32_t)site, ix.elem);
  const float tiny = 1.17549435e-38f;
  d.val = a.lat[d.t];
  d.lo = d.hi = 0.0f;
  if (site <= K) {  // Gamma(loc*beta, beta): background (site 0) or height
    const int rl = site == 0 ? TQ_ROW_BLOC(K) : TQ_ROW(TQ_P_HLOC, site - 1, K);
    const int rb = site == 0 ? TQ_ROW_BBETA(K) : TQ_ROW(TQ_P_HBETA, site - 1, K);
    const float ul = P[rl * U + ix.u], ub = P[rb * U + ix.u];
    const float loc = TQ_FEXP(ul), beta = TQ_FEXP(ub);
    if (a.draw_locals) d.val = fmaxf(tq_sample_std_gamma(&s, loc * beta) * TQ_FRCP(beta), tiny);
    d.p0 = loc;
    d.p1 = beta;
  } else {  // AffineBeta
    const int j = site - 1 - K;
    const int kind = j / K, k = j % K;  // 0: width, 1: x, 2: y
    float lo, hi, um, us;
    if (kind == 0) {
      lo = a.width_min; hi = a.width_max;
      um = P[TQ_ROW(TQ_P_WMEAN, k, K) * U + ix.u];
      us = P[TQ_ROW(TQ_P_WSIZE, k, K) * U + ix.u];
    } else {
      const float H = 0.5f * (a.P + 1);
      lo = -H; hi = H;
      um = P[TQ_ROW(kind == 1 ? TQ_P_XMEAN : TQ_P_YMEAN, k, K) * U + ix.u];
      us = P[TQ_ROW(TQ_P_SIZE, k, K) * U + ix.u];
    }
    const float sc = hi - lo;
    const float mean = (lo + a.eps) + (sc - 2.0f * a.eps) * tq_sigmoid(um);
    const float size = 2.0f + TQ_FEXP(us);
    if (a.draw_locals) {
      const float rsc = TQ_FRCP(sc);
      const float c1 = size * (mean - lo) * rsc, c0 = size * (hi - mean) * rsc;
      const float g1 = tq_sample_std_gamma(&s, c1), g0 = tq_sample_std_gamma(&s, c0);
      float tt = g1 * TQ_FRCP(g1 + g0);
      tt = fminf(fmaxf(tt, tiny), 1.0f - 5.96046448e-08f);  // torch._sample_dirichlet clamp
      d.val = fminf(fmaxf(lo + sc * tt, lo + a.eps * sc), hi - a.eps * sc);  // pyro AffineBeta.rsample clamp
    }
    d.p0 = mean;
    d.p1 = size;
    d.lo = lo;
    d.hi = hi;
  }
  return d;
}
TQ_HD void tq_site_store(const tq_cosmos_args& a, int site, const TqSiteDraw& d, const float* terms) {
  const int64_t NS = (int64_t)(1 + 4 * a.K) * tq_batch_units(a);
  a.lat[d.t] = d.val;
  // stored rows (tq_site.h: TQ_NSITE_STORED): log q, d lq/d alpha (d c1), implicit gradient; AffineBeta: + d lq/d c0, second gradient
  a.site[d.t] = terms[0];
  a.site[NS + d.t] = terms[2];
  a.site[2 * NS + d.t] = terms[4];
  if (site > a.K) {
    a.site[3 * NS + d.t] = terms[3];
    a.site[4 * NS + d.t] = terms[5];
  }
}
TQ_HD void tq_body_site(const tq_cosmos_args& a, int site, int64_t i) {
  const TqSiteDraw d = tq_site_draw(a, site, i);
  float terms[TQ_NSITE_TERMS];
  if (site <= a.K) tq_gamma_site_terms(d.val, d.p0, d.p1, terms);
  else tq_affine_beta_site_terms(d.val, d.p0, d.p1, d.lo, d.hi, a.eps, terms);
  tq_site_store(a, site, d, terms);
}

TQ_HD void tq_adam_apply(const tq_cosmos_args& a, int64_t j, float p, float dELBO);
template <bool STREAM = false>
TQ_HD void tq_adam_apply_given(const tq_cosmos_args& a, int64_t j, float p, float dELBO, float m_old, float v_old);

// ---- per-unit ELBO terms and gradients --------------------------------------------------------------------
// part[] receives this unit's contribution to the cross-unit sums (layout TQ_GS_*).
// aoi2 != nullptr: the unit's d/d(background_mean_loc, background_std_loc) partials are returned there instead of being
// written to a.aoi_part (the fused step kernel sums them per workgroup).
// LATE_MOMENTS: the Adam moments are loaded after the gradient arithmetic instead of ahead of it (the fused step kernel
// runs at twice the occupancy of tq_unit_kernel and cannot afford the 4K+... registers that holding them costs).
// pixv != nullptr: the pixel kernel's results for this unit (the rows of a.pix, in row order) are taken from there (the
// fused pixel + per-unit kernel hands them over in registers).
// STREAM: the Adam moments are read and written with non-temporal accesses (full-batch steps: touched once per step).
// COHERENT_AOI / Gp: the single-launch minibatch step, where the per-AOI parameters and the global tables are written by
// the tail workgroup of the same launch (tq_minibatch_kernel): device-scope loads, the caller's copy of the tables.
template <int K, bool LATE_MOMENTS = false, bool STREAM = false, bool COHERENT_AOI = false>
TQ_HD void tq_body_unit(const tq_cosmos_args& a, int64_t i, float* part, float* aoi2 = nullptr, const float* pixv = nullptr,
                        const TqGlobals* Gp = nullptr) {
  constexpr int M = 1 << K;
  constexpr int NL = TQ_NLOCAL(K);
  const int64_t B = tq_batch_units(a), U = tq_num_units(a);
  const TqUnitIdx ix = tq_decode_unit(a, i);
  const TqGlobals& G = Gp ? *Gp : *(const TqGlobals*)a.globals;
  TqSiteConsts C;
  C.H = 0.5f * (a.P + 1);
  C.eps = a.eps;
  C.w_lo = a.width_min; C.w_hi = a.width_max;
  C.height_std = a.height_std;
  C.bg_mean_std = a.background_mean_std; C.bg_std_std = a.background_std_std;

  TqUnitIn<K> in;
#pragma unroll
  for (int r = 0; r < NL; ++r) in.u[r] = a.params[(int64_t)r * U + ix.u];
  const int64_t ab = tq_aoi_base(a);
  const int64_t nc = (int64_t)ix.n * a.C + ix.c;
  in.u_bml = COHERENT_AOI ? TQ_LOAD_COHERENT(&a.params[ab + nc]) : a.params[ab + nc];
  in.u_bsl = COHERENT_AOI ? TQ_LOAD_COHERENT(&a.params[ab + (int64_t)a.Nt * a.C + nc]) : a.params[ab + (int64_t)a.Nt * a.C + nc];
  in.b = a.lat[i];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    in.h[k] = a.lat[(int64_t)(1 + k) * B + i];
    in.w[k] = a.lat[(int64_t)(1 + K + k) * B + i];
    in.x[k] = a.lat[(int64_t)(1 + 2 * K + k) * B + i];
    in.y[k] = a.lat[(int64_t)(1 + 3 * K + k) * B + i];
  }
#pragma unroll
  for (int mi = 0; mi < M; ++mi) in.ll[mi] = pixv ? pixv[mi] : a.pix[(int64_t)mi * B + i];
  in.gb = pixv ? pixv[M] : a.pix[(int64_t)M * B + i];
  const float g_gain = pixv ? pixv[M + 1] : a.pix[(int64_t)(M + 1) * B + i];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    in.gh[k] = pixv ? pixv[M + 2 + k] : a.pix[(int64_t)(M + 2 + k) * B + i];
    in.gw[k] = pixv ? pixv[M + 2 + K + k] : a.pix[(int64_t)(M + 2 + K + k) * B + i];
    in.gx[k] = pixv ? pixv[M + 2 + 2 * K + k] : a.pix[(int64_t)(M + 2 + 2 * K + k) * B + i];
    in.gy[k] = pixv ? pixv[M + 2 + 3 * K + k] : a.pix[(int64_t)(M + 2 + 3 * K + k) * B + i];
  }
  const int64_t NS = (int64_t)(1 + 4 * K) * B;
#pragma unroll
  for (int j = 0; j < TQ_NSITE_STORED; ++j) {
    const float* sj = a.site + (int64_t)j * NS;
    if (j < 3) in.sb[j] = sj[i];  // (Gamma sites: three stored terms)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (j < 3) in.sh[k][j] = sj[(int64_t)(1 + k) * B + i];
      in.sw[k][j] = sj[(int64_t)(1 + K + k) * B + i];
      in.sx[k][j] = sj[(int64_t)(1 + 2 * K + k) * B + i];
      in.sy[k][j] = sj[(int64_t)(1 + 3 * K + k) * B + i];
    }
  }
  const bool masked = a.aoi_mask && !a.aoi_mask[ix.n];
  in.wu = masked ? 0.0f : a.scale;
  in.on = a.is_ontarget[ix.n] ? 1 : 0;
  in.q = ix.c;

  // the Adam moments of the fused update are requested before the arithmetic below so that they arrive during it
  float m_old[NL], v_old[NL];
  if (a.fuse_adam && !LATE_MOMENTS) {
#pragma unroll
    for (int r = 0; r < NL; ++r) {
      m_old[r] = STREAM ? TQ_LOAD_STREAM(&a.exp_avg[(int64_t)r * U + ix.u]) : a.exp_avg[(int64_t)r * U + ix.u];
      v_old[r] = STREAM ? TQ_LOAD_STREAM(&a.exp_avg_sq[(int64_t)r * U + ix.u]) : a.exp_avg_sq[(int64_t)r * U + ix.u];
    }
  }

  TqUnitOut<K> out;
  tq_cosmos_unit<K>(in, G, C, &out);

  if (a.fuse_adam) {
    if (LATE_MOMENTS) {
#pragma unroll
      for (int r = 0; r < NL; ++r) {
        m_old[r] = a.exp_avg[(int64_t)r * U + ix.u];
        v_old[r] = a.exp_avg_sq[(int64_t)r * U + ix.u];
      }
    }
#pragma unroll
    for (int r = 0; r < NL; ++r)
      tq_adam_apply_given<STREAM>(a, (int64_t)r * U + ix.u, in.u[r], masked ? 0.0f : out.g[r], m_old[r], v_old[r]);
    if (a.last_step) a.last_step[ix.u] = (int32_t)(a.step + 1);  // lazy Adam clock of the unit
  } else {
#pragma unroll
    for (int r = 0; r < NL; ++r) a.grad[(int64_t)r * U + ix.u] = masked ? 0.0f : out.g[r];
  }
  if (aoi2) {
    aoi2[0] = masked ? 0.0f : out.g_bml;
    aoi2[1] = masked ? 0.0f : out.g_bsl;
  } else {
    a.aoi_part[i] = masked ? 0.0f : out.g_bml;
    a.aoi_part[B + i] = masked ? 0.0f : out.g_bsl;
  }

  // Every entry is assigned with compile-time indices and selected VALUES: indexed by the run-time channel (or written
  // under a run-time condition, which the compiler turns back into an indexed store) the caller's array went to scratch
  // memory -- 64 B per lane written and read back in the fused pixel + per-unit kernel.
  part[TQ_GS_GAIN] = masked ? 0.0f : g_gain;
  part[TQ_GS_CS] = masked ? 0.0f : out.d_cs;
  float elbo_u = out.elbo;
#pragma unroll
  for (int q = 0; q < TQ_MAXQ; ++q) {
    const bool mine = !masked && q == ix.c;
    part[TQ_GS_Q0 + 3 * q + 0] = mine ? out.d_rho : 0.0f;
    part[TQ_GS_Q0 + 3 * q + 1] = mine ? out.d_a : 0.0f;
    part[TQ_GS_Q0 + 3 * q + 2] = mine ? out.d_c : 0.0f;
  }
  if (a.crosstalk) {
    // rows after the cosmos block of pix: [ell_excess][g_alpha[q]] (tq_xtalk.h).  in.ll held the per-dye
    // MARGINAL likelihoods, so every dye's Dice sum contains E[ll]: remove the copies beyond the first
    const int64_t x0 = (int64_t)(M + 2 + 4 * K) * B;
    elbo_u -= in.wu * a.pix[x0 + i];
#pragma unroll
    for (int q = 0; q < 2; ++q) {  // (the crosstalk model is Q = C = 2: the alpha block follows the two dyes' entries)
      const float ga = a.pix[x0 + (int64_t)(1 + q) * B + i];
#pragma unroll
      for (int c = 0; c < 2; ++c) part[TQ_GS_ALPHA0(2) + q * 2 + c] = (!masked && c == ix.c) ? ga : 0.0f;
    }
  }
  part[TQ_GS_ELBO] = masked ? 0.0f : elbo_u;
}

// ---- per-AOI: finish d/d(background_mean_loc, background_std_loc) given the frame sums --------------------
TQ_HD void tq_body_aoi_finish(const tq_cosmos_args& a, int ai, int c, float sum_bml, float sum_bsl, float* elbo) {
  const int n = a.ndx ? a.ndx[ai] : ai;
  const bool masked = a.aoi_mask && !a.aoi_mask[n];
  TqSiteConsts C;
  C.bg_mean_std = a.background_mean_std; C.bg_std_std = a.background_std_std;
  const int64_t ab = tq_aoi_base(a), nc = (int64_t)n * a.C + c, NC = (int64_t)a.Nt * a.C;
  float e = 0.0f, g1 = 0.0f, g2 = 0.0f;
  if (!masked) tq_cosmos_aoi(a.params[ab + nc], a.params[ab + NC + nc], a.scale_n, C, &e, &g1, &g2);
  a.grad[ab + nc] = masked ? 0.0f : (sum_bml + g1);
  a.grad[ab + NC + nc] = masked ? 0.0f : (sum_bsl + g2);
  *elbo = e;
}

// ---- global sites: gradient of site s; returns the site's part of the ELBO ---------------------------------
TQ_HD double tq_body_globals_grad(const tq_cosmos_args& a, int s) {
  const TqGlobalConsts C = tq_global_consts(a);
  TqGlobalSite p;
  tq_globals_constrain_site(a.params + tq_global_base(a), C, s, &p);
  return tq_globals_grad_site(s, p, *(const TqGlobalBase*)a.gbase, *(const TqGlobals*)a.globals, C, a.gsum,
                              a.grad + tq_global_base(a));
}

// ---- Adam on one element (torch.optim.Adam, no amsgrad / weight decay; minimises -ELBO) ------------------
// p = current parameter value, dELBO = d ELBO / d param
TQ_HD void tq_adam_apply(const tq_cosmos_args& a, int64_t j, float p, float dELBO) {
  tq_adam_apply_given(a, j, p, dELBO, a.exp_avg[j], a.exp_avg_sq[j]);
}
template <bool STREAM>
TQ_HD void tq_adam_apply_given(const tq_cosmos_args& a, int64_t j, float p, float dELBO, float m_old, float v_old) {
  const float g = -dELBO;
  const float m = a.beta1 * m_old + (1.0f - a.beta1) * g;
  const float v = a.beta2 * v_old + (1.0f - a.beta2) * g * g;
  if (STREAM) {
    TQ_STORE_STREAM(m, &a.exp_avg[j]);
    TQ_STORE_STREAM(v, &a.exp_avg_sq[j]);
  } else {
    a.exp_avg[j] = m;
    a.exp_avg_sq[j] = v;
  }
  // p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps) with the 1-ulp hardware sqrt / rcp (the IEEE division and
  // square-root sequences cost ~70 instructions per parameter; the update is accurate to ~1e-7 of lr either way)
  const float rs2 = TQ_FRCP(TQ_FSQRT(a.bias_correction2));
  const float lr1 = a.lr * TQ_FRCP(a.bias_correction1);
  a.params[j] = p - lr1 * m * TQ_FRCP(TQ_FSQRT(v) * rs2 + a.adam_eps);
}
// beta^n for an integer n >= 0 by binary exponentiation in double (about 2 log2 n dependent multiplications; libm's pow
// is several hundred instructions and sat on the critical path of every minibatch step).  Agrees with pow to ~1e-16
// relative, far inside what 1 - beta^n needs.
TQ_HD double tq_powi(double base, int n) {
  double r = 1.0;
  while (n > 0) {
    if (n & 1) r *= base;
    base *= base;
    n >>= 1;
  }
  return r;
}

// Replay of the zero-gradient Adam steps s0..s1 of element j (lazy Adam of minibatch fits: the steps in which the
// element's unit was not in the minibatch).  Same arithmetic as tq_adam_apply_given with g = 0; the bias corrections
// 1 - beta^s are formed in double like the host's, with beta^s carried by multiplication.
TQ_HD void tq_adam_replay(const tq_cosmos_args& a, int64_t j, int s0, int s1) {
  if (s0 > s1) return;
  float p = a.params[j], m = a.exp_avg[j], v = a.exp_avg_sq[j];
  double pw1 = tq_powi(a.beta1_d, s0), pw2 = tq_powi(a.beta2_d, s0);
  int s = s0;
  // With zero gradient the increment shrinks by ~beta1 / sqrt(beta2) per step.  Once it is below a quarter of an ulp of
  // p the remaining steps leave p where it is -- in the dense kernel too, which performs the same fp32 subtraction --
  // and only scale the moments: the worst unit of a minibatch has missed ~Nt F / (nb fb) * ln(nb fb) steps (~700 at the
  // default 10 x 512 of 400 x 1000), the increments die after ~120.  Parameters at or near zero (the initial x_mean,
  // y_mean, m_probs logits) have no ulp to hide behind: there the tail of the geometric series that is cut off,
  // < 10 x 3e-9 in an unconstrained parameter, is below the rounding of the fp32 sigmoid / exp it feeds.
  // The test is made once per 8 steps: a branch on the increment in every step serialises the steps (~300 cycles
  // each for a lone wave instead of ~40 when consecutive steps overlap).
  bool live = true;
  while (live && s + 8 <= s1 + 1) {
    float inc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // unrolled: the 8 increments are independent of each other, only p chains
      m = a.beta1 * m;
      v = a.beta2 * v;
      const float rs2 = TQ_FRCP(TQ_FSQRT((float)(1.0 - pw2)));
      const float lr1 = a.lr * TQ_FRCP((float)(1.0 - pw1));
      inc = lr1 * m * TQ_FRCP(TQ_FSQRT(v) * rs2 + a.adam_eps);
      p = p - inc;
      pw1 *= a.beta1_d;
      pw2 *= a.beta2_d;
    }
    s += 8;
    live = !(fabsf(inc) < fmaxf(1.4901161e-08f * fabsf(p), 3e-9f));
  }
  if (live) {
    for (; s <= s1; ++s) {  // fewer than 8 steps left
      m = a.beta1 * m;
      v = a.beta2 * v;
      const float rs2 = TQ_FRCP(TQ_FSQRT((float)(1.0 - pw2)));
      const float lr1 = a.lr * TQ_FRCP((float)(1.0 - pw1));
      p = p - lr1 * m * TQ_FRCP(TQ_FSQRT(v) * rs2 + a.adam_eps);
      pw1 *= a.beta1_d;
      pw2 *= a.beta2_d;
    }
  }
  if (s <= s1) {  // steps s .. s1: moments only (closed form of the repeated multiplication)
    const int n = s1 - s + 1;
    m *= (float)tq_powi((double)a.beta1, n);
    v *= (float)tq_powi((double)a.beta2, n);
  }
  a.params[j] = p;
  a.exp_avg[j] = m;
  a.exp_avg_sq[j] = v;
}

// The same replay with the per-step factors -- lr / (1 - beta1^s) and 1 / sqrt(1 - beta2^s), which depend on the step
// only -- read from a table tab[2 (s - T0)], tab[2 (s - T0) + 1] for s >= T0 (tq_adam_bias_table; the single-launch
// minibatch step keeps it in LDS).  Removes the fp64 power chain, two conversions and two of the four transcendentals
// from every replayed step; steps before T0 (units that sat out more than the table's length) take the direct form.
#define TQ_BIAS_TABLE_STEPS 1024
TQ_HD void tq_adam_bias_entry(const tq_cosmos_args& a, int s, float* lr1, float* rs2) {
  *rs2 = TQ_FRCP(TQ_FSQRT((float)(1.0 - tq_powi(a.beta2_d, s))));
  *lr1 = a.lr * TQ_FRCP((float)(1.0 - tq_powi(a.beta1_d, s)));
}
TQ_HD void tq_adam_replay_tab_given(const tq_cosmos_args& a, int64_t j, int s0, int s1, const float* tab, int T0,
                                    float p, float m, float v);
TQ_HD void tq_adam_replay_tab(const tq_cosmos_args& a, int64_t j, int s0, int s1, const float* tab, int T0) {
  if (s0 > s1) return;
  tq_adam_replay_tab_given(a, j, s0, s1, tab, T0, a.params[j], a.exp_avg[j], a.exp_avg_sq[j]);
}
// (p, m, v: the element's current values, loaded by the caller ahead of time)
TQ_HD void tq_adam_replay_tab_given(const tq_cosmos_args& a, int64_t j, int s0, int s1, const float* tab, int T0,
                                    float p, float m, float v) {
  if (s0 > s1) return;
  int s = s0;
  for (; s < T0 && s <= s1; ++s) {
    float lr1, rs2;
    tq_adam_bias_entry(a, s, &lr1, &rs2);
    m = a.beta1 * m;
    v = a.beta2 * v;
    p = p - lr1 * m * TQ_FRCP(TQ_FSQRT(v) * rs2 + a.adam_eps);
  }
  bool live = true;
  while (live && s + 8 <= s1 + 1) {
    float inc = 0.0f;
    const float* t = tab + 2 * (s - T0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      m = a.beta1 * m;
      v = a.beta2 * v;
      inc = t[2 * k] * m * TQ_FRCP(TQ_FSQRT(v) * t[2 * k + 1] + a.adam_eps);
      p = p - inc;
    }
    s += 8;
    live = !(fabsf(inc) < fmaxf(1.4901161e-08f * fabsf(p), 3e-9f));
  }
  if (live) {
    for (; s <= s1; ++s) {
      const float* t = tab + 2 * (s - T0);
      m = a.beta1 * m;
      v = a.beta2 * v;
      p = p - t[0] * m * TQ_FRCP(TQ_FSQRT(v) * t[1] + a.adam_eps);
    }
  }
  if (s <= s1) {
    const int n = s1 - s + 1;
    m *= (float)tq_powi((double)a.beta1, n);
    v *= (float)tq_powi((double)a.beta2, n);
  }
  a.params[j] = p;
  a.exp_avg[j] = m;
  a.exp_avg_sq[j] = v;
}

TQ_HD void tq_body_adam(const tq_cosmos_args& a, int64_t j) {
  tq_adam_apply(a, j, a.params[j], a.grad[j]);
  if (a.zero_grad) a.grad[j] = 0.0f;
}

// ---- posterior read-out -----------------------------------------------------------------------------------
TQ_HD void tq_body_probs_globals(const tq_probs_args& a, int s, int particle) {
  TqGlobalConsts C;
  C.K = a.K; C.P = a.P; C.Q = a.C; C.eps = a.eps;
  C.xt = 0;  // alpha does not enter the z / theta posterior
  C.gain_std = C.lamda_rate = C.proximity_rate = 1.0;  // priors are not used for draws
  const int64_t U = (int64_t)a.Nt * a.F * a.C;
  const float* ug = a.params + (int64_t)TQ_NLOCAL(a.K) * U + 2 * (int64_t)a.Nt * a.C;
  TqGlobalSite p;
  tq_globals_constrain_site(ug, C, s, &p);
  tq_globals_sample_site(s, p, C, a.seed, 0x40000000u + (uint32_t)particle, a.draw,
                         (TqGlobalBase*)a.gbase_p + particle, (TqGlobals*)a.globals_p + particle);
}

template <int K>
TQ_HD void tq_body_probs_unit(const tq_probs_args& a, int64_t u) {
  const int64_t U = (int64_t)a.Nt * a.F * a.C;
  const int c = (int)((uint32_t)u % (uint32_t)a.C);  // U < 2^31 (host-checked)
  const int n = (int)((uint32_t)u / ((uint32_t)a.F * (uint32_t)a.C));
  float z1 = 0.0f, th[K];
#pragma unroll
  for (int k = 0; k < K; ++k) th[k] = 0.0f;
  if (a.is_ontarget[n]) {  // cosmos.py:615-623: only on-target AOIs are evaluated
    const float H = 0.5f * (a.P + 1);
    float um[K], xm[K], ym[K], sz[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      um[k] = a.params[(int64_t)TQ_ROW(TQ_P_MPROBS, k, K) * U + u];
      const float lo = -H + a.eps, hi = H - a.eps;
      xm[k] = lo + (hi - lo) * tq_sigmoid(a.params[(int64_t)TQ_ROW(TQ_P_XMEAN, k, K) * U + u]);
      ym[k] = lo + (hi - lo) * tq_sigmoid(a.params[(int64_t)TQ_ROW(TQ_P_YMEAN, k, K) * U + u]);
      sz[k] = 2.0f + expf(a.params[(int64_t)TQ_ROW(TQ_P_SIZE, k, K) * U + u]);
    }
    for (int p = 0; p < a.particles; ++p) {
      float x[K], y[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (a.draw) {
#pragma nounroll
          for (int ax = 0; ax < 2; ++ax) {
            TqPhilox s;
            tq_philox_init(&s, a.seed, 0x40000000u + (uint32_t)p, 0x800u + 2 * k + ax, (uint64_t)u);
            const float mean = ax ? ym[k] : xm[k];
            const float c1 = sz[k] * (mean + H) / (2.0f * H), c0 = sz[k] - c1;
            const float g1 = tq_sample_std_gamma(&s, c1), g0 = tq_sample_std_gamma(&s, c0);
            float tt = fminf(fmaxf(g1 / (g1 + g0), 1.17549435e-38f), 1.0f - 5.96046448e-08f);
            const float v = fminf(fmaxf(-H + 2.0f * H * tt, -H + a.eps * 2.0f * H), H - a.eps * 2.0f * H);
            if (ax) y[k] = v; else x[k] = v;
          }
        } else {
          x[k] = a.xy_given[((int64_t)p * 2 * K + k) * U + u];
          y[k] = a.xy_given[((int64_t)p * 2 * K + K + k) * U + u];
        }
      }
      float R[K];
      tq_zt_responsibilities<K>(x, y, um, ((const TqGlobals*)a.globals_p)[p], c, H, R);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        th[k] += R[k];
        z1 += R[k];
      }
    }
    const float inv = 1.0f / (float)a.particles;
    z1 *= inv;
#pragma unroll
    for (int k = 0; k < K; ++k) th[k] *= inv;
    a.z_probs[2 * u] = 1.0f - z1;
    a.z_probs[2 * u + 1] = z1;
  } else {
    a.z_probs[2 * u] = 0.0f;
    a.z_probs[2 * u + 1] = 0.0f;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) a.theta_probs[(int64_t)k * U + u] = th[k];
}
