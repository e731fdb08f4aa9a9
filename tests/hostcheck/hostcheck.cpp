// hostcheck.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the host+device inline math of tapqir_amd/csrc (tq_math.h, tq_pixel.h, tq_site.h,
// tq_globals.h, tq_bodies.h) with g++ and drives it with plain loops on HOST memory, so the
// CPU test-suite can check the hand-derived gradients against the oracle without a GPU.
// The product (tapqir_amd/_lib.py) never loads this library; there is no CPU fallback.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../tapqir_amd/csrc/tq_bodies.h"
#include "../../tapqir_amd/csrc/tq_pixel.h"
#include "../../tapqir_amd/csrc/tq_aux.h"
#include "../../tapqir_amd/csrc/tq_xtalk.h"

extern "C" {

void hc_lgamma_digamma(const float* a, float* lg, float* dg, int64_t n) {
  for (int64_t i = 0; i < n; ++i) tq_lgamma_digamma(a[i], &lg[i], &dg[i]);
}
void hc_std_gamma_grad(const float* alpha, const float* x, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = tq_std_gamma_grad(alpha[i], x[i]);
}
void hc_dirichlet_grad(const float* x, const float* alpha, const float* total, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = tq_dirichlet_grad(x[i], alpha[i], total[i]);
}
// both Beta implicit gradients through the fused saddle-point path (falls back to the generic one)
void hc_beta_grad_pair(const float* x, const float* c1, const float* c0, float* ga, float* gb, uint8_t* fused, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    double a, b;
    const float total = c1[i] + c0[i];
    fused[i] = tq_beta_grad_pair_mid((double)x[i], (double)c1[i], (double)total - (double)c1[i], &a, &b);
    if (fused[i]) { ga[i] = (float)a; gb[i] = (float)b; }
    else { ga[i] = tq_dirichlet_grad(x[i], c1[i], total); gb[i] = tq_dirichlet_grad(1.0f - x[i], c0[i], total); }
  }
}
// the regime-ordered evaluation of both directions (tq_beta_grad_pair_rest) next to the two plain calls it replaces
void hc_beta_grad_pair_rest(const float* x, const float* c1, const float* c0, float* ga, float* gb, float* ra, float* rb, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    const float total = c1[i] + c0[i];
    float dd[2];
    tq_beta_grad_pair_rest(x[i], c1[i], c0[i], total, dd);
    ga[i] = dd[0];
    gb[i] = dd[1];
    ra[i] = tq_dirichlet_grad(x[i], c1[i], total);
    rb[i] = tq_dirichlet_grad(1.0f - x[i], c0[i], total);
  }
}
void hc_sample_std_gamma(uint64_t seed, uint32_t step, uint32_t site, const float* alpha, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    TqPhilox s;
    tq_philox_init(&s, seed, step, site, (uint64_t)i);
    out[i] = tq_sample_std_gamma(&s, alpha[i]);
  }
}
void hc_philox(uint64_t seed, uint32_t step, uint32_t site, uint64_t elem, uint32_t* out, int n) {
  TqPhilox s;
  tq_philox_init(&s, seed, step, site, elem);
  for (int i = 0; i < n; ++i) out[i] = tq_philox_next(&s);
}
void hc_mean_frac(double lam, int K, double* val, double* dval) { tq_mean_frac(lam, K, val, dval); }

}  // extern "C"

// Host emulation of the pixel kernels (same per-pixel functions and the same algebra for the
// single-offset assembly, the moments and the gain term; plain loops, double accumulators).
template <int K>
static void ksmogn_host(const tq_ksmogn_args& a) {
  constexpr int M = 1 << K;
  const int64_t B = (int64_t)a.nb * a.fb * a.C;
  const int P = a.P, npix = P * P;
  const bool bwd = a.g_background != nullptr;
  const bool one = a.O == 1 && a.pixstats != nullptr;
  const float g = a.gain[0], rg = 1.0f / g, ln_g = logf(g);
  TqOffsetInfo hinfo;
  tq_offset_info(a.offset_samples, a.offset_logits, a.O, &hinfo);
  for (int64_t i = 0; i < B; ++i) {
    const int c = (int)(i % a.C);
    const int64_t ab = i / a.C;
    const int bi = (int)(ab % a.fb), ai = (int)(ab / a.fb);
    const int n = a.ndx ? a.ndx[ai] : ai, f = a.fdx ? a.fdx[bi] : bi;
    const int64_t u = ((int64_t)n * a.F + f) * a.C + c;
    const float tx = a.xy[2 * u], ty = a.xy[2 * u + 1], b = a.background[i];
    float hk[K], wk[K], amph[K], cx[K], cy[K], inv2v[K];
    for (int k = 0; k < K; ++k) {
      hk[k] = a.height[k * B + i];
      wk[k] = a.width[k * B + i];
      cx[k] = a.x[k * B + i] + tx;
      cy[k] = a.y[k * B + i] + ty;
      inv2v[k] = 0.5f / (wk[k] * wk[k]);
      amph[k] = hk[k] * inv2v[k] * (1.0f / TQ_PI);
    }
    float W[M];
    for (int mi = 0; mi < M; ++mi) W[mi] = 0.0f;
    if (bwd) {
      if (a.gout) {
        for (int mi = 0; mi < M; ++mi) W[mi] = a.gout[(int64_t)mi * B + i];
      } else {
        const float sc = a.scale * ((a.aoi_mask == nullptr || a.aoi_mask[n]) ? 1.0f : 0.0f);
        for (int mi = 0; mi < M; ++mi) {
          float w = sc;
          for (int k = 0; k < K; ++k) {
            const float uk = a.m_logit[k * a.m_kstride + u];
            w *= ((mi >> k) & 1) ? tq_sigmoid(uk) : tq_sigmoid(-uk);
          }
          W[mi] = w;
        }
      }
    }
    const bool fast = b * rg >= TQ_FAST_ALPHA;  // device: wave-uniform __all() of the same test
    double ll[M] = {0}, sl[M] = {0}, sS[M] = {0}, acc_b = 0, acc_g = 0;
    double S0[K] = {0}, Sx[K] = {0}, Sy[K] = {0}, Sr[K] = {0}, SN[K] = {0};
    bool bad = false;
    float S_v = 0, S_lv = 0;
    if (one) {
      S_v = a.pixstats[u];
      S_lv = a.pixstats[a.stats_stride + u];
      bad = a.pixstats[2 * a.stats_stride + u] > 0.0f;
    }
    for (int pix = 0; pix < npix; ++pix) {
      const int j = pix / P, ic = pix % P;
      const float D = a.images[u * npix + pix];
      const float fic = (float)ic, fj = (float)j;
      float spot[K];
      for (int k = 0; k < K; ++k) {
        const float dx = fic - cx[k], dy = fj - cy[k];
        spot[k] = amph[k] * expf(-dx * dx * inv2v[k]) * expf(-dy * dy * inv2v[k]);
      }
      float da[M], gq[M];
      for (int mi = 0; mi < M; ++mi) { da[mi] = 0; gq[mi] = 0; }
      if (one) {
        const float v = bad ? fmaxf(D - a.offset_samples[0], 1.0f) : D - a.offset_samples[0];
        for (int mi = 1; mi < M; ++mi) {
          float mu = b, l2, S;
          for (int k = 0; k < K; ++k)
            if ((mi >> k) & 1) mu += spot[k];
          if (fast) tq_pix_one_offset<true>(v, mu, g, rg, ln_g, &l2, &S, &da[mi]);
          else tq_pix_one_offset<false>(v, mu, g, rg, ln_g, &l2, &S, &da[mi]);
          ll[mi] += (double)mu * l2;
          sl[mi] += l2;
          sS[mi] += S;
        }
        for (int k = 0; k < K; ++k) SN[k] += spot[k];
      } else {
        float mu[M], lp[M];
        for (int mi = 0; mi < M; ++mi) {
          mu[mi] = b;
          for (int k = 0; k < K; ++k)
            if ((mi >> k) & 1) mu[mi] += spot[k];
        }
        if (fast) tq_pix_multi_offset<M, true, true>(D, mu, a.offset_samples, a.offset_logits, a.O, hinfo, g, rg, ln_g, lp, da, gq);
        else tq_pix_multi_offset<M, true, false>(D, mu, a.offset_samples, a.offset_logits, a.O, hinfo, g, rg, ln_g, lp, da, gq);
        for (int mi = 0; mi < M; ++mi) ll[mi] += lp[mi];
      }
      if (bwd) {
        float q[K];
        for (int k = 0; k < K; ++k) q[k] = 0;
        for (int mi = one ? 1 : 0; mi < M; ++mi) {
          const float cw = W[mi] * da[mi];
          acc_b += cw;
          acc_g += W[mi] * gq[mi];
          for (int k = 0; k < K; ++k)
            if ((mi >> k) & 1) q[k] += cw;
        }
        const float c0 = 0.5f * (float)(P - 1);  // moments about the tile centre, as the device kernels take them
        const float fx = fic - c0, fy = fj - c0;
        const float r2 = fx * fx + fy * fy;
        for (int k = 0; k < K; ++k) {
          const float aq = q[k] * spot[k];
          S0[k] += aq;
          Sx[k] += aq * fx;
          Sy[k] += aq * fy;
          Sr[k] += aq * r2;
        }
      }
    }
    const double fn = npix;
    if (one) {  // tq_pixel_assemble_one_offset
      TqCombo0 c0;
      tq_combo0_prepare(b, rg, g, ln_g, &c0);
      const double lw0 = a.offset_logits[0] - TQ_LN_SQRT_2PI;
      const double common = lw0 * fn - S_lv, S_lvg = S_lv - fn * ln_g, sl0 = S_lv - fn * c0.lnb, mmv = b * fn - S_v;
      ll[0] = common + rg * (b * sl0 + mmv) + 0.5 * (S_lvg - sl0) - fn * c0.S;
      for (int mi = 1; mi < M; ++mi) {
        double sn = 0;
        for (int k = 0; k < K; ++k)
          if ((mi >> k) & 1) sn += SN[k];
        ll[mi] = common + rg * (TQ_LN2 * ll[mi] + mmv + sn) + 0.5 * (S_lvg - TQ_LN2 * sl[mi]) - sS[mi];
      }
      if (bwd) acc_b += W[0] * (sl0 + fn * c0.c_da);
    }
    for (int mi = 0; mi < M; ++mi) a.ll[(int64_t)mi * B + i] = bad ? -INFINITY : (float)ll[mi];
    if (bwd) {
      const double z = (bad || ll[0] == -INFINITY) ? 0.0 : 1.0;
      a.g_background[i] = (float)(z * acc_b * rg);
      if (one) {
        double Wsum = 0, mu_da = b * acc_b, mu_w = 0;
        for (int mi = 0; mi < M; ++mi) Wsum += W[mi];
        for (int k = 0; k < K; ++k) {
          double Wk = 0;
          for (int mi = 0; mi < M; ++mi)
            if ((mi >> k) & 1) Wk += W[mi];
          mu_da += S0[k];
          mu_w += Wk * SN[k];
        }
        acc_g = rg * (mu_da + mu_w + Wsum * (b * fn - S_v));
      }
      a.g_gain[i] = (float)(-z * acc_g * rg);
      for (int k = 0; k < K; ++k) {
        const double rw = 1.0 / wk[k];
        const double c0 = 0.5 * (P - 1), ccx = cx[k] - c0, ccy = cy[k] - c0;
        const double S1x = Sx[k] - ccx * S0[k], S1y = Sy[k] - ccy * S0[k];
        const double S2 = Sr[k] - 2.0 * (ccx * Sx[k] + ccy * Sy[k]) + (ccx * ccx + ccy * ccy) * S0[k];
        a.g_height[k * B + i] = (float)(z * S0[k] * rg / hk[k]);
        a.g_x[k * B + i] = (float)(z * rg * S1x * rw * rw);
        a.g_y[k * B + i] = (float)(z * rg * S1y * rw * rw);
        a.g_width[k * B + i] = (float)(z * rg * (S2 * rw * rw * rw - 2.0 * S0[k] * rw));
      }
    }
  }
}

extern "C" void hc_image_stats(const float* images, const float* offset, float* out, int64_t U, int32_t P) {
  const int npix = P * P;
  for (int64_t u = 0; u < U; ++u) {
    double sv = 0, slv = 0;
    float nbad = 0;
    for (int p = 0; p < npix; ++p) {
      const float v = images[u * npix + p] - offset[0];
      if (v > 0.0f) { sv += v; slv += log((double)v); } else nbad += 1;
    }
    out[u] = (float)sv; out[U + u] = (float)slv; out[2 * U + u] = nbad;
  }
}

extern "C" {

int hc_ksmogn_log_prob(const tq_ksmogn_args* a) {
  switch (a->K) {
    case 1: ksmogn_host<1>(*a); break;
    case 2: ksmogn_host<2>(*a); break;
    case 3: ksmogn_host<3>(*a); break;
    default: ksmogn_host<4>(*a); break;
  }
  return 0;
}

}  // extern "C"

template <int K>
static void xtalk_host(const tq_xtalk_args& a) {
  const bool bwd = a.g_background != nullptr;
  const int64_t Bg = (int64_t)a.nb * a.fb;
  const float g = a.gain[0], rg = 1.0f / g, ln_g = logf(g);
  TqOffsetInfo h;
  tq_offset_info(a.offset_samples, a.offset_logits, a.O, &h);
  for (int64_t gi = 0; gi < Bg; ++gi) {
    TqXtGroup<K> G;
    tq_xtalk_load_group<K>(a, gi, bwd, &G);
    TqXtAcc<K> A;
    tq_xt_acc_zero<K>(A);
    const bool fast = fminf(G.b[0], G.b[1]) * rg >= TQ_FAST_ALPHA;
    if (bwd) {
      if (fast) tq_xtalk_pixels<K, true, true>(a, G, h, 0, 1, g, rg, ln_g, A);
      else tq_xtalk_pixels<K, true, false>(a, G, h, 0, 1, g, rg, ln_g, A);
      tq_xtalk_finish<K, true>(a, gi, G, A, rg);
    } else {
      if (fast) tq_xtalk_pixels<K, false, true>(a, G, h, 0, 1, g, rg, ln_g, A);
      else tq_xtalk_pixels<K, false, false>(a, G, h, 0, 1, g, rg, ln_g, A);
      tq_xtalk_finish<K, false>(a, gi, G, A, rg);
    }
  }
}

extern "C" {

int hc_ksmogn_crosstalk_log_prob(const tq_xtalk_args* a) {
  if (a->K == 1) xtalk_host<1>(*a);
  else xtalk_host<2>(*a);
  return 0;
}

int64_t hc_globals_size(void) { return (int64_t)sizeof(TqGlobals); }
int64_t hc_gbase_size(void) { return (int64_t)sizeof(TqGlobalBase); }

void hc_cosmos_sample_globals(const tq_cosmos_args* a) {
  for (int s = 0; s < tq_num_gsites(*a); ++s) tq_body_sample_globals(*a, s);
}
void hc_cosmos_sample_locals(const tq_cosmos_args* a) {
  const int64_t B = tq_batch_units(*a);
  for (int site = 0; site < 1 + 4 * a->K; ++site)
    for (int64_t i = 0; i < B; ++i) tq_body_site(*a, site, i);
}

}  // extern "C"

template <int K>
static void units_host(const tq_cosmos_args& a, double* sums) {
  const int64_t B = tq_batch_units(a);
  const int nq = tq_num_gsum(a);
  float part[3 + 3 * TQ_MAXQ];
  for (int64_t i = 0; i < B; ++i) {
    tq_body_unit<K>(a, i, part);
    for (int j = 0; j < nq; ++j) sums[j] += part[j];
  }
}

extern "C" {

void hc_cosmos_elbo_grads(const tq_cosmos_args* a) {
  const int K = a->K, M = 1 << K;
  const int64_t B = tq_batch_units(*a), U = tq_num_units(*a);
  tq_ksmogn_args k;
  memset(&k, 0, sizeof(k));
  k.images = a->images; k.images_il = nullptr; k.xy = a->xy; k.ndx = a->ndx; k.fdx = a->fdx;
  k.pixstats = a->pixstats; k.stats_stride = U;
  k.background = a->lat;
  k.height = a->lat + (int64_t)1 * B;
  k.width = a->lat + (int64_t)(1 + K) * B;
  k.x = a->lat + (int64_t)(1 + 2 * K) * B;
  k.y = a->lat + (int64_t)(1 + 3 * K) * B;
  k.gain = &((const TqGlobals*)a->globals)->gain;
  k.offset_samples = a->offset_samples; k.offset_logits = a->offset_logits;
  k.m_logit = a->params; k.m_kstride = U; k.aoi_mask = a->aoi_mask;
  k.ll = a->pix;
  k.g_background = a->pix + (int64_t)M * B;
  k.g_gain = a->pix + (int64_t)(M + 1) * B;
  k.g_height = a->pix + (int64_t)(M + 2) * B;
  k.g_width = a->pix + (int64_t)(M + 2 + K) * B;
  k.g_x = a->pix + (int64_t)(M + 2 + 2 * K) * B;
  k.g_y = a->pix + (int64_t)(M + 2 + 3 * K) * B;
  k.nb = a->nb; k.fb = a->fb; k.C = a->C; k.F = a->F; k.P = a->P; k.K = K; k.O = a->O;
  k.scale = a->scale;
  if (a->crosstalk) {
    tq_xtalk_args x;
    memset(&x, 0, sizeof(x));
    x.images = a->images; x.xy = a->xy; x.ndx = a->ndx; x.fdx = a->fdx;
    x.nb_full = a->Nt;
    x.background = k.background; x.height = k.height; x.width = k.width; x.x = k.x; x.y = k.y;
    x.gain = k.gain;
    x.alpha = &((const TqGlobals*)a->globals)->alpha[0][0];
    x.offset_samples = a->offset_samples; x.offset_logits = a->offset_logits;
    x.m_logit = a->params; x.m_kstride = U; x.aoi_mask = a->aoi_mask;
    x.ll = a->pix;
    x.ell_excess = a->pix + (int64_t)(M + 2 + 4 * K) * B;
    x.g_alpha = a->pix + (int64_t)(M + 3 + 4 * K) * B;
    x.g_background = k.g_background; x.g_gain = k.g_gain;
    x.g_height = k.g_height; x.g_width = k.g_width; x.g_x = k.g_x; x.g_y = k.g_y;
    x.nb = a->nb; x.fb = a->fb; x.C = a->C; x.F = a->F; x.P = a->P; x.K = K; x.O = a->O;
    x.scale = a->scale;
    hc_ksmogn_crosstalk_log_prob(&x);
  } else {
    hc_ksmogn_log_prob(&k);
  }
  const int nq = tq_num_gsum(*a);
  std::vector<double> sums(nq, 0.0);
  switch (K) {
    case 1: units_host<1>(*a, sums.data()); break;
    case 2: units_host<2>(*a, sums.data()); break;
    case 3: units_host<3>(*a, sums.data()); break;
    default: units_host<4>(*a, sums.data()); break;
  }
  for (int ai = 0; ai < a->nb; ++ai)
    for (int c = 0; c < a->C; ++c) {
      double s1 = 0, s2 = 0;
      for (int b = 0; b < a->fb; ++b) {
        const int64_t i = ((int64_t)ai * a->fb + b) * a->C + c;
        s1 += a->aoi_part[i];
        s2 += a->aoi_part[B + i];
      }
      float e;
      tq_body_aoi_finish(*a, ai, c, (float)s1, (float)s2, &e);
      sums[TQ_GS_ELBO] += e;
    }
  for (int j = 0; j < nq; ++j) a->gsum[j] = sums[j];
}

void hc_cosmos_globals_grad(const tq_cosmos_args* a) {
  double eg = 0.0;
  for (int s = 0; s < tq_num_gsites(*a); ++s) eg += tq_body_globals_grad(*a, s);
  a->elbo_out[0] = a->gsum[TQ_GS_ELBO] + (double)a->global_weight * eg;
}
void hc_cosmos_adam(const tq_cosmos_args* a);
void hc_cosmos_sample_globals(const tq_cosmos_args* a);
// everything after the all-reduce of gsum (+ the next step's global draws): tq_cosmos_tail_reduced
void hc_cosmos_tail_reduced(const tq_cosmos_args* a, const tq_cosmos_args* next) {
  hc_cosmos_globals_grad(a);
  hc_cosmos_adam(a);
  if (next) hc_cosmos_sample_globals(next);
}
void hc_cosmos_adam(const tq_cosmos_args* a) {
  const int64_t total = tq_num_params(*a);
  for (int64_t j = a->fuse_adam ? tq_aoi_base(*a) : 0; j < total; ++j) tq_body_adam(*a, j);
}
// tq_cosmos_adam_catchup: the loop of tq_adam_catchup_kernel (lazy Adam of minibatch steps)
void hc_cosmos_adam_catchup(const tq_cosmos_args* a, int32_t all_units) {
  const int64_t n = all_units ? tq_num_units(*a) : tq_batch_units(*a), U = tq_num_units(*a);
  for (int r = 0; r < TQ_NLOCAL(a->K); ++r)
    for (int64_t i = 0; i < n; ++i) {
      const int64_t u = all_units ? i : tq_decode_unit(*a, i).u;
      tq_adam_replay(*a, (int64_t)r * U + u, a->last_step[u] + 1, (int)a->step);
    }
}

}  // extern "C"

template <int K>
static void probs_host(const tq_probs_args& a) {
  const int64_t U = (int64_t)a.Nt * a.F * a.C;
  for (int64_t u = 0; u < U; ++u) tq_body_probs_unit<K>(a, u);
}
extern "C" void hc_cosmos_probs(const tq_probs_args* a) {
  for (int p = 0; p < a->particles; ++p)
    for (int s = 0; s < TQ_NGSITES(a->C); ++s) tq_body_probs_globals(*a, s, p);
  switch (a->K) {
    case 1: probs_host<1>(*a); break;
    case 2: probs_host<2>(*a); break;
    case 3: probs_host<3>(*a); break;
    default: probs_host<4>(*a); break;
  }
}

// ---- off-step bodies (tq_aux.h) ------------------------------------------------------------------------------------
extern "C" void hc_ksmogn_rsample(const tq_rsample_args* a) {
  const int npix = a->P * a->P;
  for (int64_t i = 0; i < a->B; ++i)
    for (int p = 0; p < npix; ++p) tq_body_rsample(*a, i, p);
}
extern "C" void hc_snr_chi2(const tq_snr_args* a) {
  for (int64_t u = 0; u < a->U; ++u) tq_body_snr_chi2(*a, u);
}
