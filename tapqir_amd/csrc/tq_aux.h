// tq_aux.h -- the two off-step users of the spot render (host+device inline bodies):
//   * KSMOGN.rsample (tapqir/distributions/ksmogn.py:171-185): one pixel of a simulated image;
//   * snr_and_chi2 (tapqir/utils/stats.py:29-86): signal-to-noise ratio of each spot and chi2 of the fitted image, one unit.
// The __global__ wrappers are in tq_aux.hip; tests/hostcheck runs the same bodies in host loops.
#pragma once
#include "../../include/tapqir_hip.h"
#include "tq_math.h"

// normalised 2-D Gaussian of one spot at pixel (column ic, row j): N(ic; x + tx, w) N(j; y + ty, w)  (util.py:15-64:
// x indexes the column axis, y the row axis)
TQ_HD float tq_spot_density(float ic, float j, float cx, float cy, float w) {
  const float inv2v = 0.5f * TQ_FRCP(w * w);
  const float dx = ic - cx, dy = j - cy;
  return inv2v * (1.0f / TQ_PI) * TQ_FEXP(-(dx * dx + dy * dy) * inv2v);
}

// ---- KSMOGN.rsample: value = Gamma(mu / g, 1 / g) + offset[odx], odx ~ Categorical(logits), per pixel ---------------
// Work item = pixel `pix` of unit `i`.  RNG stream (seed, 0, 0x900, i * P * P + pix): results do not depend on the launch
// geometry.  `height` already carries the presence indicator m_k (and alpha_qc for the crosstalk image: K = Q K' spots).
TQ_HD void tq_body_rsample(const tq_rsample_args& a, int64_t i, int pix) {
  const int P = a.P, j = pix / P, ic = pix - j * P;
  const float tx = a.xy[2 * i], ty = a.xy[2 * i + 1];
  float mu = a.background[i];
  for (int k = 0; k < a.K; ++k) {
    const float h = a.height[(int64_t)k * a.B + i];
    if (h != 0.0f)
      mu += h * tq_spot_density((float)ic, (float)j, a.x[(int64_t)k * a.B + i] + tx, a.y[(int64_t)k * a.B + i] + ty,
                                a.width[(int64_t)k * a.B + i]);
  }
  TqPhilox s;
  tq_philox_init(&s, a.seed, 0u, 0x900u, (uint64_t)i * (uint64_t)(P * P) + (uint64_t)pix);
  // offset category by inversion of the cumulative weights (softmax of the logits)
  float mx = a.offset_logits[0];
  for (int o = 1; o < a.O; ++o) mx = fmaxf(mx, a.offset_logits[o]);
  float tot = 0.0f;
  for (int o = 0; o < a.O; ++o) tot += TQ_FEXP(a.offset_logits[o] - mx);
  const float u = tq_uniform(&s) * tot;
  float cum = 0.0f, off = a.offset_samples[a.O - 1];
  for (int o = 0; o < a.O; ++o) {
    cum += TQ_FEXP(a.offset_logits[o] - mx);
    if (u < cum) {
      off = a.offset_samples[o];
      break;
    }
  }
  s.have = 0;  // the Gamma draw starts on a fresh Philox block
  const float g = a.gain[0];
  const float val = fmaxf(tq_sample_std_gamma(&s, mu * TQ_FRCP(g)) * g, 1.17549435e-38f);
  a.out[i * (int64_t)(P * P) + pix] = val + off;
}

// ---- snr_and_chi2 of one unit (n, f, c) with the posterior means of its spots and background ----------------------------
//   signal_k = sum_ij (D - b - offset_mean) N_k(i, j),   noise = sqrt(offset_var + b gain),   SNR_k = signal_k / noise
//   ideal    = b + sum_k h_k N_k,                        chi2  = mean_ij (D - ideal - offset_mean)^2 / ideal
TQ_HD void tq_body_snr_chi2(const tq_snr_args& a, int64_t u) {
  const int P = a.P, K = a.K, npix = P * P;
  const float* tile = a.images + u * npix;
  const float tx = a.xy[2 * u], ty = a.xy[2 * u + 1];
  const float b = a.background[u];
  float h[TQ_MAX_K], w[TQ_MAX_K], cx[TQ_MAX_K], cy[TQ_MAX_K], sig[TQ_MAX_K];
  for (int k = 0; k < K; ++k) {
    h[k] = a.height[(int64_t)k * a.U + u];
    w[k] = a.width[(int64_t)k * a.U + u];
    cx[k] = a.x[(int64_t)k * a.U + u] + tx;
    cy[k] = a.y[(int64_t)k * a.U + u] + ty;
    sig[k] = 0.0f;
  }
  float chi = 0.0f;
  for (int j = 0; j < P; ++j) {
    for (int ic = 0; ic < P; ++ic) {
      const float D = tile[j * P + ic];
      float ideal = b;
      for (int k = 0; k < K; ++k) {
        const float nk = tq_spot_density((float)ic, (float)j, cx[k], cy[k], w[k]);
        sig[k] += (D - b - a.offset_mean) * nk;
        ideal += h[k] * nk;
      }
      const float r = D - ideal - a.offset_mean;
      chi += r * r * TQ_FRCP(ideal);
    }
  }
  const float rnoise = TQ_FRCP(TQ_FSQRT(a.offset_var + b * a.gain));
  for (int k = 0; k < K; ++k) a.snr[(int64_t)k * a.U + u] = sig[k] * rnoise;
  a.chi2[u] = chi / (float)npix;
}
