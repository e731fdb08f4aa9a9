// tq_xtalk.h -- KSMOGN likelihood of the crosstalk model (host+device inline bodies).
//
// Reference: tapqir/distributions/ksmogn.py:93-105, 146-165 (alpha branch) and its use at
// tapqir/models/crosstalk.py:262-281: ONE observation per AOI-frame with event shape (C, P, P),
//   image_c = b_c + sum_q alpha_qc sum_k m_qk h_qk N(i; x_qk + tx_c, w_qk) N(j; y_qk + ty_c, w_qk),
// evaluated for every joint spot-presence combination m in {0,1}^(Q K)  (bit q*K + k of the combination
// index = m_qk).  Q = C = 2 (the reference's experimental model indexes dyes and channels alike).
//
// A work item is one AOI-frame GROUP g = C consecutive units g*C + c of the cosmos layout (unit (g, c)
// carries background b_c and the spots of dye q = c).  The pixel routine below is the general
// offset-histogram formulation of tq_pixel.h evaluated on chunks of 8 combinations; the group routine
// can be run by any number of cooperating lanes (pixels strided by `nl`), whose partial TqXtAcc are then
// summed and passed to tq_xtalk_finish.
#pragma once
#include "../../include/tapqir_hip.h"
#include "tq_pixel.h"

#define TQ_XT_Q 2
#define TQ_XT_CHUNK 8

template <int K>
struct TqXtAcc {
  float ll[1 << (TQ_XT_Q * K)];
  float acc_b[TQ_XT_Q], acc_g;
  float S0[TQ_XT_Q * K][TQ_XT_Q];                                   // sum q*spot per (spot, channel)
  float S1x[TQ_XT_Q * K], S1y[TQ_XT_Q * K], S2[TQ_XT_Q * K];        // centred moments, summed over channels
};

template <int K>
TQ_HD void tq_xt_acc_zero(TqXtAcc<K>& A) {
  constexpr int NS = TQ_XT_Q * K, MJ = 1 << NS;
#pragma unroll
  for (int m = 0; m < MJ; ++m) A.ll[m] = 0.0f;
  A.acc_g = 0.0f;
#pragma unroll
  for (int c = 0; c < TQ_XT_Q; ++c) A.acc_b[c] = 0.0f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    A.S1x[s] = A.S1y[s] = A.S2[s] = 0.0f;
#pragma unroll
    for (int c = 0; c < TQ_XT_Q; ++c) A.S0[s][c] = 0.0f;
  }
}

// per-group constants
template <int K>
struct TqXtGroup {
  int64_t ug;   // dataset group index n*F + f
  int n;
  float b[TQ_XT_Q];
  float h[TQ_XT_Q * K], w[TQ_XT_Q * K], x[TQ_XT_Q * K], y[TQ_XT_Q * K];  // spot s = q*K + k
  float tx[TQ_XT_Q], ty[TQ_XT_Q];
  float alpha[TQ_XT_Q][TQ_XT_Q];  // [q][c]
  float W[1 << (TQ_XT_Q * K)];    // upstream weights of the joint combinations (backward)
  float pq[TQ_XT_Q][K];           // q(m_qk = 1) (when m_logit is given)
};

TQ_HD float tq_xt_sigmoid(float u) {
  const float e = TQ_FEXP(-fabsf(u));
  const float r = TQ_FRCP(1.0f + e);
  return u >= 0.0f ? r : e * r;
}

template <int K>
TQ_HD void tq_xtalk_load_group(const tq_xtalk_args& a, int64_t g, bool bwd, TqXtGroup<K>* G) {
  constexpr int Q = TQ_XT_Q, NS = Q * K, MJ = 1 << NS;
  const int64_t B = (int64_t)a.nb * a.fb * Q, Bg = (int64_t)a.nb * a.fb;
  const int bi = (int)(g % a.fb), ai = (int)(g / a.fb);
  G->n = a.ndx ? a.ndx[ai] : ai;
  const int f = a.fdx ? a.fdx[bi] : bi;
  G->ug = (int64_t)G->n * a.F + f;
#pragma unroll
  for (int c = 0; c < Q; ++c) {
    G->b[c] = a.background[g * Q + c];
    G->tx[c] = a.xy[2 * (G->ug * Q + c)];
    G->ty[c] = a.xy[2 * (G->ug * Q + c) + 1];
#pragma unroll
    for (int q = 0; q < Q; ++q) G->alpha[q][c] = a.alpha[q * Q + c];
  }
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int64_t i = (int64_t)k * B + g * Q + q;
      G->h[q * K + k] = a.height[i];
      G->w[q * K + k] = a.width[i];
      G->x[q * K + k] = a.x[i];
      G->y[q * K + k] = a.y[i];
      G->pq[q][k] = a.m_logit ? tq_xt_sigmoid(a.m_logit[(int64_t)k * a.m_kstride + G->ug * Q + q]) : 0.5f;
    }
  if (bwd) {
    if (a.gout) {
#pragma unroll
      for (int m = 0; m < MJ; ++m) G->W[m] = a.gout[(int64_t)m * Bg + g];
    } else {
      const float sc = a.scale * ((a.aoi_mask == nullptr || a.aoi_mask[G->n]) ? 1.0f : 0.0f);
#pragma unroll
      for (int m = 0; m < MJ; ++m) {
        float wv = sc;
#pragma unroll
        for (int s = 0; s < NS; ++s) wv *= ((m >> s) & 1) ? G->pq[s / K][s % K] : 1.0f - G->pq[s / K][s % K];
        G->W[m] = wv;
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < MJ; ++m) G->W[m] = 0.0f;
  }
}

// Pixels pix = r, r + nl, ... of both channels of one group.
template <int K, bool BWD, bool FAST>
TQ_HD void tq_xtalk_pixels(const tq_xtalk_args& a, const TqXtGroup<K>& G, const TqOffsetInfo& h, int r, int nl, float g,
                           float rg, float ln_g, TqXtAcc<K>& A) {
  constexpr int Q = TQ_XT_Q, NS = Q * K, MJ = 1 << NS, CH = (MJ < TQ_XT_CHUNK ? MJ : TQ_XT_CHUNK);
  const int P = a.P, npix = P * P;
  const float off0 = a.offset_samples[0], lw0 = a.offset_logits[0];
  float nl2[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) nl2[s] = -0.5f * TQ_FRCP(G.w[s] * G.w[s]) * TQ_LOG2E;
#pragma unroll
  for (int c = 0; c < Q; ++c) {  // unrolled: every per-channel array index is a compile-time constant
    float amp[NS], cx[NS], cy[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      amp[s] = G.alpha[s / K][c] * G.h[s] * (-nl2[s]) * (TQ_LN2 / TQ_PI);  // alpha_qc h / (2 pi w^2)
      cx[s] = G.x[s] + G.tx[c];
      cy[s] = G.y[s] + G.ty[c];
    }
    const float* tile = a.images + (G.ug * Q + c) * npix;
    for (int pix = r; pix < npix; pix += nl) {
      const int j = pix / P, ic = pix - j * P;
      const float D = tile[pix];
      float spot[NS], dx[NS], dy[NS], d2[NS], qs[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        dx[s] = (float)ic - cx[s];
        dy[s] = (float)j - cy[s];
        d2[s] = dx[s] * dx[s] + dy[s] * dy[s];
        spot[s] = amp[s] * TQ_FEXP2(d2[s] * nl2[s]);
        qs[s] = 0.0f;
      }
#pragma unroll
      for (int base = 0; base < MJ; base += CH) {
        float mu[CH], lp[CH], da[CH], gq[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          mu[e] = G.b[c];
#pragma unroll
          for (int s = 0; s < NS; ++s)
            if (((base + e) >> s) & 1) mu[e] += spot[s];
        }
        if (a.O == 1) tq_pix_single_offset<CH, BWD, FAST>(D, mu, off0, lw0, g, rg, ln_g, lp, da, gq);
        else tq_pix_multi_offset<CH, BWD, FAST>(D, mu, a.offset_samples, a.offset_logits, a.O, h, g, rg, ln_g, lp, da, gq);
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          A.ll[base + e] += lp[e];
          if (BWD) {
            const float cw = G.W[base + e] * da[e];
            A.acc_b[c] += cw;
            A.acc_g += G.W[base + e] * gq[e];
#pragma unroll
            for (int s = 0; s < NS; ++s)
              if (((base + e) >> s) & 1) qs[s] += cw;
          }
        }
      }
      if (BWD) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float aq = qs[s] * spot[s];
          A.S0[s][c] += aq;
          A.S1x[s] += aq * dx[s];
          A.S1y[s] += aq * dy[s];
          A.S2[s] += aq * d2[s];
        }
      }
    }
  }
}

// Outputs of one group from its (lane-reduced) sums.
template <int K, bool BWD>
TQ_HD void tq_xtalk_finish(const tq_xtalk_args& a, int64_t g, const TqXtGroup<K>& G, const TqXtAcc<K>& A, float rg) {
  constexpr int Q = TQ_XT_Q, NS = Q * K, MJ = 1 << NS, MK = 1 << K;
  const int64_t B = (int64_t)a.nb * a.fb * Q, Bg = (int64_t)a.nb * a.fb;
  if (a.ll_joint) {
#pragma unroll
    for (int m = 0; m < MJ; ++m) a.ll_joint[(int64_t)m * Bg + g] = A.ll[m];
  }
  if (a.ll) {
    // per-dye marginal of the likelihood over the OTHER dye's guide-enumerated spots:
    //   LL_q(m_q) = sum_{m_-q} prod_{k} q(m_-q,k) ll(m_q, m_-q),   E[ll] = sum_m W(m) ll(m)
    float Wq[Q][MK];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int mq = 0; mq < MK; ++mq) {
        float wv = 1.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) wv *= ((mq >> k) & 1) ? G.pq[q][k] : 1.0f - G.pq[q][k];
        Wq[q][mq] = wv;
      }
    float ell = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
#pragma unroll
      for (int mq = 0; mq < MK; ++mq) {
        float acc = 0.0f;
#pragma unroll
        for (int mo = 0; mo < MK; ++mo) {
          const int m = q == 0 ? (mq | (mo << K)) : (mo | (mq << K));
          const float wv = Wq[1 - q][mo];
          acc += wv == 0.0f ? 0.0f : wv * A.ll[m];
        }
        a.ll[(int64_t)mq * B + g * Q + q] = acc;
        if (q == 0) ell += Wq[0][mq] == 0.0f ? 0.0f : Wq[0][mq] * acc;
      }
    }
    // every dye's unit adds E[ll] to the ELBO through its own Dice sum: units c >= 1 carry the excess
    if (a.ell_excess) {
      a.ell_excess[g * Q] = 0.0f;
#pragma unroll
      for (int c = 1; c < Q; ++c) a.ell_excess[g * Q + c] = ell;
    }
  }
  if (BWD) {
    // an AOI-frame with a pixel at or below every offset has log p = -inf for every combination: no gradient
    // (selected, not multiplied: the sums of such an AOI-frame may hold inf / NaN)
    const bool dead = A.ll[0] == -INFINITY;
#pragma unroll
    for (int c = 0; c < Q; ++c) {
      a.g_background[g * Q + c] = dead ? 0.0f : A.acc_b[c] * rg;
      a.g_gain[g * Q + c] = (c == 0 && !dead) ? -A.acc_g * rg : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int s = q * K + k;
        const int64_t i = (int64_t)k * B + g * Q + q;
        const float rw = TQ_FRCP(G.w[s]);
        const float s0 = A.S0[s][0] + A.S0[s][1];
        a.g_height[i] = dead ? 0.0f : s0 * rg * TQ_FRCP(G.h[s]);
        a.g_x[i] = dead ? 0.0f : rg * A.S1x[s] * rw * rw;
        a.g_y[i] = dead ? 0.0f : rg * A.S1y[s] * rw * rw;
        a.g_width[i] = dead ? 0.0f : rg * (A.S2[s] * rw * rw * rw - 2.0f * s0 * rw);
      }
#pragma unroll
      for (int c = 0; c < Q; ++c) {
        float sa = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) sa += A.S0[q * K + k][c];
        if (a.g_alpha) a.g_alpha[(int64_t)q * B + g * Q + c] = dead ? 0.0f : rg * sa * TQ_FRCP(G.alpha[q][c]);
      }
    }
  }
}
