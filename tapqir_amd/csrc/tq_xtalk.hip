// tq_xtalk.hip -- KSMOGN likelihood of the crosstalk model for CDNA4 / gfx950 (include/tapqir_hip.h:
// tq_ksmogn_crosstalk_log_prob; math in tq_xtalk.h).  16 lanes per AOI-frame group: the lanes stride the
// pixels of both channels, per-group sums are reduced over the DPP row and lane 0 writes the outputs.
#include <hip/hip_runtime.h>

#include "tq_dpp.h"
#include "tq_xtalk.h"

void tq_set_error(const char* msg);

template <int K, bool BWD>
__global__ __launch_bounds__(256) void tq_xtalk_kernel(const tq_xtalk_args a, const int64_t Bg) {
  constexpr int Q = TQ_XT_Q, NS = Q * K, MJ = 1 << NS;
  const int grp = threadIdx.x >> 4, r = threadIdx.x & 15;
  const int64_t g_raw = (int64_t)blockIdx.x * 16 + grp;
  const bool live = g_raw < Bg;
  const int64_t g = live ? g_raw : (Bg - 1);  // idle groups shadow the last one, stores masked
  const float gain = a.gain[0];
  const float rg = TQ_FRCP(gain);
  const float ln_g = TQ_FLOG(gain);
  TqOffsetInfo h;
  tq_offset_info(a.offset_samples, a.offset_logits, a.O, &h);
  TqXtGroup<K> G;
  tq_xtalk_load_group<K>(a, g, BWD, &G);
  TqXtAcc<K> A;
  tq_xt_acc_zero<K>(A);
  if (__all(fminf(G.b[0], G.b[1]) * rg >= TQ_FAST_ALPHA)) tq_xtalk_pixels<K, BWD, true>(a, G, h, r, 16, gain, rg, ln_g, A);
  else tq_xtalk_pixels<K, BWD, false>(a, G, h, r, 16, gain, rg, ln_g, A);
#pragma unroll
  for (int m = 0; m < MJ; ++m) A.ll[m] = tq_group_sum16(A.ll[m]);
  if (BWD) {
    A.acc_g = tq_group_sum16(A.acc_g);
#pragma unroll
    for (int c = 0; c < Q; ++c) A.acc_b[c] = tq_group_sum16(A.acc_b[c]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      A.S1x[s] = tq_group_sum16(A.S1x[s]);
      A.S1y[s] = tq_group_sum16(A.S1y[s]);
      A.S2[s] = tq_group_sum16(A.S2[s]);
#pragma unroll
      for (int c = 0; c < Q; ++c) A.S0[s][c] = tq_group_sum16(A.S0[s][c]);
    }
  }
  if (live && r == 0) tq_xtalk_finish<K, BWD>(a, g, G, A, rg);
}

// =============================================================================================
// Packed lane-per-AOI-frame kernel (contiguous batches, one offset, K = 2, alpha >= TQ_FAST_ALPHA everywhere).
//
// Same ideas as tq_ksmogn_il2_kernel (tq_ksmogn.hip): one lane per AOI-frame on the tile-interleaved copy of
// its (C, P, P) tile (a wave64 reads one contiguous 1 KiB row per 4 pixels), two horizontally adjacent pixels per
// lane in float2 registers so that the per-pixel algebra issues as packed v_pk_* instructions, the single-offset
// formulation of tq_pixel.h (per combination and pixel: rcp, log2 and a handful of packed fmas; the spot-free
// combination comes from the per-unit data statistics), one accumulator per combination.
// mu(m) of the 16 joint combinations is (low pair of spots) + (high pair of spots): 4 + 3 stored sums.
// =============================================================================================
template <bool BWD>
__global__ __launch_bounds__(256, 2) void tq_xtalk_il_kernel(const tq_xtalk_args a, const int64_t Bg) {
  constexpr int K = 2, Q = TQ_XT_Q, NS = Q * K, MJ = 1 << NS;
  const int64_t g_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = g_raw < Bg;
  const int64_t g = live ? g_raw : (Bg - 1);
  const int P = a.P, npix = P * P;
  const int npix4c = npix >> 2;          // float4 groups per channel (P even)
  const int npix4 = 2 * npix4c;          // per AOI-frame tile
  // idle lanes of the last workgroup read the LAST tile too: the interleaved buffer ends with its 64-tile block
  const float4* src = reinterpret_cast<const float4*>(a.images_il) + ((g >> 6) * npix4) * 64 + (g & 63);

  const float gain = a.gain[0];
  const float rg = TQ_FRCP(gain);
  const float ln_g = TQ_FLOG(gain);
  const float off0 = a.offset_samples[0];
  TqXtGroup<K> G;
  tq_xtalk_load_group<K>(a, g, BWD, &G);
  TqXtAcc<K> A;
  tq_xt_acc_zero<K>(A);

  if (!__all(fminf(G.b[0], G.b[1]) * rg >= TQ_FAST_ALPHA)) {
    // some AOI-frame of this wave has a small background / gain: exact Binet terms, scalar routine on the plain layout
    TqOffsetInfo h;
    tq_offset_info(a.offset_samples, a.offset_logits, a.O, &h);
    tq_xtalk_pixels<K, BWD, false>(a, G, h, 0, 1, gain, rg, ln_g, A);
    if (live) tq_xtalk_finish<K, BWD>(a, g, G, A, rg);
    return;
  }

  // constants of the fast single-offset pixel (tq_ksmogn.hip: TqFastConst)
  const float g2 = gain * gain, rl2 = 1.0f / TQ_LN2;
  const float c_ca = TQ_LN2 * rg, c_cb = 0.5f * TQ_LN2;
  const float c_s1 = gain * (1.0f / 12.0f), c_s3 = -g2 * gain * (1.0f / 360.0f);
  const float c_d1 = 0.5f * gain * rl2, c_d2 = g2 * (1.0f / 12.0f) * rl2, c_d4 = -g2 * g2 * (1.0f / 120.0f) * rl2;

  float nl2[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) nl2[s] = -0.5f * TQ_FRCP(G.w[s] * G.w[s]) * TQ_LOG2E;

  tq_f2 T[MJ];
#pragma unroll
  for (int m = 0; m < MJ; ++m) T[m] = tq2(0.0f);
  float SN[NS][Q];
  bool bad = false;
  float4 cur = src[0];
#pragma unroll
  for (int c = 0; c < Q; ++c) {
    float amp[NS], cx[NS], cy[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      amp[s] = G.alpha[s / K][c] * G.h[s] * (-nl2[s]) * (TQ_LN2 / TQ_PI);
      cx[s] = G.x[s] + G.tx[c];
      cy[s] = G.y[s] + G.ty[c];
    }
    tq_f2 acc_b = tq2(0.0f), sn[NS], s0[NS], s1x[NS], s1y[NS], s2[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) sn[s] = s0[s] = s1x[s] = s1y[s] = s2[s] = tq2(0.0f);
    int ic = 0, jr = 0;
    float dy[NS], dy2[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      dy[s] = -cy[s];
      dy2[s] = dy[s] * dy[s];
    }
    for (int q4 = 0; q4 < npix4c; ++q4) {
      const float4 d4 = cur;
      const int nxt = c * npix4c + q4 + 1;
      if (nxt < npix4) cur = src[(int64_t)nxt * 64];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const tq_f2 v = (half ? (tq_f2){d4.z, d4.w} : (tq_f2){d4.x, d4.y}) - off0;
        bad = bad || !(v.x > 0.0f) || !(v.y > 0.0f);
        const tq_f2 fic = (tq_f2){(float)ic, (float)(ic + 1)};
        tq_f2 spot[NS], dx[NS], d2[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          dx[s] = fic - cx[s];
          d2[s] = dx[s] * dx[s] + dy2[s];
          spot[s] = amp[s] * tq2_exp2(d2[s] * nl2[s]);
          sn[s] += spot[s];
        }
        // mu(m) = lo[m & 3] + hi[m >> 2]
        tq_f2 lo[4], hi[4];
        lo[0] = tq2(G.b[c]);
        lo[1] = lo[0] + spot[0];
        lo[2] = lo[0] + spot[1];
        lo[3] = lo[1] + spot[1];
        hi[0] = tq2(0.0f);
        hi[1] = spot[2];
        hi[2] = spot[3];
        hi[3] = spot[2] + spot[3];
        tq_f2 qs[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) qs[s] = tq2(0.0f);
#pragma unroll
        for (int m = 1; m < MJ; ++m) {
          const tq_f2 mu = (m >> 2) ? lo[m & 3] + hi[m >> 2] : lo[m & 3];
          const tq_f2 r = tq2_rcp(mu);
          const tq_f2 l2 = tq2_log2(v * r);
          const tq_f2 u = r * r;
          T[m] += (mu * c_ca - c_cb) * l2;
          T[m] -= r * (u * c_s3 + c_s1);
          if (BWD) {
            const tq_f2 cw = G.W[m] * (r * (r * (u * c_d4 + c_d2) + c_d1) + l2);  // W(m) da / ln2
            acc_b += cw;
#pragma unroll
            for (int s = 0; s < NS; ++s)
              if ((m >> s) & 1) qs[s] += cw;
          }
        }
        if (BWD) {
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            const tq_f2 aq = qs[s] * spot[s];
            s0[s] += aq;
            s1x[s] += aq * dx[s];
            s1y[s] += aq * dy[s];
            s2[s] += aq * d2[s];
          }
        }
        ic += 2;
        if (ic == P) {
          ic = 0;
          ++jr;
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            dy[s] = (float)jr - cy[s];
            dy2[s] = dy[s] * dy[s];
          }
        }
      }
    }
    // fold the two pixel slots of this channel
    A.acc_b[c] = (acc_b.x + acc_b.y) * TQ_LN2;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      SN[s][c] = sn[s].x + sn[s].y;
      if (BWD) {
        A.S0[s][c] = (s0[s].x + s0[s].y) * TQ_LN2;
        A.S1x[s] += (s1x[s].x + s1x[s].y) * TQ_LN2;
        A.S1y[s] += (s1y[s].x + s1y[s].y) * TQ_LN2;
        A.S2[s] += (s2[s].x + s2[s].y) * TQ_LN2;
      }
    }
  }

  // assembly from the data statistics of the two (n, f, c) tiles (formulas: tq_pixel.h, single-offset path)
  const float fnpix = (float)npix;
  const float lw0 = a.offset_logits[0] - TQ_LN_SQRT_2PI;
  float Wsum = 0.0f;
#pragma unroll
  for (int m = 0; m < MJ; ++m) Wsum += G.W[m];
#pragma unroll
  for (int m = 1; m < MJ; ++m) A.ll[m] = T[m].x + T[m].y;
  A.ll[0] = 0.0f;
  float acc_g = 0.0f;
#pragma unroll
  for (int c = 0; c < Q; ++c) {
    const int64_t u = G.ug * Q + c;
    const float S_v = a.pixstats[u], S_lv = a.pixstats[a.nb_full * (int64_t)a.F * Q + u];
    bad = bad || a.pixstats[2 * (a.nb_full * (int64_t)a.F * Q) + u] > 0.0f;
    TqCombo0 c0;
    tq_combo0_prepare(G.b[c], rg, gain, ln_g, &c0);
    const float common = lw0 * fnpix - S_lv;
    const float S_lvg = S_lv - fnpix * ln_g;
    const float sl0 = S_lv - fnpix * c0.lnb;
    const float mmv = G.b[c] * fnpix - S_v;
    A.ll[0] += common + rg * (G.b[c] * sl0 + mmv) + 0.5f * (S_lvg - sl0) - fnpix * c0.S;
#pragma unroll
    for (int m = 1; m < MJ; ++m) {
      float sn = 0.0f;
#pragma unroll
      for (int s = 0; s < NS; ++s)
        if ((m >> s) & 1) sn += SN[s][c];
      A.ll[m] += common + rg * (mmv + sn) + 0.5f * S_lvg;
    }
    if (BWD) {
      A.acc_b[c] += G.W[0] * (sl0 + fnpix * c0.c_da);
      // sum_m W_m [alpha_m (da_m + 1) - v/g] = (1/g) [ sum W mu da + sum W mu - (sum W) v ]
      float mu_da = G.b[c] * A.acc_b[c], mu_w = 0.0f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float Ws = 0.0f;
#pragma unroll
        for (int m = 0; m < MJ; ++m)
          if ((m >> s) & 1) Ws += G.W[m];
        mu_da += A.S0[s][c];
        mu_w += Ws * SN[s][c];
      }
      acc_g += rg * (mu_da + mu_w + Wsum * mmv);
    }
  }
  A.acc_g = acc_g;
  if (bad) {  // a pixel at or below the offset: log 0, no gradient (ksmogn.py:226)
#pragma unroll
    for (int m = 0; m < MJ; ++m) A.ll[m] = -INFINITY;
    A.acc_g = 0.0f;
#pragma unroll
    for (int c = 0; c < Q; ++c) A.acc_b[c] = 0.0f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      A.S1x[s] = A.S1y[s] = A.S2[s] = 0.0f;
#pragma unroll
      for (int c = 0; c < Q; ++c) A.S0[s][c] = 0.0f;
    }
  }
  if (live) tq_xtalk_finish<K, BWD>(a, g, G, A, rg);
}

extern "C" int tq_ksmogn_crosstalk_log_prob(const tq_xtalk_args* a, void* stream) {
  if (!a || !a->images || !a->xy || !a->background || !a->height || !a->width || !a->x || !a->y || !a->gain ||
      !a->alpha || !a->offset_samples || !a->offset_logits || (!a->ll && !a->ll_joint)) {
    tq_set_error("tq_ksmogn_crosstalk_log_prob: NULL required pointer");
    return TQ_ERR_ARG;
  }
  if (a->C != TQ_XT_Q || a->K < 1 || a->K > 2 || a->P < 2 || a->P > TQ_MAX_P || a->O < 1 || a->nb < 1 || a->fb < 1 ||
      (int64_t)a->nb * a->fb * a->C >= (int64_t)1 << 31) {
    tq_set_error("tq_ksmogn_crosstalk_log_prob: needs Q = C = 2, K <= 2 and a non-empty batch");
    return TQ_ERR_ARG;
  }
  if (a->ll && !a->m_logit) {
    tq_set_error("tq_ksmogn_crosstalk_log_prob: the per-dye marginals need m_logit");
    return TQ_ERR_ARG;
  }
  const bool bwd = a->g_background != nullptr;
  if (bwd && (!a->g_height || !a->g_width || !a->g_x || !a->g_y || !a->g_gain || (!a->gout && !a->m_logit))) {
    tq_set_error("tq_ksmogn_crosstalk_log_prob: backward requested but a gradient output or the upstream weights are NULL");
    return TQ_ERR_ARG;
  }
  const int64_t Bg = (int64_t)a->nb * a->fb;
  hipStream_t st = (hipStream_t)stream;
  if (a->images_il && a->pixstats && a->K == 2 && a->O == 1 && (a->P % 2) == 0 && !a->ndx && !a->fdx &&
      a->nb == a->nb_full && a->fb == a->F && Bg >= a->il_min_units) {
    const dim3 gridl((unsigned)((Bg + 255) / 256)), blockl(256);
    if (bwd) hipLaunchKernelGGL((tq_xtalk_il_kernel<true>), gridl, blockl, 0, st, *a, Bg);
    else hipLaunchKernelGGL((tq_xtalk_il_kernel<false>), gridl, blockl, 0, st, *a, Bg);
    const hipError_t el = hipGetLastError();
    if (el != hipSuccess) {
      char buf[200];
      snprintf(buf, sizeof(buf), "tq_xtalk_il_kernel: %s", hipGetErrorString(el));
      tq_set_error(buf);
      return TQ_ERR_LAUNCH;
    }
    return TQ_OK;
  }
  const dim3 grid((unsigned)((Bg + 15) / 16)), block(256);
  if (a->K == 1) {
    if (bwd) hipLaunchKernelGGL((tq_xtalk_kernel<1, true>), grid, block, 0, st, *a, Bg);
    else hipLaunchKernelGGL((tq_xtalk_kernel<1, false>), grid, block, 0, st, *a, Bg);
  } else {
    if (bwd) hipLaunchKernelGGL((tq_xtalk_kernel<2, true>), grid, block, 0, st, *a, Bg);
    else hipLaunchKernelGGL((tq_xtalk_kernel<2, false>), grid, block, 0, st, *a, Bg);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[200];
    snprintf(buf, sizeof(buf), "tq_xtalk_kernel: %s", hipGetErrorString(e));
    tq_set_error(buf);
    return TQ_ERR_LAUNCH;
  }
  return TQ_OK;
}
