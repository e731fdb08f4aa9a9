// tq_ksmogn_dev.h -- device-side building blocks of the KSMOGN likelihood kernels shared by tq_ksmogn.hip and
// tq_cosmos.hip: per-lane accumulators, the per-pixel routines, assembly / store of a unit's outputs, the Dice weights,
// and the 16-lanes-per-unit tile routine (LDS-staged tile, separable Gaussian factors in LDS, DPP row sums).
// See tq_ksmogn.hip for the kernels and tq_pixel.h for the arithmetic.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/tapqir_hip.h"
#include "tq_dpp.h"
#include "tq_pixel.h"

#define TQ_LANES_PER_UNIT 16
#define TQ_UNITS_PER_BLOCK 16
#define TQ_BLOCK (TQ_LANES_PER_UNIT * TQ_UNITS_PER_BLOCK)

__device__ __forceinline__ float tq_fast_sigmoid(float u) { return TQ_FRCP(1.0f + TQ_FEXP(-u)); }

// LDS of the 16-lane kernel: [16 units][tile stride] staged pixels, then [16 units][2K][P] factors.
// The tile stride is npix rounded up to 16 (mod 32) floats so that the two units sharing a
// 32-lane ds_read_b32 group hit disjoint bank halves.
__host__ __device__ inline int tq_tile_stride(int npix) { return ((npix + 15) / 32) * 32 + 16; }

// ---- per-lane accumulators of the pixel loop ------------------------------------------------------
template <int K>
struct TqPixAcc {
  // single-offset path, combinations mi >= 1 (tq_pixel.h): sums of mu*log2(v/mu), log2(v/mu), S(alpha)
  // general path: ll[mi] = log-likelihood per combination, the other two unused
  float ll[1 << K], sl[1 << K], sS[1 << K];
  float acc_b;                       // sum_m W_m da_m                (combination 0 added analytically)
  float acc_g;                       // general path only: sum_m W_m [alpha (da+1) - E_o v / g]
  float S0[K], Sx[K], Sy[K], Sr[K];  // spot-weighted moments: sum q*spot*{1, i, j, i^2+j^2}
  float SN[K];                       // single-offset path: sum spot_k
};

template <int K>
__device__ __forceinline__ void tq_acc_zero(TqPixAcc<K>& A) {
#pragma unroll
  for (int mi = 0; mi < (1 << K); ++mi) A.ll[mi] = A.sl[mi] = A.sS[mi] = 0.0f;
  A.acc_b = A.acc_g = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) A.S0[k] = A.Sx[k] = A.Sy[k] = A.Sr[k] = A.SN[k] = 0.0f;
}

// One pixel, single-offset path (the unit's data statistics cover everything else).
template <int K, bool BWD, bool FAST>
__device__ __forceinline__ void tq_pixel_one_offset(TqPixAcc<K>& A, float v, float b, const float* spot, const float* W,
                                                    float fic, float fj, float g, float rg, float ln_g) {
  constexpr int M = 1 << K;
  float da[M];
#pragma unroll
  for (int mi = 1; mi < M; ++mi) {
    float mu = b;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if ((mi >> k) & 1) mu += spot[k];
    float l2, S;
    tq_pix_one_offset<FAST>(v, mu, g, rg, ln_g, &l2, &S, &da[mi]);
    A.ll[mi] += mu * l2;
    A.sl[mi] += l2;
    A.sS[mi] += S;
  }
  if (BWD) {
    float q[K];
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = 0.0f;
#pragma unroll
    for (int mi = 1; mi < M; ++mi) {
      const float cw = W[mi] * da[mi];
      A.acc_b += cw;
#pragma unroll
      for (int k = 0; k < K; ++k)
        if ((mi >> k) & 1) q[k] += cw;
    }
    const float r2 = fic * fic + fj * fj;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float aq = q[k] * spot[k];
      A.S0[k] += aq;
      A.Sx[k] += aq * fic;
      A.Sy[k] += aq * fj;
      A.Sr[k] += aq * r2;
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) A.SN[k] += spot[k];
}

// One pixel, general path (offset histogram, tq_pixel.h).
template <int K, bool BWD, bool FAST>
__device__ __forceinline__ void tq_pixel_multi_offset(TqPixAcc<K>& A, const tq_ksmogn_args& a, const TqOffsetInfo& h,
                                                      float D, float ln_g, float b, const float* spot, const float* W,
                                                      float fic, float fj, float g, float rg, const float* tab = nullptr) {
  constexpr int M = 1 << K;
  float mu[M], lp[M], da[M], gq[M];
#pragma unroll
  for (int mi = 0; mi < M; ++mi) {
    mu[mi] = b;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if ((mi >> k) & 1) mu[mi] += spot[k];
  }
  if (tab) tq_pix_multi_offset_tab<M, BWD, FAST>(D, mu, tab, a.O, h, g, rg, ln_g, lp, da, gq);
  else tq_pix_multi_offset<M, BWD, FAST>(D, mu, a.offset_samples, a.offset_logits, a.O, h, g, rg, ln_g, lp, da, gq);
  float q[K];
#pragma unroll
  for (int k = 0; k < K; ++k) q[k] = 0.0f;
#pragma unroll
  for (int mi = 0; mi < M; ++mi) {
    A.ll[mi] += lp[mi];
    if (BWD) {
      const float cw = W[mi] * da[mi];
      A.acc_b += cw;
      A.acc_g += W[mi] * gq[mi];
#pragma unroll
      for (int k = 0; k < K; ++k)
        if ((mi >> k) & 1) q[k] += cw;
    }
  }
  if (BWD) {
    const float r2 = fic * fic + fj * fj;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float aq = q[k] * spot[k];
      A.S0[k] += aq;
      A.Sx[k] += aq * fic;
      A.Sy[k] += aq * fj;
      A.Sr[k] += aq * r2;
    }
  }
}

// Single-offset path: assemble log p(m) and the combination-0 gradient part from the running sums and
// the unit's data statistics (formulas in tq_pixel.h).
template <int K, bool BWD>
__device__ __forceinline__ void tq_pixel_assemble_one_offset(const tq_ksmogn_args& a, TqPixAcc<K>& A, const float* W,
                                                             float b, float g, float rg, float ln_g, float fnpix,
                                                             float S_v, float S_lv, const float* logit0 = nullptr) {
  // logit0: the log-weight of the single offset if the caller has it in a register (the persistent kernel keeps
  // register-destination loads out of its tile loop)
  constexpr int M = 1 << K;
  TqCombo0 c0;
  tq_combo0_prepare(b, rg, g, ln_g, &c0);
  const float lw0 = (logit0 ? *logit0 : a.offset_logits[0]) - TQ_LN_SQRT_2PI;
  const float common = lw0 * fnpix - S_lv;             // sum [ln w - ln sqrt(2pi) - ln v]
  const float S_lvg = S_lv - fnpix * ln_g;             // sum [ln v - ln g]
  const float sl0 = S_lv - fnpix * c0.lnb;             // sum ln(v / b)
  const float mu_minus_v0 = b * fnpix - S_v;           // sum (b - v)
  A.ll[0] = common + rg * (b * sl0 + mu_minus_v0) + 0.5f * (S_lvg - sl0) - fnpix * c0.S;
#pragma unroll
  for (int mi = 1; mi < M; ++mi) {
    float sn = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if ((mi >> k) & 1) sn += A.SN[k];
    A.ll[mi] = common + rg * (TQ_LN2 * A.ll[mi] + mu_minus_v0 + sn) + 0.5f * (S_lvg - TQ_LN2 * A.sl[mi]) - A.sS[mi];
  }
  if (BWD) A.acc_b += W[0] * (sl0 + fnpix * c0.c_da);  // sum_pix W_0 da_0
}

// Turn the (already lane-reduced) pixel sums of one unit into its outputs.
template <int K, bool ONE_OFFSET, bool BWD>
__device__ __forceinline__ void tq_pixel_store(const tq_ksmogn_args& a, int64_t B, int64_t i, const TqPixAcc<K>& A,
                                               const float* W, float b, float rg, const float* hk, const float* wk,
                                               const float* cx, const float* cy, float fnpix, float S_v, bool bad,
                                               float* OUT = nullptr) {
  // OUT (registers of the caller: ll[M], g_background, g_gain, g_height[K], g_width[K], g_x[K], g_y[K] -- the row order
  // of the step workspace `pix`) replaces the stores: the fused pixel + per-unit kernel consumes the values in place
  constexpr int M = 1 << K;
#pragma unroll
  for (int mi = 0; mi < M; ++mi) {
    const float v = bad ? -INFINITY : A.ll[mi];
    if (OUT) OUT[mi] = v;
    else a.ll[(int64_t)mi * B + i] = v;
  }
  if (BWD) {
    // a unit with a pixel at or below every offset has log p = -inf for every combination: no gradient
    // (selected, not multiplied: the sums of such a unit may hold inf / NaN)
    const bool dead = bad || A.ll[0] == -INFINITY;
    // d alpha = d mu / g for every mu-parameter
    const float gb = dead ? 0.0f : A.acc_b * rg;
    if (OUT) OUT[M] = gb;
    else a.g_background[i] = gb;
    float acc_g = A.acc_g;
    if (ONE_OFFSET) {
      // sum_m W_m [alpha_m (da_m + 1) - v/g] = (1/g) [ sum_m W_m mu_m da_m + sum_m W_m mu_m - (sum_m W_m) v ]
      float Wsum = 0.0f, mu_da = b * A.acc_b, mu_w = 0.0f;
#pragma unroll
      for (int mi = 0; mi < M; ++mi) Wsum += W[mi];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float Wk = 0.0f;
#pragma unroll
        for (int mi = 0; mi < M; ++mi)
          if ((mi >> k) & 1) Wk += W[mi];
        mu_da += A.S0[k];
        mu_w += Wk * A.SN[k];
      }
      acc_g = rg * (mu_da + mu_w + Wsum * (b * fnpix - S_v));
    }
    const float gg = dead ? 0.0f : -acc_g * rg;
    if (OUT) OUT[M + 1] = gg;
    else a.g_gain[i] = gg;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float rw = TQ_FRCP(wk[k]);
      const float S1x = A.Sx[k] - cx[k] * A.S0[k];
      const float S1y = A.Sy[k] - cy[k] * A.S0[k];
      const float S2 = A.Sr[k] - 2.0f * (cx[k] * A.Sx[k] + cy[k] * A.Sy[k]) + (cx[k] * cx[k] + cy[k] * cy[k]) * A.S0[k];
      const float gh = dead ? 0.0f : A.S0[k] * rg * TQ_FRCP(hk[k]);
      const float gx = dead ? 0.0f : rg * S1x * rw * rw;
      const float gy = dead ? 0.0f : rg * S1y * rw * rw;
      const float gw = dead ? 0.0f : rg * (S2 * rw * rw * rw - 2.0f * A.S0[k] * rw);
      if (OUT) {
        OUT[M + 2 + k] = gh;
        OUT[M + 2 + K + k] = gw;
        OUT[M + 2 + 2 * K + k] = gx;
        OUT[M + 2 + 3 * K + k] = gy;
      } else {
        a.g_height[k * B + i] = gh;
        a.g_x[k * B + i] = gx;
        a.g_y[k * B + i] = gy;
        a.g_width[k * B + i] = gw;
      }
    }
  }
}

// Dice weights of the cosmos guide (or the caller's upstream weights)
template <int K>
__device__ __forceinline__ void tq_load_weights(const tq_ksmogn_args& a, int64_t B, int64_t i, int64_t u, int n,
                                                float* W) {
  constexpr int M = 1 << K;
  if (a.gout) {
#pragma unroll
    for (int mi = 0; mi < M; ++mi) W[mi] = a.gout[(int64_t)mi * B + i];
  } else {
    float p1[K], p0[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float uk = a.m_logit[k * a.m_kstride + u];
      p1[k] = tq_fast_sigmoid(uk);
      p0[k] = tq_fast_sigmoid(-uk);
    }
    const float sc = a.scale * ((a.aoi_mask == nullptr || a.aoi_mask[n]) ? 1.0f : 0.0f);
#pragma unroll
    for (int mi = 0; mi < M; ++mi) {
      float w = sc;
#pragma unroll
      for (int k = 0; k < K; ++k) w *= ((mi >> k) & 1) ? p1[k] : p0[k];
      W[mi] = w;
    }
  }
}

// =============================================================================================
// 16 lanes per unit
// =============================================================================================
template <int K, bool ONE_OFFSET, bool BWD, bool FAST, int LANES = TQ_LANES_PER_UNIT>
__device__ __forceinline__ void tq_pixel_loop(TqPixAcc<K>& A, const tq_ksmogn_args& a,
                                              const float* __restrict__ s_tile, const float* __restrict__ s_fac,
                                              int r, int P, int npix, float b, const float* amph, float g, float rg,
                                              float ln_g, const float* W, const float* __restrict__ s_off = nullptr) {
  // s_off (offset histograms): [4] TqOffsetInfo, then the per-offset table of tq_pix_multi_offset_tab, in LDS
  const uint32_t magic = (1u << 20) / (uint32_t)P + 1u;  // exact pix / P for pix < 4096, P <= 64
  const float off0 = a.offset_samples[0];
  const float c0 = 0.5f * (float)(P - 1);
  TqOffsetInfo h;
  if (!ONE_OFFSET) {
    if (s_off) {
      h.dmin = s_off[0]; h.dmax = s_off[1]; h.lw2max = s_off[2]; h.lw2min = s_off[3];
    } else {
      tq_offset_info(a.offset_samples, a.offset_logits, a.O, &h);
    }
  }
  for (int pix = r; pix < npix; pix += LANES) {
    const int j = (int)(((uint32_t)pix * magic) >> 20);
    const int ic = pix - j * P;
    const float D = s_tile[pix];
    // spot-weighted moments are taken about the tile centre: |coordinate| <= P/2 instead of P keeps the cancellation
    // in the width gradient (second moment minus 2 w^2) four times smaller in fp32
    const float fic = (float)ic - c0, fj = (float)j - c0;
    float spot[K];
#pragma unroll
    for (int k = 0; k < K; ++k) spot[k] = amph[k] * s_fac[(2 * k) * TQ_MAX_P + ic] * s_fac[(2 * k + 1) * TQ_MAX_P + j];
    if (ONE_OFFSET) tq_pixel_one_offset<K, BWD, FAST>(A, D - off0, b, spot, W, fic, fj, g, rg, ln_g);
    else tq_pixel_multi_offset<K, BWD, FAST>(A, a, h, D, ln_g, b, spot, W, fic, fj, g, rg, s_off ? s_off + 4 : nullptr);
  }
}

#ifndef TQ_PIX_WAVES
#define TQ_PIX_WAVES 4
#endif
// The 16 units [blk * 16, blk * 16 + 16) of the batch by one workgroup of 256 threads; `smem` holds
// tq_tile16_lds_floats(P, K) floats.  Called by tq_ksmogn_kernel (tq_ksmogn.hip) and by the single-launch minibatch step
// (tq_cosmos.hip), which runs it between its sampling and per-unit phases.
#define TQ_OFFTAB_MAX 2048  /* offsets whose table the workgroup keeps in LDS (16 B each: 32 KB); longer histograms read global memory */
__host__ __device__ inline size_t tq_tile16_lds_floats(int P, int K, int O = 1, int units = TQ_UNITS_PER_BLOCK) {
  return (size_t)units * (tq_tile_stride(P * P) + 2 * K * TQ_MAX_P) + ((O > 1 && O <= TQ_OFFTAB_MAX) ? 4 + 4 * (size_t)O : 0);
}
// LANES lanes per unit (16: sixteen units per workgroup of 256 threads; 64: four units, a wave each -- the offset-histogram
// form of gathered batches, whose offset loop is long enough to want every wave slot of the chip: a 10 x 512 minibatch is
// 5120 units = 1280 waves at 16 lanes per unit, 5120 at 64).
// tq_ksmogn_tile_at: the TQ_BLOCK / LANES units [i0, i0 + TQ_BLOCK / LANES) of the batch below i_end; LDS_UNITS = unit slots of
// the LDS layout (tiles, factor tables, then the offset table): a caller that runs a 16-lane pass and then a 64-lane pass over
// the same LDS (the minibatch step with 20 units per workgroup) keeps the layout of 16 slots for both -- the four units of
// the second pass use slots 0, 4, 8, 12, which the SAME wave owned in the first pass, so no workgroup barrier separates the
// passes -- and the offset table built by the first pass (TAB_READY).
template <int K, bool ONE_OFFSET, bool BWD, int LANES, int LDS_UNITS, bool TAB_READY>
__device__ __forceinline__ void tq_ksmogn_tile_at(const tq_ksmogn_args& a, const int64_t B, const int64_t i0, const int64_t i_end, float* smem) {
  constexpr int M = 1 << K;
  constexpr int UNITS = TQ_BLOCK / LANES;
  static_assert(LDS_UNITS % UNITS == 0, "slots of a pass are a stride of the layout's");

  const int tid = threadIdx.x;
  const int grp = tid / LANES;
  const int r = tid % LANES;
  const int64_t i_raw = i0 + grp;
  const bool live = i_raw < i_end;
  const int64_t i = live ? i_raw : (B - 1);  // dead groups shadow the last unit, stores masked

  // ---- decode minibatch position -> dataset unit -------------------------------------------
  const int P = a.P;
  const int npix = P * P;
  const int stride = tq_tile_stride(npix);
  const int slot = grp * (LDS_UNITS / UNITS);
  float* s_tile = smem + slot * stride;
  float* s_fac = smem + LDS_UNITS * stride + slot * (2 * K * TQ_MAX_P);
  // (B < 2^31 is checked on the host; 32-bit arithmetic, and no division at all for a contiguous batch)
  const uint32_t iu = (uint32_t)i;
  int n;
  int64_t u, u_img;  // dataset unit; unit whose tile `images` holds at that position (differs for a streamed window)
  if (a.ndx == nullptr && a.fdx == nullptr) {
    u = u_img = i;
    n = (int)(iu / (uint32_t)(a.F * a.C));
  } else {
    const uint32_t c = iu % (uint32_t)a.C, ab = iu / (uint32_t)a.C;
    const uint32_t bi = ab % (uint32_t)a.fb, ai = ab / (uint32_t)a.fb;
    n = a.ndx ? a.ndx[ai] : (int)ai;
    const int f = a.fdx ? a.fdx[bi] : (int)bi;
    u = ((int64_t)n * a.F + f) * a.C + c;
    u_img = a.images_by_slot ? ((int64_t)ai * a.F + f) * a.C + c : u;
  }

  // ---- stage the P x P tile: all loads of a unit in flight at once, 16 B per lane when aligned ----
  const float* tile = a.images + u_img * npix;
  if ((npix & 3) == 0) {
    const float4* t4 = reinterpret_cast<const float4*>(tile);
    const int n4 = npix >> 2;
    for (int e = r; e < n4; e += LANES) *reinterpret_cast<float4*>(s_tile + 4 * e) = t4[e];
  } else {
    for (int e = r; e < npix; e += LANES) s_tile[e] = tile[e];
  }

  const float g = a.gain[0];
  const float rg = TQ_FRCP(g);
  const float ln_g = TQ_FLOG(g);
  const float tx = a.xy[2 * u], ty = a.xy[2 * u + 1];
  const float b = a.background[i];

  float hk[K], wk[K], amph[K], cx[K], cy[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    hk[k] = a.height[k * B + i];
    wk[k] = a.width[k * B + i];
    cx[k] = a.x[k * B + i] + tx;
    cy[k] = a.y[k * B + i] + ty;
    const float inv2v = 0.5f * TQ_FRCP(wk[k] * wk[k]);
    amph[k] = hk[k] * inv2v * (1.0f / TQ_PI);  // h / (2 pi w^2)
    // separable factors: lanes 0..15 fill the 2P entries of spot k
    for (int e = r; e < 2 * P; e += LANES) {
      const int axis = e >= P;
      const int p = e - axis * P;
      const float d = (float)p - (axis ? cy[k] : cx[k]);
      s_fac[(2 * k + axis) * TQ_MAX_P + p] = TQ_FEXP(-d * d * inv2v);
    }
  }

  float W[M];
  if (BWD) tq_load_weights<K>(a, B, i, u, n, W);

  // offset histogram: extrema and per-offset constants once per workgroup (every thread of the workgroup is here)
  float* s_off = nullptr;
  if (!ONE_OFFSET && a.O <= TQ_OFFTAB_MAX) s_off = smem + LDS_UNITS * (stride + 2 * K * TQ_MAX_P);
  if (!ONE_OFFSET && a.O <= TQ_OFFTAB_MAX && !TAB_READY) {
    if (tid < 64) {  // min / max are exact: the values of tq_offset_info
      float lo = INFINITY, hi = -INFINITY, lw = -INFINITY, lwn = INFINITY;
      for (int o = tid; o < a.O; o += 64) {
        const float sv = a.offset_samples[o], lv = a.offset_logits[o];
        lo = fminf(lo, sv); hi = fmaxf(hi, sv); lw = fmaxf(lw, lv); lwn = fminf(lwn, lv);
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, d, 64)); hi = fmaxf(hi, __shfl_xor(hi, d, 64));
        lw = fmaxf(lw, __shfl_xor(lw, d, 64)); lwn = fminf(lwn, __shfl_xor(lwn, d, 64));
      }
      if (tid == 0) {
        s_off[0] = lo; s_off[1] = hi; s_off[2] = lw * TQ_LOG2E; s_off[3] = lwn * TQ_LOG2E;
      }
    }
    __syncthreads();
    const float dmin = s_off[0], lw2max = s_off[2], beta2 = rg * TQ_LOG2E;
    for (int o = tid; o < a.O; o += TQ_BLOCK) {  // the expressions of tq_pix_multi_offset
      const float sv = a.offset_samples[o];
      const float dd = sv - dmin;
      s_off[4 + 4 * o] = sv;
      s_off[4 + 4 * o + 1] = dd;
      s_off[4 + 4 * o + 2] = (a.offset_logits[o] * TQ_LOG2E - lw2max) + beta2 * dd;
    }
    __syncthreads();
  }

  // the tile and the factor table of a unit are written and read by the same 16 lanes of ONE wave:
  // LDS operations of a wave complete in order, so no workgroup barrier is needed
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- pixel loop -----------------------------------------------------------------------------
  TqPixAcc<K> A;
  tq_acc_zero<K>(A);
  // alpha(m) >= background / gain for every combination and pixel: one wave-uniform test picks
  // the branch-free loop (one-term Binet correction valid) or the general one
  if (__all(b * rg >= TQ_FAST_ALPHA)) tq_pixel_loop<K, ONE_OFFSET, BWD, true, LANES>(A, a, s_tile, s_fac, r, P, npix, b, amph, g, rg, ln_g, W, s_off);
  else tq_pixel_loop<K, ONE_OFFSET, BWD, false, LANES>(A, a, s_tile, s_fac, r, P, npix, b, amph, g, rg, ln_g, W, s_off);

  // ---- reduce over the unit's 16 lanes, assemble and store ------------------------------------------
#pragma unroll
  for (int mi = 0; mi < M; ++mi) {
    A.ll[mi] = tq_group_sum<LANES>(A.ll[mi]);
    if (ONE_OFFSET) {
      A.sl[mi] = tq_group_sum<LANES>(A.sl[mi]);
      A.sS[mi] = tq_group_sum<LANES>(A.sS[mi]);
    }
  }
  if (BWD) {
    A.acc_b = tq_group_sum<LANES>(A.acc_b);
    if (!ONE_OFFSET) A.acc_g = tq_group_sum<LANES>(A.acc_g);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      A.S0[k] = tq_group_sum<LANES>(A.S0[k]);
      A.Sx[k] = tq_group_sum<LANES>(A.Sx[k]);
      A.Sy[k] = tq_group_sum<LANES>(A.Sy[k]);
      A.Sr[k] = tq_group_sum<LANES>(A.Sr[k]);
    }
  }
  float S_v = 0.0f;
  bool bad = false;
  if (ONE_OFFSET) {
#pragma unroll
    for (int k = 0; k < K; ++k) A.SN[k] = tq_group_sum<LANES>(A.SN[k]);
    S_v = a.pixstats[u];
    const float S_lv = a.pixstats[a.stats_stride + u];
    bad = a.pixstats[2 * a.stats_stride + u] > 0.0f;
    tq_pixel_assemble_one_offset<K, BWD>(a, A, W, b, g, rg, ln_g, (float)npix, S_v, S_lv);
  }
  if (live && r == 0) {
    float cxs[K], cys[K];  // the moments were taken about the tile centre (tq_pixel_loop)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      cxs[k] = cx[k] - 0.5f * (float)(P - 1);
      cys[k] = cy[k] - 0.5f * (float)(P - 1);
    }
    tq_pixel_store<K, ONE_OFFSET, BWD>(a, B, i, A, W, b, rg, hk, wk, cxs, cys, (float)npix, S_v, bad);
  }
}

template <int K, bool ONE_OFFSET, bool BWD, int LANES = TQ_LANES_PER_UNIT>
__device__ __forceinline__ void tq_ksmogn_tile16(const tq_ksmogn_args& a, const int64_t B, const int64_t blk, float* smem) {
  constexpr int UNITS = TQ_BLOCK / LANES;
  tq_ksmogn_tile_at<K, ONE_OFFSET, BWD, LANES, UNITS, false>(a, B, blk * UNITS, B, smem);
}

