// tq_ksmogn.hip -- fused spot render + offset-marginalised Gamma log-likelihood (+ backward)
// for CDNA4 / gfx950.  Replaces tapqir/distributions/ksmogn.py:146-238 and
// tapqir/distributions/util.py:15-64 (see include/tapqir_hip.h).
//
// Mapping.  wave64 = 4 units x 16 lanes; a 256-thread workgroup owns 16 consecutive units, so
// the 16 P*P tiles it reads are one contiguous span of the (Nt,F,C,P,P) tensor when the
// minibatch is contiguous.  Lane r of a unit walks pixels r, r+16, ... (P=14: 13 trips, 94 %
// of lanes busy; P=20: 25 trips, 100 %).  The (2^K, ..., K, P, P) Gaussian stack of the
// reference is never materialised: per unit the 2*K*P separable factors exp(-(p-c)^2/2w^2)
// are computed once into LDS (2P exps per spot instead of P^2) and every pixel forms
// mu(m) = b + sum_{k in m} A_k Gx_k[i] Gy_k[j] for all 2^K presence combinations in registers.
// Pixel sums (log-likelihood per combination + the weighted gradient moments) are reduced
// over the 16 lanes with wave shuffles; no atomics, no global scratch.
//
// Not MFMA work: there is no contraction, only transcendental-heavy pointwise math and a
// 196-term reduction; the roofline that bounds it is HBM (algorithmic 844 B/unit at
// K=2,P=14) vs the quarter-rate transcendental pipe -- see DESIGN.md.
#include <hip/hip_runtime.h>

#include "../../include/tapqir_hip.h"
#include "tq_pixel.h"

#define TQ_LANES_PER_UNIT 16
#define TQ_UNITS_PER_BLOCK 16
#define TQ_BLOCK (TQ_LANES_PER_UNIT * TQ_UNITS_PER_BLOCK)

__device__ __forceinline__ float tq_group_sum16(float v) {
  // butterfly over the 16 lanes of a unit (rows of 16 never straddle a wave)
  v += __shfl_xor(v, 1, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 8, 16);
  return v;
}

// LDS: [16 units][tile stride] staged pixels, then [16 units][2K][P] separable Gaussian factors.
// The tile stride is npix rounded up to 16 (mod 32) floats so that the two units sharing a
// 32-lane ds_read_b32 group hit disjoint bank halves.
__host__ __device__ inline int tq_tile_stride(int npix) { return ((npix + 15) / 32) * 32 + 16; }

template <int K, bool ONE_OFFSET, bool BWD, bool FAST>
__device__ __forceinline__ void tq_pixel_loop(const tq_ksmogn_args& a, const float* __restrict__ s_tile,
                                              const float* __restrict__ s_fac, int r, int P, int npix, float b,
                                              const float* hk, const float* amp, const float* cx, const float* cy,
                                              float g, float rg, float ln_g, const float* W, float* ll, float& acc_b,
                                              float& acc_g, float* S0, float* S1x, float* S1y, float* S2) {
  constexpr int M = 1 << K;
  const uint32_t magic = (1u << 20) / (uint32_t)P + 1u;  // exact pix / P for pix < 4096, P <= 64
  const float off0 = a.offset_samples[0];
  const float lw0 = a.offset_logits[0] - TQ_LN_SQRT_2PI;
  TqCombo0 c0;
  if (ONE_OFFSET) tq_combo0_prepare(b, rg, g, ln_g, &c0);

  for (int pix = r; pix < npix; pix += TQ_LANES_PER_UNIT) {
    const int j = (int)(((uint32_t)pix * magic) >> 20);
    const int ic = pix - j * P;
    const float D = s_tile[pix];

    float spot[K], spotn[K], dx[K], dy[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      spotn[k] = amp[k] * s_fac[(2 * k) * TQ_MAX_P + ic] * s_fac[(2 * k + 1) * TQ_MAX_P + j];
      spot[k] = hk[k] * spotn[k];
      dx[k] = (float)ic - cx[k];
      dy[k] = (float)j - cy[k];
    }
    float mu[M];
#pragma unroll
    for (int mi = 0; mi < M; ++mi) {
      float m_ = b;
#pragma unroll
      for (int k = 0; k < K; ++k)
        if ((mi >> k) & 1) m_ += spot[k];
      mu[mi] = m_;
    }

    float lp[M], da[M], gq[M];
    if (ONE_OFFSET) {
      const float v = D - off0;
      if (v > 0.0f) {
        const float lv = TQ_FLOG(v);
        const float base = lw0 - lv;
        tq_pix_combo0(c0, v, lv, base, &lp[0], &da[0], &gq[0]);
#pragma unroll
        for (int mi = 1; mi < M; ++mi)
          tq_pix_one_offset<FAST>(v, lv, base, mu[mi], rg, g, ln_g, &lp[mi], &da[mi], &gq[mi]);
      } else {
#pragma unroll
        for (int mi = 0; mi < M; ++mi) {
          lp[mi] = -INFINITY;
          da[mi] = 0.0f;
          gq[mi] = 0.0f;
        }
      }
    } else {
      TqComboPix cp[M];
      TqLse acc[M];
#pragma unroll
      for (int mi = 0; mi < M; ++mi) {
        tq_combo_prepare(mu[mi], rg, g, ln_g, &cp[mi]);
        tq_lse_init(&acc[mi]);
      }
      for (int o = 0; o < a.O; ++o) {
        const float v = D - a.offset_samples[o];
        if (v > 0.0f) {  // ksmogn.py:226 / KeOps Step(x - g - 1): offsets at or above the pixel are excluded
          const float lv = TQ_FLOG(v);
          const float lwl = a.offset_logits[o] - lv;
#pragma unroll
          for (int mi = 0; mi < M; ++mi) tq_lse_push(&acc[mi], cp[mi], v, lv, lwl);
        }
      }
#pragma unroll
      for (int mi = 0; mi < M; ++mi) tq_lse_finish<FAST>(acc[mi], cp[mi], rg, &lp[mi], &da[mi], &gq[mi]);
    }

#pragma unroll
    for (int mi = 0; mi < M; ++mi) ll[mi] += lp[mi];

    if (BWD) {
      float q[K];
#pragma unroll
      for (int k = 0; k < K; ++k) q[k] = 0.0f;
#pragma unroll
      for (int mi = 0; mi < M; ++mi) {
        const float cw = W[mi] * da[mi];
        acc_b += cw;
        acc_g += W[mi] * gq[mi];
#pragma unroll
        for (int k = 0; k < K; ++k)
          if ((mi >> k) & 1) q[k] += cw;
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float aq = q[k] * spotn[k];
        S0[k] += aq;
        S1x[k] += aq * dx[k];
        S1y[k] += aq * dy[k];
        S2[k] += aq * (dx[k] * dx[k] + dy[k] * dy[k]);
      }
    }
  }
}

template <int K, bool ONE_OFFSET, bool BWD>
__global__ __launch_bounds__(TQ_BLOCK) void tq_ksmogn_kernel(const tq_ksmogn_args a, const int64_t B) {
  constexpr int M = 1 << K;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int grp = tid >> 4;
  const int r = tid & 15;
  const int64_t i_raw = (int64_t)blockIdx.x * TQ_UNITS_PER_BLOCK + grp;
  const bool live = i_raw < B;
  const int64_t i = live ? i_raw : (B - 1);  // dead groups shadow the last unit, stores masked

  // ---- decode minibatch position -> dataset unit -------------------------------------------
  const int P = a.P;
  const int npix = P * P;
  const int stride = tq_tile_stride(npix);
  float* s_tile = smem + grp * stride;
  float* s_fac = smem + TQ_UNITS_PER_BLOCK * stride + grp * (2 * K * TQ_MAX_P);
  const int c = (int)(i % a.C);
  const int64_t ab = i / a.C;
  const int bi = (int)(ab % a.fb);
  const int ai = (int)(ab / a.fb);
  const int n = a.ndx ? a.ndx[ai] : ai;
  const int f = a.fdx ? a.fdx[bi] : bi;
  const int64_t u = ((int64_t)n * a.F + f) * a.C + c;

  // ---- stage the P x P tile: all loads of a unit in flight at once, 16 B per lane when aligned ----
  const float* tile = a.images + u * npix;
  if ((npix & 3) == 0) {
    const float4* t4 = reinterpret_cast<const float4*>(tile);
    const int n4 = npix >> 2;
    for (int e = r; e < n4; e += TQ_LANES_PER_UNIT) *reinterpret_cast<float4*>(s_tile + 4 * e) = t4[e];
  } else {
    for (int e = r; e < npix; e += TQ_LANES_PER_UNIT) s_tile[e] = tile[e];
  }

  const float g = a.gain[0];
  const float rg = 1.0f / g;
  const float ln_g = logf(g);
  const float tx = a.xy[2 * u], ty = a.xy[2 * u + 1];
  const float b = a.background[i];

  float hk[K], wk[K], amp[K], cx[K], cy[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    hk[k] = a.height[k * B + i];
    wk[k] = a.width[k * B + i];
    cx[k] = a.x[k * B + i] + tx;
    cy[k] = a.y[k * B + i] + ty;
    const float inv2v = 0.5f / (wk[k] * wk[k]);
    amp[k] = inv2v * (1.0f / TQ_PI);  // 1 / (2 pi w^2)
    // separable factors: lanes 0..15 fill the 2P entries of spot k
    for (int e = r; e < 2 * P; e += TQ_LANES_PER_UNIT) {
      const int axis = e >= P;
      const int p = e - axis * P;
      const float d = (float)p - (axis ? cy[k] : cx[k]);
      s_fac[(2 * k + axis) * TQ_MAX_P + p] = TQ_FEXP(-d * d * inv2v);
    }
  }

  // ---- upstream weights ---------------------------------------------------------------------
  float W[M];
  if (BWD) {
    if (a.gout) {
#pragma unroll
      for (int mi = 0; mi < M; ++mi) W[mi] = a.gout[(int64_t)mi * B + i];
    } else {
      float p1[K], p0[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float uk = a.m_logit[k * a.m_kstride + u];
        p1[k] = tq_sigmoid(uk);
        p0[k] = tq_sigmoid(-uk);
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
  __syncthreads();

  // ---- pixel loop -----------------------------------------------------------------------------
  float ll[M];
#pragma unroll
  for (int mi = 0; mi < M; ++mi) ll[mi] = 0.0f;
  float acc_b = 0.0f, acc_g = 0.0f;
  float S0[K], S1x[K], S1y[K], S2[K];
#pragma unroll
  for (int k = 0; k < K; ++k) S0[k] = S1x[k] = S1y[k] = S2[k] = 0.0f;

  // alpha(m) >= background / gain for every combination and pixel: one wave-uniform test picks
  // the branch-free loop (Binet series valid) or the general one
  if (__all(b * rg >= 8.0f))
    tq_pixel_loop<K, ONE_OFFSET, BWD, true>(a, s_tile, s_fac, r, P, npix, b, hk, amp, cx, cy, g, rg, ln_g, W, ll,
                                            acc_b, acc_g, S0, S1x, S1y, S2);
  else
    tq_pixel_loop<K, ONE_OFFSET, BWD, false>(a, s_tile, s_fac, r, P, npix, b, hk, amp, cx, cy, g, rg, ln_g, W, ll,
                                             acc_b, acc_g, S0, S1x, S1y, S2);

  // ---- reduce over the unit's 16 lanes and store ------------------------------------------------
#pragma unroll
  for (int mi = 0; mi < M; ++mi) ll[mi] = tq_group_sum16(ll[mi]);
  if (BWD) {
    acc_b = tq_group_sum16(acc_b);
    acc_g = tq_group_sum16(acc_g);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      S0[k] = tq_group_sum16(S0[k]);
      S1x[k] = tq_group_sum16(S1x[k]);
      S1y[k] = tq_group_sum16(S1y[k]);
      S2[k] = tq_group_sum16(S2[k]);
    }
  }
  if (live && r == 0) {
#pragma unroll
    for (int mi = 0; mi < M; ++mi) a.ll[(int64_t)mi * B + i] = ll[mi];
    if (BWD) {
      // d alpha = d mu / g for every mu-parameter
      a.g_background[i] = acc_b * rg;
      a.g_gain[i] = -acc_g * rg;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float rw = 1.0f / wk[k];
        const float hs = hk[k] * rg;
        a.g_height[k * B + i] = S0[k] * rg;
        a.g_x[k * B + i] = hs * S1x[k] * rw * rw;
        a.g_y[k * B + i] = hs * S1y[k] * rw * rw;
        a.g_width[k * B + i] = hs * (S2[k] * rw * rw * rw - 2.0f * S0[k] * rw);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
extern "C" const char* tq_last_error(void) { return g_err; }
extern "C" int tq_version(void) { return 100; }
void tq_set_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }

template <int K>
static int launch_k(const tq_ksmogn_args& a, int64_t B, hipStream_t st) {
  const dim3 grid((unsigned)((B + TQ_UNITS_PER_BLOCK - 1) / TQ_UNITS_PER_BLOCK)), block(TQ_BLOCK);
  const bool bwd = a.g_background != nullptr;
  const bool one = a.O == 1;
  const size_t lds = sizeof(float) * TQ_UNITS_PER_BLOCK * (tq_tile_stride(a.P * a.P) + 2 * K * TQ_MAX_P);
  if (one && bwd) hipLaunchKernelGGL((tq_ksmogn_kernel<K, true, true>), grid, block, lds, st, a, B);
  else if (one) hipLaunchKernelGGL((tq_ksmogn_kernel<K, true, false>), grid, block, lds, st, a, B);
  else if (bwd) hipLaunchKernelGGL((tq_ksmogn_kernel<K, false, true>), grid, block, lds, st, a, B);
  else hipLaunchKernelGGL((tq_ksmogn_kernel<K, false, false>), grid, block, lds, st, a, B);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    tq_set_error(hipGetErrorString(e));
    return TQ_ERR_LAUNCH;
  }
  return TQ_OK;
}

extern "C" int tq_ksmogn_log_prob(const tq_ksmogn_args* a, void* stream) {
  if (!a || !a->images || !a->xy || !a->background || !a->height || !a->width || !a->x || !a->y || !a->gain ||
      !a->offset_samples || !a->offset_logits || !a->ll) {
    tq_set_error("tq_ksmogn_log_prob: NULL required pointer");
    return TQ_ERR_ARG;
  }
  if (a->K < 1 || a->K > TQ_MAX_K || a->P < 2 || a->P > TQ_MAX_P || a->O < 1 || a->nb < 1 || a->fb < 1 || a->C < 1) {
    tq_set_error("tq_ksmogn_log_prob: unsupported K/P/O or empty batch");
    return TQ_ERR_ARG;
  }
  const bool bwd = a->g_background != nullptr;
  if (bwd && (!a->g_height || !a->g_width || !a->g_x || !a->g_y || !a->g_gain || (!a->gout && !a->m_logit))) {
    tq_set_error("tq_ksmogn_log_prob: backward requested but a gradient output or the upstream weights are NULL");
    return TQ_ERR_ARG;
  }
  const int64_t B = (int64_t)a->nb * a->fb * a->C;
  hipStream_t st = (hipStream_t)stream;
  switch (a->K) {
    case 1: return launch_k<1>(*a, B, st);
    case 2: return launch_k<2>(*a, B, st);
    case 3: return launch_k<3>(*a, B, st);
    default: return launch_k<4>(*a, B, st);
  }
}
