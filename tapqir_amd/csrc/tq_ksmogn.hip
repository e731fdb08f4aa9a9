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

template <int CTRL>
__device__ __forceinline__ float tq_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float tq_group_sum16(float v) {
  // sum over the 16 lanes of a unit = one DPP row: data-parallel-primitive adds, no LDS crossbar.
  v += tq_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += tq_dpp<0x4E>(v);   // quad_perm [2,3,0,1]   -> quad sums
  v += tq_dpp<0x141>(v);  // row_half_mirror       -> sums of 8
  v += tq_dpp<0x140>(v);  // row_mirror            -> sum of 16, in every lane
  return v;
}
__device__ __forceinline__ float tq_fast_sigmoid(float u) { return TQ_FRCP(1.0f + TQ_FEXP(-u)); }

// LDS: [16 units][tile stride] staged pixels, then [16 units][2K][P] separable Gaussian factors.
// The tile stride is npix rounded up to 16 (mod 32) floats so that the two units sharing a
// 32-lane ds_read_b32 group hit disjoint bank halves.
__host__ __device__ inline int tq_tile_stride(int npix) { return ((npix + 15) / 32) * 32 + 16; }

// per-lane accumulators of the pixel loop
template <int K>
struct TqPixAcc {
  float ll[1 << K];   // general path: log-likelihood per combination.  Single-offset path: the
                      // "rest" part (1/2 ln alpha - S) for combinations >= 1 (entry 0 unused)
  float mphi[1 << K]; // single-offset path: sum mu*phi per combination (entry 0: sum phi of combination 0)
  float base;         // single-offset path: sum ln v (the combination-independent part)
  float acc_b;        // sum_m W_m da_m
  float acc_g;        // general path only: sum_m W_m [alpha (da+1) - E_o v / g]
  float sv, cnt;      // single-offset path: sum v, number of valid pixels
  float S0[K], Sx[K], Sy[K], Sr[K];  // spot-weighted moments: sum q*spot*{1, i, j, i^2+j^2}
  float SN[K];        // single-offset path: sum spot_k over valid pixels
  float bad;          // single-offset path: > 0 if the unit has a pixel at or below the offset (log 0)
};

// One pixel, single-offset path.  CHECK = some lane of the wave has D <= offset (masked pixel).
template <int K, bool BWD, bool FAST, bool CHECK>
__device__ __forceinline__ void tq_pixel_one_offset(TqPixAcc<K>& A, float v, bool ok, float ln_g,
                                                    float b, const float* spot, const float* W, float fic, float fj,
                                                    const TqCombo0& c0, float g, float rg) {
  constexpr int M = 1 << K;
  if (CHECK && !ok) {
    A.bad = 1.0f;  // log 0 for every combination; the pixel contributes no gradient
    return;
  }
  const float lv = TQ_FLOG(v);
  A.base += lv;
  const float lvg = lv - ln_g;
  float da[M];
  {
    float phi0;
    tq_pix_combo0(c0, v, lv, &phi0, &da[0]);
    A.mphi[0] += phi0;
  }
#pragma unroll
  for (int mi = 1; mi < M; ++mi) {
    float mu = b;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if ((mi >> k) & 1) mu += spot[k];
    float mphi, rest;
    tq_pix_one_offset<FAST>(v, lvg, mu, rg, g, &mphi, &rest, &da[mi]);
    A.mphi[mi] += mphi;
    A.ll[mi] += rest;
  }
  if (BWD) {
    float q[K];
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = 0.0f;
#pragma unroll
    for (int mi = 0; mi < M; ++mi) {
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
      A.SN[k] += spot[k];
    }
    A.sv += v;
  }
  A.cnt += 1.0f;
}

// One pixel, general path: online log-sum-exp over the offset samples.
template <int K, bool BWD, bool FAST>
__device__ __forceinline__ void tq_pixel_multi_offset(TqPixAcc<K>& A, const tq_ksmogn_args& a, float D, float ln_g,
                                                      float b, const float* spot, const float* W, float fic,
                                                      float fj, float g, float rg) {
  constexpr int M = 1 << K;
  TqComboPix cp[M];
  TqLse acc[M];
#pragma unroll
  for (int mi = 0; mi < M; ++mi) {
    float mu = b;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if ((mi >> k) & 1) mu += spot[k];
    tq_combo_prepare(mu, rg, g, ln_g, &cp[mi]);
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
  float q[K];
#pragma unroll
  for (int k = 0; k < K; ++k) q[k] = 0.0f;
#pragma unroll
  for (int mi = 0; mi < M; ++mi) {
    float lp, da, gq;
    tq_lse_finish<FAST>(acc[mi], cp[mi], rg, &lp, &da, &gq);
    A.ll[mi] += lp;
    if (BWD) {
      const float cw = W[mi] * da;
      A.acc_b += cw;
      A.acc_g += W[mi] * gq;
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

template <int K, bool ONE_OFFSET, bool BWD, bool FAST>
__device__ __forceinline__ void tq_pixel_loop(TqPixAcc<K>& A, const tq_ksmogn_args& a,
                                              const float* __restrict__ s_tile, const float* __restrict__ s_fac,
                                              int r, int P, int npix, float b, const float* amph, float g, float rg,
                                              float ln_g, const float* W, const TqCombo0& c0) {
  const uint32_t magic = (1u << 20) / (uint32_t)P + 1u;  // exact pix / P for pix < 4096, P <= 64
  const float off0 = a.offset_samples[0];

  for (int pix = r; pix < npix; pix += TQ_LANES_PER_UNIT) {
    const int j = (int)(((uint32_t)pix * magic) >> 20);
    const int ic = pix - j * P;
    const float D = s_tile[pix];
    const float fic = (float)ic, fj = (float)j;
    float spot[K];
#pragma unroll
    for (int k = 0; k < K; ++k) spot[k] = amph[k] * s_fac[(2 * k) * TQ_MAX_P + ic] * s_fac[(2 * k + 1) * TQ_MAX_P + j];

    if (ONE_OFFSET) {
      const float v = D - off0;
      const bool ok = v > 0.0f;
      // every pixel of real data exceeds the offset (glimpse_reader.py:407-411); one wave-uniform
      // test keeps the per-lane predication out of the common path
      if (__all(ok)) tq_pixel_one_offset<K, BWD, FAST, false>(A, v, ok, ln_g, b, spot, W, fic, fj, c0, g, rg);
      else tq_pixel_one_offset<K, BWD, FAST, true>(A, v, ok, ln_g, b, spot, W, fic, fj, c0, g, rg);
    } else {
      tq_pixel_multi_offset<K, BWD, FAST>(A, a, D, ln_g, b, spot, W, fic, fj, g, rg);
    }
  }
}

#ifndef TQ_PIX_WAVES
#define TQ_PIX_WAVES 4
#endif
template <int K, bool ONE_OFFSET, bool BWD>
__global__ __launch_bounds__(TQ_BLOCK, TQ_PIX_WAVES) void tq_ksmogn_kernel(const tq_ksmogn_args a, const int64_t B) {
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
  // (B < 2^31 is checked on the host; 32-bit arithmetic, and no division at all for a contiguous batch)
  const uint32_t iu = (uint32_t)i;
  int n;
  int64_t u;
  if (a.ndx == nullptr && a.fdx == nullptr) {
    u = i;
    n = (int)(iu / (uint32_t)(a.F * a.C));
  } else {
    const uint32_t c = iu % (uint32_t)a.C, ab = iu / (uint32_t)a.C;
    const uint32_t bi = ab % (uint32_t)a.fb, ai = ab / (uint32_t)a.fb;
    n = a.ndx ? a.ndx[ai] : (int)ai;
    const int f = a.fdx ? a.fdx[bi] : (int)bi;
    u = ((int64_t)n * a.F + f) * a.C + c;
  }

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
  // the tile and the factor table of a unit are written and read by the same 16 lanes of ONE wave:
  // LDS operations of a wave complete in order, so no workgroup barrier is needed
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- pixel loop -----------------------------------------------------------------------------
  TqPixAcc<K> A;
#pragma unroll
  for (int mi = 0; mi < M; ++mi) A.ll[mi] = A.mphi[mi] = 0.0f;
  A.base = A.acc_b = A.acc_g = A.sv = A.cnt = A.bad = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) A.S0[k] = A.Sx[k] = A.Sy[k] = A.Sr[k] = A.SN[k] = 0.0f;
  TqCombo0 c0;
  if (ONE_OFFSET) tq_combo0_prepare(b, rg, g, ln_g, &c0);

  // alpha(m) >= background / gain for every combination and pixel: one wave-uniform test picks
  // the branch-free loop (one-term Binet correction valid) or the general one
  if (__all(b * rg >= TQ_FAST_ALPHA))
    tq_pixel_loop<K, ONE_OFFSET, BWD, true>(A, a, s_tile, s_fac, r, P, npix, b, amph, g, rg, ln_g, W, c0);
  else
    tq_pixel_loop<K, ONE_OFFSET, BWD, false>(A, a, s_tile, s_fac, r, P, npix, b, amph, g, rg, ln_g, W, c0);

  // ---- reduce over the unit's 16 lanes and store ------------------------------------------------
  if (ONE_OFFSET) {
    // log p = [ln w - ln sqrt(2pi) - ln v] + (1/g) mu phi + rest ; combination 0: alpha0 phi + c_lp
    const float lw0 = a.offset_logits[0] - TQ_LN_SQRT_2PI;
    const float common = lw0 * A.cnt - A.base;
    A.ll[0] = common + c0.alpha * A.mphi[0] + c0.c_lp * A.cnt;
#pragma unroll
    for (int mi = 1; mi < M; ++mi) A.ll[mi] += common + rg * A.mphi[mi];
    A.bad = tq_group_sum16(A.bad);
  }
#pragma unroll
  for (int mi = 0; mi < M; ++mi) A.ll[mi] = tq_group_sum16(A.ll[mi]);
  if (ONE_OFFSET && A.bad > 0.0f) {
#pragma unroll
    for (int mi = 0; mi < M; ++mi) A.ll[mi] = -INFINITY;
  }
  if (BWD) {
    A.acc_b = tq_group_sum16(A.acc_b);
    if (ONE_OFFSET) {
      A.sv = tq_group_sum16(A.sv);
      A.cnt = tq_group_sum16(A.cnt);
    } else {
      A.acc_g = tq_group_sum16(A.acc_g);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      A.S0[k] = tq_group_sum16(A.S0[k]);
      A.Sx[k] = tq_group_sum16(A.Sx[k]);
      A.Sy[k] = tq_group_sum16(A.Sy[k]);
      A.Sr[k] = tq_group_sum16(A.Sr[k]);
      if (ONE_OFFSET) A.SN[k] = tq_group_sum16(A.SN[k]);
    }
  }
  if (live && r == 0) {
#pragma unroll
    for (int mi = 0; mi < M; ++mi) a.ll[(int64_t)mi * B + i] = A.ll[mi];
    if (BWD) {
      // d alpha = d mu / g for every mu-parameter
      a.g_background[i] = A.acc_b * rg;
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
        acc_g = rg * (mu_da + mu_w + Wsum * (b * A.cnt - A.sv));
      }
      a.g_gain[i] = -acc_g * rg;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float rw = TQ_FRCP(wk[k]);
        const float S1x = A.Sx[k] - cx[k] * A.S0[k];
        const float S1y = A.Sy[k] - cy[k] * A.S0[k];
        const float S2 = A.Sr[k] - 2.0f * (cx[k] * A.Sx[k] + cy[k] * A.Sy[k]) + (cx[k] * cx[k] + cy[k] * cy[k]) * A.S0[k];
        a.g_height[k * B + i] = A.S0[k] * rg * TQ_FRCP(hk[k]);
        a.g_x[k * B + i] = rg * S1x * rw * rw;
        a.g_y[k * B + i] = rg * S1y * rw * rw;
        a.g_width[k * B + i] = rg * (S2 * rw * rw * rw - 2.0f * A.S0[k] * rw);
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
  if (a->K < 1 || a->K > TQ_MAX_K || a->P < 2 || a->P > TQ_MAX_P || a->O < 1 || a->nb < 1 || a->fb < 1 || a->C < 1 ||
      (int64_t)a->nb * a->fb * a->C >= (int64_t)1 << 31) {
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
