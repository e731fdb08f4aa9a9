// tq_ksmogn.hip -- fused spot render + offset-marginalised Gamma log-likelihood (+ backward)
// for CDNA4 / gfx950.  Replaces tapqir/distributions/ksmogn.py:146-238 and
// tapqir/distributions/util.py:15-64 (see include/tapqir_hip.h).
//
// The (2^K, ..., K, P, P) Gaussian stack and the (..., P, P, O) mixture tensor of the reference are
// never materialised: every pixel forms mu(m) = b + sum_{k in m} A_k Gx_k[i] Gy_k[j] for all 2^K
// spot-presence combinations in registers, evaluates the log-density and (optionally) its gradient,
// and only per-unit sums leave the kernel.  Two mappings share the per-pixel code:
//
//   tq_ksmogn_il_kernel  (contiguous batches, the full-batch SVI step)
//       one LANE per unit on a tile-interleaved copy of the images: a wave64 reads its 64 tiles with
//       one fully coalesced 1 KiB global_load_dwordx4 per 4 pixels (next rows prefetched), no LDS, no
//       cross-lane reduction, wave-uniform pixel coordinates, unit-contiguous parameter loads/stores.
//   tq_ksmogn_kernel     (gathered minibatches, small batches, the plain KSMOGN.log_prob API)
//       16 lanes per unit, 16 units per 256-thread workgroup; the P x P tile is staged through LDS
//       with 16 B loads, the 2*K*P separable Gaussian factors are computed once per unit into LDS,
//       pixel sums are reduced over the 16 lanes of a DPP row.
//
// Not MFMA work: there is no contraction, only transcendental-heavy pointwise math and a
// P*P-term reduction; the roofline that bounds it is HBM (algorithmic 844 B/unit at K=2,P=14)
// vs the quarter-rate transcendental pipe -- see DESIGN.md.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "../../include/tapqir_hip.h"
#include "tq_dpp.h"
#include "tq_pixel.h"

#include "tq_ksmogn_dev.h"
#include "tq_ksmogn_il2.h"

template <int K, bool ONE_OFFSET, bool BWD, int LANES = TQ_LANES_PER_UNIT>
__global__ __launch_bounds__(TQ_BLOCK, TQ_PIX_WAVES) void tq_ksmogn_kernel(const tq_ksmogn_args a, const int64_t B) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  tq_ksmogn_tile16<K, ONE_OFFSET, BWD, LANES>(a, B, (int64_t)blockIdx.x, smem);
}


template <int K, bool ONE_OFFSET, bool BWD>
__global__ __launch_bounds__(256) void tq_ksmogn_il_kernel(const tq_ksmogn_args a, const int64_t B) {
  constexpr int M = 1 << K;
  const int64_t i_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i_raw < B;
  const int64_t i = live ? i_raw : (B - 1);  // idle lanes of the last wave shadow the last unit
  const int P = a.P, npix = P * P;
  const int npix4 = (npix + 3) >> 2;
  // idle lanes of the last workgroup read the LAST tile too: the interleaved buffer ends with its 64-tile block
  const float4* src = reinterpret_cast<const float4*>(a.images_il) + ((i >> 6) * npix4) * 64 + (i & 63);

  const float g = a.gain[0];
  const float rg = TQ_FRCP(g);
  const float ln_g = TQ_FLOG(g);
  const float tx = a.xy[2 * i], ty = a.xy[2 * i + 1];
  const float b = a.background[i];
  float hk[K], wk[K], amph[K], nl2[K], cx[K], cy[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    hk[k] = a.height[k * B + i];
    wk[k] = a.width[k * B + i];
    cx[k] = a.x[k * B + i] + tx;
    cy[k] = a.y[k * B + i] + ty;
    const float inv2v = 0.5f * TQ_FRCP(wk[k] * wk[k]);
    amph[k] = hk[k] * inv2v * (1.0f / TQ_PI);  // h / (2 pi w^2)
    nl2[k] = -inv2v * 1.44269504088896340736f;  // exp(-d^2/(2w^2)) = exp2(d^2 * nl2)
  }
  float W[M];
  if (BWD) tq_load_weights<K>(a, B, i, i, (int)((uint32_t)i / (uint32_t)(a.F * a.C)), W);

  TqPixAcc<K> A;
  tq_acc_zero<K>(A);
  if (__all(b * rg >= TQ_FAST_ALPHA)) tq_il_pixel_loop<K, ONE_OFFSET, BWD, true>(A, a, src, P, npix, b, amph, nl2, cx, cy, g, rg, ln_g, W);
  else tq_il_pixel_loop<K, ONE_OFFSET, BWD, false>(A, a, src, P, npix, b, amph, nl2, cx, cy, g, rg, ln_g, W);

  float S_v = 0.0f;
  bool bad = false;
  if (ONE_OFFSET) {
    S_v = a.pixstats[i];
    const float S_lv = a.pixstats[a.stats_stride + i];
    bad = a.pixstats[2 * a.stats_stride + i] > 0.0f;
    tq_pixel_assemble_one_offset<K, BWD>(a, A, W, b, g, rg, ln_g, (float)npix, S_v, S_lv);
  }
  if (live) {
    float cxs[K], cys[K];  // the moments were taken about the tile centre (tq_il_pixel_loop)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      cxs[k] = cx[k] - 0.5f * (float)(P - 1);
      cys[k] = cy[k] - 0.5f * (float)(P - 1);
    }
    tq_pixel_store<K, ONE_OFFSET, BWD>(a, B, i, A, W, b, rg, hk, wk, cxs, cys, (float)npix, S_v, bad);
  }
}

// =============================================================================================
// Packed-math variant of the lane-per-unit kernel (single offset, P in {14, 20}).
//
// Measured on MI355X (profiles/r01_pmc_*): a wave64 VALU instruction holds its SIMD for 4 cycles
// and instructions of different waves do not overlap, so these kernels are bound by the NUMBER of
// VALU instructions; the FP32 peak of the chip is only reachable with the packed v_pk_{fma,mul,add}_f32
// forms (two FP32 operations per lane per instruction).  Here every lane carries TWO horizontally
// adjacent pixels of its unit in float2 registers, so all the per-pixel algebra issues as packed
// instructions; only the transcendentals (exp2 for the x-factor, rcp and log2 per combination) are
// per element.  The loop body covers R rows (R = 2 for P = 14, 1 for P = 20) so that every pixel
// coordinate and every position in the float4 stream is a compile-time constant; the float4 holding a
// group of 4 pixels is re-loaded, right after its last use, with the same group of the next body
// (one body = 28 or 20 pixels of latency cover, G float4 registers).
// =============================================================================================

template <int K, int P, bool BWD>
__global__ __launch_bounds__(64, (K <= 3 ? 2 : 1)) void tq_ksmogn_il2_kernel(const tq_ksmogn_args a, const int64_t B) {
  tq_il2_lane<K, P, BWD>(a, B);
}

// =============================================================================================
// Persistent form of the packed kernel (backward pass of full-batch steps).
//
// tq_ksmogn_il2_kernel spends ~7 us per round of its one-wave workgroups outside the pixel loop (a fresh wave waits for
// its unit parameters, computes the x-factor table, runs the loop, waits again for the data statistics before it can
// store) and, inside the loop, waits for ALL image loads of a body at its top (the waits the compiler derives for
// register-destination loads degrade to vmcnt(0) as soon as the loop is not the peeled single-tile form).  Here
//   * a wave stays resident and walks over tiles (tile = 64 units);
//   * EVERY global read of the hot path is an LDS-DMA request (global_load_lds_*: memory -> LDS, no VGPRs) issued
//     through inline assembly, so the compiler neither tracks nor waits for them and the waits are written by hand:
//       - the image stream runs through a ring of 8 slots x 1 KiB in LDS, 8 groups (32 pixels per lane) ahead of the
//         arithmetic: a group is read back with one ds_read_b128 and its slot is refilled at once;
//       - the next tile's per-unit scalars (target position, draws, m_probs logits, data statistics: 6 + 5K dwords
//         per unit) stream into one of two wave-private slabs while the last body of the current tile runs.
//   * vector-memory operations complete in issue order, so "the request for group q has landed" is s_waitcnt vmcnt(N)
//     with N = the number of requests issued after it: always the 7 younger ring requests, plus the 6 + 5K scalar
//     requests inside the last body, plus the result stores of the previous tile in the first body.  The request
//     pattern is kept identical in the last tile (dummy requests to valid addresses) so that N never overstates.
// =============================================================================================
#define TQ_P_NPAR(K) (6 + 5 * (K))
#ifndef TQ_P_NSLOT
#define TQ_P_NSLOT 8  // ring slots = groups the image stream runs ahead
#endif
#ifndef TQ_P_WAVES
#define TQ_P_WAVES 2  // resident waves per SIMD the register allocation is sized for (K <= 3)
#endif
#ifndef TQ_P_COLACC
#define TQ_P_COLACC 1  // x-moments as per-column sums (K * P / 2 float2 registers for two packed instructions per pair and spot)
#endif

__device__ __forceinline__ uint32_t tq_lds_addr(const void* p) {
  return (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)p;
}
// memory -> LDS, 4 / 16 bytes per lane: lane l reads base + voff and writes LDS byte lds + l * 4 (16)
__device__ __forceinline__ void tq_dma4(const void* base, uint32_t voff, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds), "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void tq_dma16(const void* base, uint32_t voff, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void tq_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int K, int P, bool BWD>
__global__ __launch_bounds__(64, (K <= 3 ? TQ_P_WAVES : 1)) void tq_ksmogn_il2p_kernel(const tq_ksmogn_args a, const int64_t B,
                                                                              const int64_t Bn, const int ntiles) {
  // B = units of the batch (row stride of the [K][B] arrays); Bn <= B = units this launch covers (tiles 0 .. ntiles-1)
  static_assert(P % 2 == 0 && (P * P) % 4 == 0, "packed kernel needs an even tile side");
  constexpr int M = 1 << K;
  constexpr int R = ((P / 2) % 2) ? 2 : 1;  // rows per loop body so that the body starts on a float4 boundary
  constexpr int G = R * P / 4;              // float4 groups per body
  constexpr int NB = P / R;                 // bodies per tile
  constexpr int npix = P * P, NG = npix / 4;
  constexpr bool COLACC = BWD && (K * P <= 28) && TQ_P_COLACC;
  constexpr int NPAR = TQ_P_NPAR(K), NSLOT = TQ_P_NSLOT;
  constexpr int NSTORE = BWD ? M + 2 + 4 * K : M;  // result stores of a tile (vector-memory operations, in issue order)
  static_assert((NSLOT & (NSLOT - 1)) == 0 && NSLOT <= NG && NB >= 3, "ring slots: a power of two");
  // slab rows: 0 tx, 1 ty, 2 b, 3.. h[K], w[K], x[K], y[K], m_logit[K], then the 3 data statistics
  constexpr int R_H = 3, R_W = 3 + K, R_X = 3 + 2 * K, R_Y = 3 + 3 * K, R_M = 3 + 4 * K, R_S = 3 + 5 * K;
  __shared__ float4 s_ring[NSLOT][64];
  __shared__ float s_par[2][NPAR][64];
  const int lane = threadIdx.x;
  const uint32_t ring_lds = tq_lds_addr(&s_ring[0][0]), par_lds = tq_lds_addr(&s_par[0][0][0]);
  const float4* il = reinterpret_cast<const float4*>(a.images_il);

  const float g = a.gain[0];
  const float rg = TQ_FRCP(g);
  const float ln_g = TQ_FLOG(g);
  const float off0 = a.offset_samples[0];
  const float logit0 = a.offset_logits[0];
  TqFastConst c;
  {
    const float g2 = g * g, rl2 = 1.0f / TQ_LN2;
    c.ca = TQ_LN2 * rg;
    c.cb = 0.5f * TQ_LN2;
    c.s1 = g * (1.0f / 12.0f);
    c.s3 = -g2 * g * (1.0f / 360.0f);
    c.d1 = 0.5f * g * rl2;
    c.d2 = g2 * (1.0f / 12.0f) * rl2;
    c.d4 = -g2 * g2 * (1.0f / 120.0f) * rl2;
  }
  const float sc_plate = a.scale;

  // per-unit scalars of tile `t` -> slab `buf`: NPAR requests (lanes beyond Bn shadow the last unit)
  auto request_params = [&](const int t, const int buf) __attribute__((always_inline)) {
    const int64_t t0 = (int64_t)t * 64;
    const int64_t rem = Bn - 1 - t0;  // >= 0: t < ntiles
    const uint32_t l = (uint32_t)lane < (uint32_t)rem ? (uint32_t)lane : (uint32_t)rem;
    const uint32_t slab = par_lds + (uint32_t)buf * (NPAR * 256);
    auto put = [&](const float* row_base, const uint32_t off_bytes, const int row) { tq_dma4(row_base, off_bytes, slab + row * 256); };
    put(a.xy + 2 * t0, 8 * l, 0);
    put(a.xy + 2 * t0 + 1, 8 * l, 1);
    put(a.background + t0, 4 * l, 2);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      put(a.height + k * B + t0, 4 * l, R_H + k);
      put(a.width + k * B + t0, 4 * l, R_W + k);
      put(a.x + k * B + t0, 4 * l, R_X + k);
      put(a.y + k * B + t0, 4 * l, R_Y + k);
      put(a.m_logit + k * a.m_kstride + t0, 4 * l, R_M + k);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) put(a.pixstats + j * a.stats_stride + t0, 4 * l, R_S + j);
  };
  // group `j` (4 pixels of each of the 64 units) of tile `t` -> ring slot `slot`
  const uint32_t lane16 = (uint32_t)lane * 16;
  auto request_group = [&](const int t, const int j, const int slot) __attribute__((always_inline)) {
    tq_dma16(il + ((int64_t)t * NG + j) * 64, lane16, ring_lds + (uint32_t)slot * 1024);
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  request_params(t, 0);
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) request_group(t, j, j);
  tq_wait_vm<0>();  // once per wave; afterwards the counts below are exact or understate
  int buf = 0, slot0 = 0;  // slab of the current tile; ring slot of its group 0
  bool first = true;
  for (; t < ntiles; t += gridDim.x, buf ^= 1, slot0 = (slot0 + NG) & (NSLOT - 1)) {
    const int tn = t + (int)gridDim.x;
    const int tnx = tn < ntiles ? tn : t;  // no next tile: the same requests go to this tile's (valid) addresses
    const int64_t i_raw = (int64_t)t * 64 + lane;
    const bool live = i_raw < Bn;
    const int64_t i = live ? i_raw : (Bn - 1);
    // the slab of this tile was requested at the top of the previous tile's last body: G ring requests and that tile's
    // result stores are younger
    if (!first) tq_wait_vm<G + NSTORE>();
    first = false;
    const float (*sp)[64] = s_par[buf];
    const float tx = sp[0][lane], ty = sp[1][lane];
    const float b = sp[2][lane];
    float amph[K], nl2[K], cx[K], cy[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float hk = sp[R_H + k][lane], wk = sp[R_W + k][lane];
      cx[k] = sp[R_X + k][lane] + tx;
      cy[k] = sp[R_Y + k][lane] + ty;
      const float inv2v = 0.5f * TQ_FRCP(wk * wk);
      amph[k] = hk * inv2v * (1.0f / TQ_PI);
      nl2[k] = -inv2v * 1.44269504088896340736f;
    }
    float W[M];
#pragma unroll
    for (int mi = 0; mi < M; ++mi) W[mi] = 0.0f;
    if (BWD) {
      float p1[K], p0[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float uk = sp[R_M + k][lane];
        p1[k] = tq_fast_sigmoid(uk);
        p0[k] = tq_fast_sigmoid(-uk);
      }
      float sc = sc_plate;
      if (a.aoi_mask != nullptr) {  // (the host passes NULL when no AOI is masked: a register-destination load and its wait)
        const int n = (int)((uint32_t)i / (uint32_t)(a.F * a.C));
        sc = a.aoi_mask[n] ? sc_plate : 0.0f;
      }
#pragma unroll
      for (int mi = 0; mi < M; ++mi) {
        float w = sc;
#pragma unroll
        for (int k = 0; k < K; ++k) w *= ((mi >> k) & 1) ? p1[k] : p0[k];
        W[mi] = w;
      }
    }

    TqPixAcc<K> S;
    tq_acc_zero<K>(S);
    const bool fastpath = __all(b * rg >= TQ_FAST_ALPHA);
    if (fastpath) {
      tq_f2 ex[K][P / 2];
      float sum_ex[K], sum_gy[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        tq_f2 acc = tq2(0.0f);
#pragma unroll
        for (int ip = 0; ip < P / 2; ++ip) {
          const tq_f2 dx = (tq_f2){(float)(2 * ip), (float)(2 * ip + 1)} - cx[k];
          ex[k][ip] = tq2_exp2(dx * dx * nl2[k]);
          acc += ex[k][ip];
        }
        sum_ex[k] = acc.x + acc.y;
        sum_gy[k] = 0.0f;
      }
      TqPixAcc2<K, P, COLACC> A;
#pragma unroll
      for (int mi = 0; mi < M; ++mi) A.T[mi] = tq2(0.0f);
      A.acc_b = tq2(0.0f);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        A.S0r[k] = A.S0[k] = A.Sy[k] = A.Syy[k] = A.Sx[k] = A.Sxx[k] = tq2(0.0f);
        if (COLACC) {
#pragma unroll
          for (int ip = 0; ip < P / 2; ++ip) A.col[k][ip] = tq2(0.0f);
        }
      }
      // one loop body = R rows = G groups.  EXTRA = requests other than the ring's that are younger than the requests of
      // the first NSLOT groups of this body (the previous tile's stores in the first body, the next tile's scalars in the
      // last one); the requests of later groups of the body were issued after those
      auto run_body = [&](const int body, auto extra_tag) __attribute__((always_inline)) {
        constexpr int EXTRA = decltype(extra_tag)::value;
        tq_f2 d_lo = tq2(0.0f), d_hi = tq2(0.0f);  // the two pixel pairs of the group being consumed
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
          const float fj = (float)(body * R + rr);
          float agy[K], dyk[K];
#pragma unroll
          for (int k = 0; k < K; ++k) {
            dyk[k] = fj - cy[k];
            agy[k] = amph[k] * __builtin_amdgcn_exp2f(dyk[k] * dyk[k] * nl2[k]);
            sum_gy[k] += agy[k];
          }
#pragma unroll
          for (int ip = 0; ip < P / 2; ++ip) {
            const int pair = rr * (P / 2) + ip;  // pair index within the body (compile-time after unrolling)
            if ((pair & 1) == 0) {
              const int j = body * G + (pair >> 1);  // group of the tile
              const int slot = (slot0 + j) & (NSLOT - 1);
              if ((pair >> 1) < NSLOT) tq_wait_vm<NSLOT - 1 + EXTRA>();
              else tq_wait_vm<NSLOT - 1>();
              const tq_f2* rp = reinterpret_cast<const tq_f2*>(&s_ring[slot][lane]);
              d_lo = rp[0];
              d_hi = rp[1];
              asm volatile("" : "+v"(d_lo), "+v"(d_hi)::"memory");  // the read has completed: the slot is free
              const int jn = j + NSLOT;  // the group 8 ahead in the stream refills it (next tile's first groups at the end)
              request_group(jn < NG ? t : tnx, jn < NG ? jn : jn - NG, slot);
            }
            const tq_f2 v = ((pair & 1) ? d_hi : d_lo) - off0;
            tq_f2 spot[K];
#pragma unroll
            for (int k = 0; k < K; ++k) spot[k] = agy[k] * ex[k][ip];
            tq_pixel_pair<K, P, BWD, COLACC>(A, v, b, spot, W, ip, ip == 0, c.ca, c.cb, c.s1, c.s3, c.d1, c.d2, c.d4);
          }
          if (BWD) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
              if (!COLACC) A.S0[k] += A.S0r[k];
              A.Sy[k] += A.S0r[k] * dyk[k];
              A.Syy[k] += A.S0r[k] * (dyk[k] * dyk[k]);
            }
          }
        }
      };
      // first body: the previous tile's result stores are younger than the requests of its groups (all of them < 8);
      // last body: the next tile's scalar requests go first and are younger than the requests of its groups
      run_body(0, std::integral_constant<int, NSTORE>{});
#pragma unroll 1
      for (int body = 1; body < NB - 1; ++body) run_body(body, std::integral_constant<int, 0>{});
      request_params(tnx, buf ^ 1);
      run_body(NB - 1, std::integral_constant<int, NPAR>{});

#pragma unroll
      for (int mi = 1; mi < M; ++mi) S.sS[mi] = -(A.T[mi].x + A.T[mi].y);  // the assembly adds -sS; ll = sl = 0
#pragma unroll
      for (int k = 0; k < K; ++k) S.SN[k] = sum_ex[k] * sum_gy[k];  // sum over the tile of a separable spot
      if (BWD) {
        S.acc_b = (A.acc_b.x + A.acc_b.y) * TQ_LN2;  // da was carried in units of ln 2
#pragma unroll
        for (int k = 0; k < K; ++k) {
          tq_f2 s0 = A.S0[k], sx = A.Sx[k], sxx = A.Sxx[k];
          if (COLACC) {  // column sums folded with the column offsets from the spot's own centre
            s0 = sx = sxx = tq2(0.0f);
            float cxe = cx[k];
            asm volatile("" : "+v"(cxe));
#pragma unroll
            for (int ip = 0; ip < P / 2; ++ip) {
              const tq_f2 dxc = (tq_f2){(float)(2 * ip), (float)(2 * ip + 1)} - cxe;
              s0 += A.col[k][ip];
              sx += A.col[k][ip] * dxc;
              sxx += A.col[k][ip] * (dxc * dxc);
            }
          }
          S.S0[k] = (s0.x + s0.y) * TQ_LN2;
          S.Sx[k] = (sx.x + sx.y) * TQ_LN2;
          S.Sy[k] = (A.Sy[k].x + A.Sy[k].y) * TQ_LN2;
          S.Sr[k] = (sxx.x + sxx.y + A.Syy[k].x + A.Syy[k].y) * TQ_LN2;
        }
      }
    } else {
      // some unit of this tile has a small alpha = background / gain: general (scalar, exact Binet) loop on the tile in
      // memory.  The ring requests in flight are abandoned; afterwards the request pattern of a tile's end is re-issued
      // (scalars, then the next tile's first 8 groups) so that the counts of the next tile hold.
      tq_wait_vm<0>();
      const float4* src = il + ((int64_t)t * NG) * 64 + lane;
      tq_il_pixel_loop<K, true, BWD, false>(S, a, src, P, npix, b, amph, nl2, cx, cy, g, rg, ln_g, W);
      tq_wait_vm<0>();
      request_params(tnx, buf ^ 1);
#pragma unroll
      for (int j = 0; j < NSLOT; ++j) request_group(tnx, j, (slot0 + NG + j) & (NSLOT - 1));
    }
    const float S_v = sp[R_S][lane];
    const float S_lv = sp[R_S + 1][lane];
    const bool bad = sp[R_S + 2][lane] > 0.0f;
    tq_pixel_assemble_one_offset<K, BWD>(a, S, W, b, g, rg, ln_g, (float)npix, S_v, S_lv, &logit0);
    {
      float hk[K], wk[K], cxs[K], cys[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        hk[k] = sp[R_H + k][lane];
        wk[k] = sp[R_W + k][lane];
        // reference points of the moments: the packed loop centres y on the spot and x on the spot (column sums) or on
        // the tile centre; the scalar fallback loop uses the tile centre for both
        cxs[k] = fastpath ? (COLACC ? 0.0f : cx[k] - 0.5f * (float)(P - 1)) : cx[k] - 0.5f * (float)(P - 1);
        cys[k] = fastpath ? 0.0f : cy[k] - 0.5f * (float)(P - 1);
      }
      // (every tile has a live lane, so the NSTORE store instructions that the counts above rely on are issued whatever
      // the tile; idle lanes of the last tile are masked out of them)
      if (live) tq_pixel_store<K, true, BWD>(a, B, i, S, W, b, rg, hk, wk, cxs, cys, (float)npix, S_v, bad);
    }
  }
}

// =============================================================================================
// Packed lane-per-unit kernel for an offset HISTOGRAM (O > 1; real data, glimpse_reader.py:414-421).
// Same mapping as above (one lane per unit, two horizontally adjacent pixels per lane in float2
// registers); per pixel pair the loop over the offset samples runs the formulation of tq_pixel.h:
// per (offset, combination) one packed fma, one packed add, two exp2 and three packed accumulations.
// The per-offset constants (delta_o - delta_min, db_o) are tabulated once per workgroup in LDS and read
// back as wave-uniform broadcasts.  A pixel pair with masked offsets (some delta_o >= D) anywhere in
// the wave takes the scalar routine instead.
// =============================================================================================
#define TQ_MO_MAX_O 1024

// WFAC: the per-offset factor 2^db_o is pulled out of the exponential (t = 2^db_o * exp2(a dl + c)) and folded
// into the accumulations as a multiplier, which removes the packed add; valid while |db_o| stays far
// inside the fp32 exponent range (checked once per kernel from the histogram and the gain).
// Table entry of offset o: WFAC ? {2^db_o * (delta_o - delta_min), 2^db_o} : {delta_o - delta_min, db_o}.
template <int K, bool BWD, bool WFAC>
__device__ __forceinline__ void tq_pair_multi_offset(TqPixAcc<K>& A, const tq_ksmogn_args& a, const TqOffsetInfo& h,
                                                     const float2* __restrict__ s_tab, const bool fast, tq_f2 D,
                                                     float b, const tq_f2* spot, const float* W, tq_f2 fic, float fj,
                                                     float g, float rg, float ln_g) {
  constexpr int M = 1 << K;
  tq_f2 mu[M], lp[M], da[M], gq[M];
  mu[0] = tq2(b);
#pragma unroll
  for (int mi = 1; mi < M; ++mi) {
    const int hi = 31 - __builtin_clz(mi);
    mu[mi] = mu[mi & ~(1 << hi)] + spot[hi];
  }
  const tq_f2 vhi = D - h.dmin;
  const tq_f2 vlo = D - h.dmax;
  if (__all(vlo.x > 0.0f && vlo.y > 0.0f)) {
    // every offset is below both pixels in every lane: no masks
    const tq_f2 rvhi = tq2_rcp(vhi);
    tq_f2 av[M], cv[M], vs[M], S0[M], S1[M], S2[M];
#pragma unroll
    for (int mi = 0; mi < M; ++mi) {
      const tq_f2 t = mu[mi] - g;
      vs[mi] = (tq_f2){fminf(fmaxf(t.x, vlo.x), vhi.x), fminf(fmaxf(t.y, vlo.y), vhi.y)};
      av[mi] = mu[mi] * rg - 1.0f;
      cv[mi] = -av[mi] * tq2_log2(vs[mi] * rvhi) - (rg * TQ_LOG2E) * (vhi - vs[mi]);  // (as tq_mo_reference)
      S0[mi] = S1[mi] = S2[mi] = tq2(0.0f);
    }
#pragma unroll 4
    for (int o = 0; o < a.O; ++o) {
      const float so = a.offset_samples[o];
      const float2 tb = s_tab[o];
      const tq_f2 dl = tq2_log2((D - so) * rvhi);
      if (WFAC) {
        const tq_f2 wdl = dl * tb.y;
#pragma unroll
        for (int mi = 0; mi < M; ++mi) {
          const tq_f2 e = tq2_exp2(av[mi] * dl + cv[mi]);
          S0[mi] += e * tb.y;
          if (BWD) {
            S1[mi] += e * wdl;
            S2[mi] += e * tb.x;
          }
        }
      } else {
#pragma unroll
        for (int mi = 0; mi < M; ++mi) {
          const tq_f2 t = tq2_exp2((av[mi] * dl + cv[mi]) + tb.y);
          S0[mi] += t;
          if (BWD) {
            S1[mi] += t * dl;
            S2[mi] += t * tb.x;
          }
        }
      }
    }
#pragma unroll
    for (int mi = 0; mi < M; ++mi) {
      float l0, d0, q0, l1, d1, q1;
      tq_mo_finish(fast, mu[mi].x, vs[mi].x, vhi.x, S0[mi].x, S1[mi].x, S2[mi].x, h, g, rg, ln_g, &l0, &d0, &q0);
      tq_mo_finish(fast, mu[mi].y, vs[mi].y, vhi.y, S0[mi].y, S1[mi].y, S2[mi].y, h, g, rg, ln_g, &l1, &d1, &q1);
      lp[mi] = (tq_f2){l0, l1};
      da[mi] = (tq_f2){d0, d1};
      gq[mi] = (tq_f2){q0, q1};
    }
  } else {
    float m0[M], m1[M], l0[M], l1[M], d0[M], d1[M], q0[M], q1[M];
#pragma unroll
    for (int mi = 0; mi < M; ++mi) {
      m0[mi] = mu[mi].x;
      m1[mi] = mu[mi].y;
    }
    if (fast) {
      tq_pix_multi_offset<M, BWD, true>(D.x, m0, a.offset_samples, a.offset_logits, a.O, h, g, rg, ln_g, l0, d0, q0);
      tq_pix_multi_offset<M, BWD, true>(D.y, m1, a.offset_samples, a.offset_logits, a.O, h, g, rg, ln_g, l1, d1, q1);
    } else {
      tq_pix_multi_offset<M, BWD, false>(D.x, m0, a.offset_samples, a.offset_logits, a.O, h, g, rg, ln_g, l0, d0, q0);
      tq_pix_multi_offset<M, BWD, false>(D.y, m1, a.offset_samples, a.offset_logits, a.O, h, g, rg, ln_g, l1, d1, q1);
    }
#pragma unroll
    for (int mi = 0; mi < M; ++mi) {
      lp[mi] = (tq_f2){l0[mi], l1[mi]};
      da[mi] = (tq_f2){d0[mi], d1[mi]};
      gq[mi] = (tq_f2){q0[mi], q1[mi]};
    }
  }
  tq_f2 q[K];
#pragma unroll
  for (int k = 0; k < K; ++k) q[k] = tq2(0.0f);
#pragma unroll
  for (int mi = 0; mi < M; ++mi) {
    A.ll[mi] += lp[mi].x + lp[mi].y;
    if (BWD) {
      const tq_f2 cw = W[mi] * da[mi];
      const tq_f2 gw = W[mi] * gq[mi];
      A.acc_b += cw.x + cw.y;
      A.acc_g += gw.x + gw.y;
#pragma unroll
      for (int k = 0; k < K; ++k)
        if ((mi >> k) & 1) q[k] += cw;
    }
  }
  if (BWD) {
    const tq_f2 r2 = fic * fic + fj * fj;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const tq_f2 aq = q[k] * spot[k];
      const tq_f2 ax = aq * fic, ar = aq * r2;
      A.S0[k] += aq.x + aq.y;
      A.Sx[k] += ax.x + ax.y;
      A.Sy[k] += (aq.x + aq.y) * fj;
      A.Sr[k] += ar.x + ar.y;
    }
  }
}

template <int K, bool BWD, bool WFAC>
__device__ __forceinline__ void tq_il2m_pixel_loop(TqPixAcc<K>& A, const tq_ksmogn_args& a, const TqOffsetInfo& h,
                                                   const float2* __restrict__ s_tab, const bool fast,
                                                   const float4* __restrict__ src, int P, float b, const float* amph,
                                                   const float* nl2, const float* cx, const float* cy, float g,
                                                   float rg, float ln_g, const float* W) {
  const int npix4 = (P * P) >> 2;
  const float c0 = 0.5f * (float)(P - 1);  // spot-weighted moments are taken about the tile centre
  int ic = 0, jr = 0;  // wave-uniform coordinates of the next pixel pair (P even: a pair never straddles rows)
  float agy[K];
#pragma unroll
  for (int k = 0; k < K; ++k) agy[k] = amph[k] * __builtin_amdgcn_exp2f(cy[k] * cy[k] * nl2[k]);
  float4 cur = src[0];
  for (int q = 0; q < npix4; ++q) {
    const float4 d4 = cur;
    if (q + 1 < npix4) cur = src[(int64_t)(q + 1) * 64];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const tq_f2 D = half ? (tq_f2){d4.z, d4.w} : (tq_f2){d4.x, d4.y};
      const tq_f2 fic = (tq_f2){(float)ic, (float)(ic + 1)};
      const float fj = (float)jr;
      tq_f2 spot[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const tq_f2 dx = fic - cx[k];
        spot[k] = agy[k] * tq2_exp2(dx * dx * nl2[k]);
      }
      tq_pair_multi_offset<K, BWD, WFAC>(A, a, h, s_tab, fast, D, b, spot, W, fic - c0, fj - c0, g, rg, ln_g);
      ic += 2;
      if (ic == P) {
        ic = 0;
        ++jr;
        const float fn = (float)jr;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float dy = fn - cy[k];
          agy[k] = amph[k] * __builtin_amdgcn_exp2f(dy * dy * nl2[k]);
        }
      }
    }
  }
}

template <int K, bool BWD>
__device__ __forceinline__ void tq_il2m_body(const tq_ksmogn_args& a, const int64_t B, float2* s_tab) {
  constexpr int M = 1 << K;
  const int64_t i_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i_raw < B;
  const int64_t i = live ? i_raw : (B - 1);
  const int P = a.P, npix = P * P;
  const int npix4 = npix >> 2;
  // idle lanes of the last workgroup read the LAST tile too: the interleaved buffer ends with its 64-tile block
  const float4* src = reinterpret_cast<const float4*>(a.images_il) + ((i >> 6) * npix4) * 64 + (i & 63);

  const float g = a.gain[0];
  const float rg = TQ_FRCP(g);
  const float ln_g = TQ_FLOG(g);
  TqOffsetInfo h;
  tq_offset_info(a.offset_samples, a.offset_logits, a.O, &h);
  const float beta2 = rg * TQ_LOG2E;
  // |db_o| <= max(log2(w_max / w_min), beta2 (delta_max - delta_min)): the factored form needs it well inside +-126
  const bool wfac = fmaxf(h.lw2max - h.lw2min, beta2 * (h.dmax - h.dmin)) < 40.0f;
  for (int o = threadIdx.x; o < a.O; o += 256) {
    const float dd = a.offset_samples[o] - h.dmin;
    const float db = (a.offset_logits[o] * TQ_LOG2E - h.lw2max) + beta2 * dd;
    const float w = __builtin_amdgcn_exp2f(db);
    s_tab[o] = wfac ? make_float2(w * dd, w) : make_float2(dd, db);
  }
  __syncthreads();

  const float tx = a.xy[2 * i], ty = a.xy[2 * i + 1];
  const float b = a.background[i];
  float hk[K], wk[K], amph[K], nl2[K], cx[K], cy[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    hk[k] = a.height[k * B + i];
    wk[k] = a.width[k * B + i];
    cx[k] = a.x[k * B + i] + tx;
    cy[k] = a.y[k * B + i] + ty;
    const float inv2v = 0.5f * TQ_FRCP(wk[k] * wk[k]);
    amph[k] = hk[k] * inv2v * (1.0f / TQ_PI);
    nl2[k] = -inv2v * 1.44269504088896340736f;
  }
  float W[M];
#pragma unroll
  for (int mi = 0; mi < M; ++mi) W[mi] = 0.0f;
  if (BWD) tq_load_weights<K>(a, B, i, i, (int)((uint32_t)i / (uint32_t)(a.F * a.C)), W);

  TqPixAcc<K> A;
  tq_acc_zero<K>(A);
  const bool fast = __all(b * rg >= TQ_FAST_ALPHA);
  if (wfac) tq_il2m_pixel_loop<K, BWD, true>(A, a, h, s_tab, fast, src, P, b, amph, nl2, cx, cy, g, rg, ln_g, W);
  else tq_il2m_pixel_loop<K, BWD, false>(A, a, h, s_tab, fast, src, P, b, amph, nl2, cx, cy, g, rg, ln_g, W);
  if (live) {
    float cxs[K], cys[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      cxs[k] = cx[k] - 0.5f * (float)(P - 1);
      cys[k] = cy[k] - 0.5f * (float)(P - 1);
    }
    tq_pixel_store<K, false, BWD>(a, B, i, A, W, b, rg, hk, wk, cxs, cys, (float)npix, 0.0f, false);
  }
}
// K <= 2 with the cap of two waves per SIMD: at K = 2 with the backward the allocator lands on 270 registers without it (since
// the reference constant moved into the exponent), a second wave no longer fits and the launch takes 18 % longer; 13 spilled
// registers with it.  K >= 3 would spill hundreds under the cap and keeps one wave.
template <int K, bool BWD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void tq_ksmogn_il2m_kernel(const tq_ksmogn_args a, const int64_t B) {
  __shared__ float2 s_tab[TQ_MO_MAX_O];
  tq_il2m_body<K, BWD>(a, B, s_tab);
}
template <int K, bool BWD>
__global__ __launch_bounds__(256) void tq_ksmogn_il2m_wide_kernel(const tq_ksmogn_args a, const int64_t B) {
  __shared__ float2 s_tab[TQ_MO_MAX_O];
  tq_il2m_body<K, BWD>(a, B, s_tab);
}

// (U, npix) row-major tiles -> the interleaved layout above; out holds ceil(U/64) * npix4 * 256 floats
__global__ __launch_bounds__(256) void tq_interleave_kernel(const float* __restrict__ images, float* __restrict__ out,
                                                            const int64_t U, const int npix, const int64_t total4) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one float4 of the output
  if (t >= total4) return;
  const int npix4 = (npix + 3) >> 2;
  const int lane = (int)(t & 63);
  const int64_t rowq = t >> 6;
  const int q = (int)(rowq % npix4);
  const int64_t u = (rowq / npix4) * 64 + lane;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (u < U) {
    const float* tile = images + u * npix;
    const int p = 4 * q;
    v.x = p < npix ? tile[p] : 0.f;
    v.y = p + 1 < npix ? tile[p + 1] : 0.f;
    v.z = p + 2 < npix ? tile[p + 2] : 0.f;
    v.w = p + 3 < npix ? tile[p + 3] : 0.f;
  }
  reinterpret_cast<float4*>(out)[t] = v;
}

// Per-unit data statistics of the single-offset path: [0][u] = sum (D - delta), [1][u] = sum ln(D - delta),
// [2][u] = number of pixels with D <= delta.  One wave per unit, sums in double.
__global__ __launch_bounds__(256) void tq_image_stats_kernel(const float* __restrict__ images, const float* offset,
                                                             float* __restrict__ out, const int64_t U, const int npix) {
  const int64_t u = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (u >= U) return;
  const float off = offset[0];
  double sv = 0.0, slv = 0.0;
  float nbad = 0.0f;
  for (int p = lane; p < npix; p += 64) {
    const float v = images[u * npix + p] - off;
    if (v > 0.0f) {
      sv += v;
      slv += log((double)v);
    } else {
      nbad += 1.0f;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sv += __shfl_down(sv, o, 64);
    slv += __shfl_down(slv, o, 64);
    nbad += __shfl_down(nbad, o, 64);
  }
  if (lane == 0) {
    out[u] = (float)sv;
    out[U + u] = (float)slv;
    out[2 * U + u] = nbad;
  }
}

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
extern "C" const char* tq_last_error(void) { return g_err; }
extern "C" int tq_version(void) { return 101; }
void tq_set_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }

static int launch_status(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[200];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    tq_set_error(buf);
    return TQ_ERR_LAUNCH;
  }
  return TQ_OK;
}

template <int K, bool ONE>
static int launch_kb(const tq_ksmogn_args& a, int64_t B, hipStream_t st) {
  const bool bwd = a.g_background != nullptr;
  // contiguous batch + interleaved copy available + enough units to fill the chip with 64-unit waves
  if (a.images_il && !a.ndx && !a.fdx && a.nb == a.nb_full && a.fb == a.F && B >= a.il_min_units) {
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
    if (ONE && (a.P == 14 || a.P == 20)) {
      const dim3 grid1((unsigned)((B + 63) / 64)), block1(64);
      // a.pixel_mode: 1 = persistent waves (two per SIMD walk over the tiles), 0 = one wave per tile.  Which is faster
      // depends on the box (wave-launch and memory latencies; measured 114-151 us vs 120-131 us at 400 000 units across
      // the pool): the host times both once and says (CosmosEngine); TAPQIR_AMD_PERSIST=0/1 overrides for A/B runs
      static const int forced = [] {
        const char* e = getenv("TAPQIR_AMD_PERSIST");
        return e ? atoi(e) : -1;
      }();
      const int persist = forced >= 0 ? forced : a.pixel_mode;
      if (bwd && !a.gout && persist > 0 && K <= 2) {  // (K = 3 spills in this form)
        int ntiles = (int)((B + 63) / 64);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        static const int waves_forced = [] {  // tests: few resident waves = many tiles per wave
          const char* e = getenv("TAPQIR_AMD_PERSIST_WAVES");
          return e ? atoi(e) : 0;
        }();
        const int waves = waves_forced > 0 ? waves_forced : cus * 4 * TQ_P_WAVES;
        const int64_t Bn = B;
        ntiles = (int)((Bn + 63) / 64);
        const dim3 gridp((unsigned)(ntiles < waves ? ntiles : waves));
        if (a.P == 14) hipLaunchKernelGGL((tq_ksmogn_il2p_kernel<K, 14, true>), gridp, block1, 0, st, a, B, Bn, ntiles);
        else hipLaunchKernelGGL((tq_ksmogn_il2p_kernel<K, 20, true>), gridp, block1, 0, st, a, B, Bn, ntiles);
        return launch_status("tq_ksmogn_il2p_kernel");
      }
      if (a.P == 14) {
        if (bwd) hipLaunchKernelGGL((tq_ksmogn_il2_kernel<K, 14, true>), grid1, block1, 0, st, a, B);
        else hipLaunchKernelGGL((tq_ksmogn_il2_kernel<K, 14, false>), grid1, block1, 0, st, a, B);
      } else {
        if (bwd) hipLaunchKernelGGL((tq_ksmogn_il2_kernel<K, 20, true>), grid1, block1, 0, st, a, B);
        else hipLaunchKernelGGL((tq_ksmogn_il2_kernel<K, 20, false>), grid1, block1, 0, st, a, B);
      }
      return launch_status("tq_ksmogn_il2_kernel");
    }
    if (!ONE && (a.P % 2) == 0 && a.O <= TQ_MO_MAX_O) {
      if constexpr (K <= 2) {
        if (bwd) hipLaunchKernelGGL((tq_ksmogn_il2m_kernel<K, true>), grid, block, 0, st, a, B);
        else hipLaunchKernelGGL((tq_ksmogn_il2m_kernel<K, false>), grid, block, 0, st, a, B);
      } else {
        if (bwd) hipLaunchKernelGGL((tq_ksmogn_il2m_wide_kernel<K, true>), grid, block, 0, st, a, B);
        else hipLaunchKernelGGL((tq_ksmogn_il2m_wide_kernel<K, false>), grid, block, 0, st, a, B);
      }
      return launch_status("tq_ksmogn_il2m_kernel");
    }
    if (bwd) hipLaunchKernelGGL((tq_ksmogn_il_kernel<K, ONE, true>), grid, block, 0, st, a, B);
    else hipLaunchKernelGGL((tq_ksmogn_il_kernel<K, ONE, false>), grid, block, 0, st, a, B);
    return launch_status("tq_ksmogn_il_kernel");
  }
  if (!a.images) {
    tq_set_error("tq_ksmogn_log_prob: images is NULL and the interleaved kernel does not apply to this batch");
    return TQ_ERR_ARG;
  }
  if (!ONE && a.O >= 8) {
    // offset histograms: a wave per unit (the offset loop makes a unit ~O times the work of the single-offset form, and a
    // gathered batch is small: at 16 lanes per unit a 10 x 512 minibatch fills 1.25 waves per SIMD)
    static const bool wide = [] {
      const char* e = getenv("TAPQIR_AMD_WIDE_HIST");
      return !(e && e[0] == '0');
    }();
    if (wide) {
      constexpr int UNITS = TQ_BLOCK / 64;
      const dim3 gridw((unsigned)((B + UNITS - 1) / UNITS)), blockw(TQ_BLOCK);
      const size_t ldsw = sizeof(float) * tq_tile16_lds_floats(a.P, K, a.O, UNITS);
      if (bwd) hipLaunchKernelGGL((tq_ksmogn_kernel<K, false, true, 64>), gridw, blockw, ldsw, st, a, B);
      else hipLaunchKernelGGL((tq_ksmogn_kernel<K, false, false, 64>), gridw, blockw, ldsw, st, a, B);
      return launch_status("tq_ksmogn_kernel (64 lanes per unit)");
    }
  }
  const dim3 grid((unsigned)((B + TQ_UNITS_PER_BLOCK - 1) / TQ_UNITS_PER_BLOCK)), block(TQ_BLOCK);
  const size_t lds = sizeof(float) * tq_tile16_lds_floats(a.P, K, a.O);
  if (bwd) hipLaunchKernelGGL((tq_ksmogn_kernel<K, ONE, true>), grid, block, lds, st, a, B);
  else hipLaunchKernelGGL((tq_ksmogn_kernel<K, ONE, false>), grid, block, lds, st, a, B);
  return launch_status("tq_ksmogn_kernel");
}

template <int K>
static int launch_k(const tq_ksmogn_args& a, int64_t B, hipStream_t st) {
  // the single-offset formulation needs the per-unit data statistics; without them the general
  // (online log-sum-exp) path handles O == 1 as well
  if (a.O == 1 && a.pixstats) return launch_kb<K, true>(a, B, st);
  return launch_kb<K, false>(a, B, st);
}

extern "C" int tq_ksmogn_log_prob(const tq_ksmogn_args* a, void* stream) {
  if (!a || (!a->images && !a->images_il) || !a->xy || !a->background || !a->height || !a->width || !a->x || !a->y ||
      !a->gain || !a->offset_samples || !a->offset_logits || !a->ll) {
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

extern "C" int64_t tq_interleaved_floats_n(int64_t U, int32_t npix) {
  const int64_t npix4 = ((int64_t)npix + 3) / 4;
  return ((U + 63) / 64) * npix4 * 256;
}
extern "C" int64_t tq_interleaved_floats(int64_t U, int32_t P) { return tq_interleaved_floats_n(U, P * P); }

extern "C" int tq_images_interleave_n(const float* images, float* images_il, int64_t U, int32_t npix, void* stream) {
  if (!images || !images_il || U < 1 || npix < 4) {
    tq_set_error("tq_images_interleave_n: bad argument");
    return TQ_ERR_ARG;
  }
  const int64_t total4 = tq_interleaved_floats_n(U, npix) / 4;
  hipLaunchKernelGGL(tq_interleave_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     images, images_il, U, npix, total4);
  return launch_status("tq_interleave_kernel");
}

extern "C" int tq_images_interleave(const float* images, float* images_il, int64_t U, int32_t P, void* stream) {
  if (!images || !images_il || U < 1 || P < 2 || P > TQ_MAX_P) {
    tq_set_error("tq_images_interleave: bad argument");
    return TQ_ERR_ARG;
  }
  const int64_t total4 = tq_interleaved_floats(U, P) / 4;
  hipLaunchKernelGGL(tq_interleave_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     images, images_il, U, P * P, total4);
  return launch_status("tq_interleave_kernel");
}

extern "C" int tq_image_stats(const float* images, const float* offset, float* pixstats, int64_t U, int32_t P,
                              void* stream) {
  if (!images || !offset || !pixstats || U < 1 || P < 2 || P > TQ_MAX_P) {
    tq_set_error("tq_image_stats: bad argument");
    return TQ_ERR_ARG;
  }
  const int64_t threads = U * 64;
  hipLaunchKernelGGL(tq_image_stats_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     images, offset, pixstats, U, P * P);
  return launch_status("tq_image_stats_kernel");
}
