// Packed lane-per-unit pixel routine (single offset, P in {14, 20}), shared by tq_ksmogn_il2_kernel (tq_ksmogn.hip)
// and the fused pixel + per-unit kernel of full-batch steps (tq_cosmos.hip).  See tq_ksmogn.hip for the description.
#pragma once
#include <type_traits>

#include "tq_ksmogn_dev.h"

// =============================================================================================
// Lane-per-unit variant for CONTIGUOUS batches (ndx == fdx == NULL: minibatch unit i = dataset unit i).
//
// Image layout (built once per dataset by tq_images_interleave): units are grouped in blocks of 64,
// and within a block the pixels are interleaved so that the j-th group of 4 pixels of the 64 units is
// one contiguous 1 KiB row:
//     images_il[((u / 64) * npix4 + q) * 64 + (u % 64)] = float4{ pixels 4q .. 4q+3 of unit u },  npix4 = ceil(P*P / 4)
// Per pixel each lane evaluates exp for the x-factor of each spot (the y-factor is per row);
// everything else is the same per-pixel code as above.
// =============================================================================================
template <int K, bool ONE_OFFSET, bool BWD, bool FAST>
__device__ __forceinline__ void tq_il_pixel_loop(TqPixAcc<K>& A, const tq_ksmogn_args& a, const float4* __restrict__ src,
                                                 int P, int npix, float b, const float* amph, const float* nl2,
                                                 const float* cx, const float* cy, float g, float rg, float ln_g,
                                                 const float* W) {
  const int npix4 = (npix + 3) >> 2;
  const float off0 = a.offset_samples[0];
  TqOffsetInfo h;
  if (!ONE_OFFSET) tq_offset_info(a.offset_samples, a.offset_logits, a.O, &h);
  int ic = 0, jr = 0;  // wave-uniform pixel coordinates
  const float c0 = 0.5f * (float)(P - 1);  // spot-weighted moments are taken about the tile centre (see tq_pixel_loop)
  float fj = 0.0f, agy[K];
#pragma unroll
  for (int k = 0; k < K; ++k) agy[k] = amph[k] * __builtin_amdgcn_exp2f(cy[k] * cy[k] * nl2[k]);

  // one group of 4 pixels held in a float4
  auto group = [&](const float4& d, int q) {
    const float d4[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (4 * q + e < npix) {
        const float fic = (float)ic;
        float spot[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float dx = fic - cx[k];
          spot[k] = agy[k] * __builtin_amdgcn_exp2f(dx * dx * nl2[k]);
        }
        if (ONE_OFFSET) tq_pixel_one_offset<K, BWD, FAST>(A, d4[e] - off0, b, spot, W, fic - c0, fj - c0, g, rg, ln_g);
        else tq_pixel_multi_offset<K, BWD, FAST>(A, a, h, d4[e], ln_g, b, spot, W, fic - c0, fj - c0, g, rg);
        if (++ic == P) {  // next row: refresh the y-factors
          ic = 0;
          ++jr;
          fj = (float)jr;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const float dy = fj - cy[k];
            agy[k] = amph[k] * __builtin_amdgcn_exp2f(dy * dy * nl2[k]);
          }
        }
      }
    }
  };
  // three named registers in rotation: each is reloaded right after it has been consumed, so two
  // 1 KiB rows per wave are always in flight and no register move (hence no vmcnt(0)) is needed
  float4 r0 = src[0];
  float4 r1 = npix4 > 1 ? src[64] : r0;
  float4 r2 = npix4 > 2 ? src[128] : r0;
  for (int q = 0; q < npix4; q += 3) {
    group(r0, q);
    if (q + 3 < npix4) r0 = src[(int64_t)(q + 3) * 64];
    if (q + 1 < npix4) {
      group(r1, q + 1);
      if (q + 4 < npix4) r1 = src[(int64_t)(q + 4) * 64];
    }
    if (q + 2 < npix4) {
      group(r2, q + 2);
      if (q + 5 < npix4) r2 = src[(int64_t)(q + 5) * 64];
    }
  }
}

// Running sums of the packed loop.  COLACC: the x-moments are kept as per-COLUMN sums (one fma per pixel pair
// and spot instead of three) and folded with the column coordinates once per unit; used when the K*P/2
// extra float2 registers fit.
template <int K, int P, bool COLACC>
struct TqPixAcc2 {
  tq_f2 T[1 << K];                       // sum over pixels of (alpha - 1/2) ln(v/mu) - S(alpha)
  tq_f2 acc_b;
  tq_f2 S0r[K];                          // sum of q*spot over the current row
  tq_f2 S0[K], Sy[K], Syy[K];            // per-row folds: sum S0r * {1, j, j^2}
  tq_f2 Sx[K], Sxx[K];                   // !COLACC: sum q*spot*{i, i^2}
  tq_f2 col[COLACC ? K : 1][COLACC ? P / 2 : 1];  // COLACC: sum over rows of q*spot per column pair
};

// per-unit constants of the fast (alpha >= TQ_FAST_ALPHA, two Binet terms) single-offset pixel:
//   S(alpha)    = r (g/12 - g^3/360 r^2),                      r = 1/mu, alpha = mu/g
//   da / ln2    = log2(v/mu) + r (g/2 + r (g^2/12 - g^4/120 r^2)) / ln2
struct TqFastConst {
  float ca, cb, s1, s3, d1, d2, d4;  // ca = ln2 / g, cb = ln2 / 2: (alpha - 1/2) ln(v/mu) = (ca mu - cb) log2(v/mu)
};

// one pair of horizontally adjacent pixels (column pair ip of the current row), single offset, fast alpha
template <int K, int P, bool BWD, bool COLACC>
__device__ __forceinline__ void tq_pixel_pair(TqPixAcc2<K, P, COLACC>& A, tq_f2 v, float b, const tq_f2* spot,
                                              const float* W, const int ip, const bool first_in_row,
                                              const float c_ca, const float c_cb, const float c_s1, const float c_s3,
                                              const float c_d1, const float c_d2, const float c_d4) {
  // (the constants of TqFastConst as scalars: as a struct they end up in scratch in some instantiations)
  constexpr int M = 1 << K;
  tq_f2 da[M];
  tq_f2 mus[M];
  mus[0] = tq2(b);
#pragma unroll
  for (int mi = 1; mi < M; ++mi) {
    // mu(m) = mu(m without its highest spot) + that spot: one packed add per combination
    const int hi = 31 - __builtin_clz(mi);
    mus[mi] = mus[mi & ~(1 << hi)] + spot[hi];
    const tq_f2 r = tq2_rcp(mus[mi]);
    const tq_f2 l2 = tq2_log2(v * r);
    const tq_f2 u = r * r;
    A.T[mi] += (mus[mi] * c_ca - c_cb) * l2;
    A.T[mi] -= r * (u * c_s3 + c_s1);
    if (BWD) da[mi] = r * (r * (u * c_d4 + c_d2) + c_d1) + l2;  // = da / ln2
  }
  if (BWD) {
    // q_k = sum_{m containing k} W_m da_m,  acc_b += sum_m W_m da_m
    tq_f2 q[K];
    if (K == 2) {
      const tq_f2 t3 = W[3] * da[3];
      q[0] = W[1] * da[1] + t3;
      q[1] = W[2] * da[2] + t3;
      A.acc_b += q[0];
      A.acc_b += W[2] * da[2];
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) q[k] = tq2(0.0f);
#pragma unroll
      for (int mi = 1; mi < M; ++mi) {
        const tq_f2 cw = W[mi] * da[mi];
        A.acc_b += cw;
#pragma unroll
        for (int k = 0; k < K; ++k)
          if ((mi >> k) & 1) q[k] += cw;
      }
    }
    const tq_f2 fic = (tq_f2){(float)(2 * ip) - 0.5f * (float)(P - 1), (float)(2 * ip + 1) - 0.5f * (float)(P - 1)};
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (COLACC) {
        A.col[k][ip] += q[k] * spot[k];
        A.S0r[k] = first_in_row ? q[k] * spot[k] : A.S0r[k] + q[k] * spot[k];
      } else {
        const tq_f2 aq = q[k] * spot[k];
        A.S0r[k] = first_in_row ? aq : A.S0r[k] + aq;
        A.Sx[k] += aq * fic;  // fic: column coordinates about the tile centre (compile-time constants)
        A.Sxx[k] += aq * (fic * fic);
      }
    }
  }
}


// One lane = one unit of tile blockIdx.x (64 units).  OUT != nullptr: the results of the unit (TQ_PIXOUT(K) floats:
// ll[2^K], g_background, g_gain, g_height[K], g_width[K], g_x[K], g_y[K]) are returned there and NOT stored.
#define TQ_PIXOUT(K) ((1 << (K)) + 2 + 4 * (K))
template <int K, int P, bool BWD>
__device__ __forceinline__ void tq_il2_lane(const tq_ksmogn_args& a, const int64_t B, float* OUT = nullptr) {
  static_assert(P % 2 == 0 && (P * P) % 4 == 0, "packed kernel needs an even tile side");
  constexpr int M = 1 << K;
  constexpr int R = ((P / 2) % 2) ? 2 : 1;  // rows per loop body so that the body starts on a float4 boundary
  constexpr int G = R * P / 4;              // float4 groups per body
  constexpr int NB = P / R;                 // bodies per tile
  constexpr int npix = P * P, npix4 = npix / 4;
  constexpr bool COLACC = BWD && (K * P <= 28);
  constexpr bool PINNED = BWD && K == 2;  // prefetches fenced in place + first body peeled (see run_body)
  const int64_t i_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;  // one wave per workgroup: no barriers, finest dispatch granularity
  const bool live = i_raw < B;
  const int64_t i = live ? i_raw : (B - 1);
  // idle lanes of the last workgroup read the LAST tile too: the interleaved buffer ends with its 64-tile block
  const float4* src = reinterpret_cast<const float4*>(a.images_il) + ((i >> 6) * npix4) * 64 + (i & 63);

  const float g = a.gain[0];
  const float rg = TQ_FRCP(g);
  const float ln_g = TQ_FLOG(g);
  const float off0 = a.offset_samples[0];
  const float tx = a.xy[2 * i], ty = a.xy[2 * i + 1];
  const float b = a.background[i];
  float amph[K], nl2[K], cx[K], cy[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float hk = a.height[k * B + i], wk = a.width[k * B + i];
    cx[k] = a.x[k * B + i] + tx;
    cy[k] = a.y[k * B + i] + ty;
    const float inv2v = 0.5f * TQ_FRCP(wk * wk);
    amph[k] = hk * inv2v * (1.0f / TQ_PI);
    nl2[k] = -inv2v * 1.44269504088896340736f;
  }
  float W[M];
#pragma unroll
  for (int mi = 0; mi < M; ++mi) W[mi] = 0.0f;
  if (BWD) tq_load_weights<K>(a, B, i, i, (int)((uint32_t)i / (uint32_t)(a.F * a.C)), W);

  TqPixAcc<K> S;
  tq_acc_zero<K>(S);
  const bool fastpath = __all(b * rg >= TQ_FAST_ALPHA);
  if (fastpath) {
    // the x-factor of a spot does not depend on the row: P values per spot, kept in registers
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

    float4 ring[G];
#pragma unroll
    for (int j = 0; j < G; ++j) ring[j] = src[j * 64];
    // one loop body = R rows; MORE: the groups of the next body are fetched as those of this one retire
    // (the last body is peeled so that the prefetches are unconditional and stay where they are written)
    auto run_body = [&](const int body, auto more_tag) {
      constexpr bool MORE = decltype(more_tag)::value;
      const float4* nxt = src + (int64_t)(body + 1) * G * 64;
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
          const int gi = pair >> 1;            // float4 group within the body
          const float4 d4 = ring[gi];
          // v = D - delta is taken BEFORE the group is re-loaded, and (backward kernel) the load is fenced in place:
          // the old registers are dead at the load, which writes the next body's group straight into them.  A refill
          // hoisted above the last use of its registers costs a copy at the end of the body, and that copy waits for
          // every load of the body (vmcnt(0) once per body, ~1 us each).  Together with the peeled first body (the
          // loop is then entered with the loads pending in the same order as at its back edge, so the waits the
          // compiler derives are exact: vmcnt(6) before each group) this is worth 1-8 % of the K = 2 backward launch,
          // depending on the box; the forward kernels do not profit and the K = 1 backward kernel would lose its
          // fourth wave per SIMD (130 registers), so they keep the plain form (PINNED).
          const tq_f2 v = ((pair & 1) ? (tq_f2){d4.z, d4.w} : (tq_f2){d4.x, d4.y}) - off0;
          if (MORE && (pair & 1)) {
            if (PINNED) __builtin_amdgcn_sched_barrier(0);
            ring[gi] = nxt[gi * 64];
            if (PINNED) __builtin_amdgcn_sched_barrier(0);
          }
          tq_f2 spot[K];
#pragma unroll
          for (int k = 0; k < K; ++k) spot[k] = agy[k] * ex[k][ip];
          tq_pixel_pair<K, P, BWD, COLACC>(A, v, b, spot, W, ip, ip == 0, c.ca, c.cb, c.s1, c.s3, c.d1, c.d2, c.d4);
        }
        if (BWD) {
          // y-moments about the spot's own centre (the row offset dy is at hand): no cancellation later
#pragma unroll
          for (int k = 0; k < K; ++k) {
            if (!COLACC) A.S0[k] += A.S0r[k];
            A.Sy[k] += A.S0r[k] * dyk[k];
            A.Syy[k] += A.S0r[k] * (dyk[k] * dyk[k]);
          }
        }
      }
    };
    if (PINNED) {  // first body peeled: see above
      run_body(0, std::true_type{});
#pragma unroll 1
      for (int body = 1; body < NB - 1; ++body) run_body(body, std::true_type{});
    } else {
#pragma unroll 1
      for (int body = 0; body < NB - 1; ++body) run_body(body, std::true_type{});
    }
    run_body(NB - 1, std::false_type{});

    // fold the two pixel slots, then the common single-offset assembly / store
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
          // the offsets are RE-computed from an opaque copy of the centre: shared with the prologue's (i - cx) and
          // (i - cx)^2 they would stay live across the pixel loop, which at 256 registers means 64 B of scratch
          // stores + loads per unit (PMC: +25 MB each way per launch)
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
    // some unit of this wave has a small alpha = background / gain: general (scalar, exact Binet) loop
    tq_il_pixel_loop<K, true, BWD, false>(S, a, src, P, npix, b, amph, nl2, cx, cy, g, rg, ln_g, W);
  }
  const float S_v = a.pixstats[i];
  const float S_lv = a.pixstats[a.stats_stride + i];
  const bool bad = a.pixstats[2 * a.stats_stride + i] > 0.0f;
  tq_pixel_assemble_one_offset<K, BWD>(a, S, W, b, g, rg, ln_g, (float)npix, S_v, S_lv);
  if (live) {
    float hk[K], wk[K], cxs[K], cys[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      hk[k] = a.height[k * B + i];
      wk[k] = a.width[k * B + i];
      // reference points of the moments: the packed loop centres y on the spot and x on the spot (column sums) or on
      // the tile centre; the scalar fallback loop uses the tile centre for both
      cxs[k] = fastpath ? (COLACC ? 0.0f : cx[k] - 0.5f * (float)(P - 1)) : cx[k] - 0.5f * (float)(P - 1);
      cys[k] = fastpath ? 0.0f : cy[k] - 0.5f * (float)(P - 1);
    }
    tq_pixel_store<K, true, BWD>(a, B, i, S, W, b, rg, hk, wk, cxs, cys, (float)npix, S_v, bad, OUT);
  }
}
