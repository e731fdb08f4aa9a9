// tq_cosmos.hip -- kernels of one cosmos SVI step around the pixel kernel of tq_ksmogn.hip
// (guide sampling, per-unit ELBO terms + gradients, per-AOI terms, cross-unit sums, global
// sites, dense Adam).  Replaces what pyro's SVI/TraceEnum_ELBO/optim.Adam execute for
// tapqir/models/model.py:212 -- see include/tapqir_hip.h.
//
// Launch shapes: everything except the pixel kernel is one lane per work item, SoA so that
// consecutive lanes touch consecutive addresses (the flat parameter buffer is [row][unit]).
// Cross-unit sums are deterministic: wave64 __shfl_down -> LDS -> one row per workgroup ->
// a single-workgroup fp64 finish; no float atomics.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "tq_bodies.h"
#include "tq_dpp.h"

void tq_set_error(const char* msg);

#define TQ_UNIT_BLOCK 256
#define TQ_MAX_NGSUM (3 + 3 * TQ_MAXQ)  // >= 3 + 3*2 + 2*2 of the crosstalk model

// sum over the wave (every lane active), in every lane: DPP adds inside the four rows of 16 lanes, then the four row sums
// read as scalars -- no LDS crossbar (six ds_bpermute per sum in the shuffle form; the fused pixel + per-unit kernel ends
// every wave with 22 such sums)
__device__ __forceinline__ float tq_wave_sum(float v) {
  v = tq_group_sum16(v);
  const int b = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}

// ---- sampling ------------------------------------------------------------------------------------------
// one wave per global site (4 independent instruction streams instead of one serial lane)
__global__ __launch_bounds__(64) void tq_sample_globals_kernel(const tq_cosmos_args a) {
  if (threadIdx.x == 0) tq_body_sample_globals(a, blockIdx.x);
}

// ---- regime compaction of the AffineBeta implicit gradients -------------------------------------------------------
// torch's _dirichlet_grad is piecewise (two series regimes, a saddle-point expansion, a rational fit) and a wave executes
// every regime one of its lanes needs.  At the reference's initial parameters all draws sit in the saddle-point regime;
// in a converged fit the guide concentrations of absent spots have shrunk (size 5..18) and the lanes of EVERY wave are
// spread over all of them (scripts/regime_mix.py: per lane 0.50 pair / 0.40 x-small series / 0.10 (1-x)-small series /
// 0.40 rational, per wave 1.0 each), which made the sampling launch the largest of the step (157 us against 75).
// Here the workgroup (256 draws of one site kind) first classifies its draws, writes one task per needed evaluation into
// a queue in LDS ordered by regime, and evaluates the queue with consecutive lanes on consecutive tasks: a wave then runs
// one regime (two at a boundary), and each regime runs on as many waves as its tasks fill.  Same routines on the same
// arguments as tq_affine_beta_site_terms: bit-identical results.  Workgroups whose draws are all in the common
// (saddle-point pair) class skip the queue.
#define TQ_BC_NT 256
// Task queues in LDS.  A draw needs at most two evaluations, so the five classes fit three regions filled from both ends
// (no class needs another one's count before it can write): R1 = {x-small series up, rational down}, R2 = {pair up,
// (1-x)-small series down}, R3 = {saddle point of one direction}.  One 16-byte record per task.
struct TqBetaCompactLds {
  float4 r1[2 * TQ_BC_NT], r2[2 * TQ_BC_NT], r3[TQ_BC_NT];  // {draw, its alpha, size, bits((lane << 2) | direction code)}
  float c0[TQ_BC_NT];                                       // class 3: the other direction's alpha (rounding fallback)
  float res[2 * TQ_BC_NT];
  int cnt[8];                                               // tasks per class
};

__device__ __forceinline__ int tq_mbcnt(uint64_t m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// slot of task number k of class c in its region
__device__ __forceinline__ float4* tq_bc_slot(TqBetaCompactLds& L, int c, int k) {
  switch (c) {
    case 0: return &L.r2[k];
    case 1: return &L.r1[k];
    case 2: return &L.r2[2 * TQ_BC_NT - 1 - k];
    case 3: return &L.r3[k];
    default: return &L.r1[2 * TQ_BC_NT - 1 - k];
  }
}

__device__ __forceinline__ void tq_site_beta_compact(const tq_cosmos_args& a, const int site, const int64_t i, const bool live,
                                                     TqBetaCompactLds& L) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid < 8) L.cnt[tid] = 0;
  __syncthreads();  // (at the top of the kernel: every wave arrives at once)
  TqSiteDraw d;
  float t = 0.0f, c1 = 0.0f, c0 = 0.0f, size = 0.0f;
  int r0 = -1, r1 = -1;
  bool pair = false, clamped = true;
  if (live) {
    d = tq_site_draw(a, site, i);
    const float sc = d.hi - d.lo, rsc = TQ_FRCP(sc);  // (the expressions of tq_affine_beta_site_terms)
    t = (d.val - d.lo) * rsc;
    size = d.p1;
    c1 = size * (d.p0 - d.lo) * rsc;
    c0 = size * (d.hi - d.p0) * rsc;
    clamped = (d.val <= d.lo + a.eps * sc) || (d.val >= d.hi - a.eps * sc);
#ifdef TQ_DIAG_NO_BETAGRAD  // (diagnostic builds, scripts/gpu_site_diag.sh: no gradient is evaluated)
    clamped = true;
#endif
    if (!clamped) {
      pair = tq_beta_grad_pair_applies((double)t, (double)c1, (double)size - (double)c1);
      if (!pair) {
        const double total = size;
        r0 = tq_dirichlet_grad_regime((double)t, (double)c1, total - (double)c1, total);
        r1 = tq_dirichlet_grad_regime((double)(1.0f - t), (double)c0, total - (double)c0, total);
      }
    }
  }
  // A wave whose draws are all in the common class (saddle-point pair: every wave at the reference's initial parameters)
  // is already uniform: it evaluates in place and only joins the barriers (and the evaluation of other waves' tasks).
  float dd[2] = {0.0f, 0.0f};
  const bool wave_mixed = __ballot(r0 >= 0) != 0;
  if (!wave_mixed) {
    if (pair) {
      double ga = t, gb = c1;
#ifndef TQ_DIAG_NO_RP
      tq_beta_grad_pair_mid((double)t, (double)c1, (double)size - (double)c1, &ga, &gb);
#endif
      dd[0] = (float)ga;
      dd[1] = (float)gb;
    }
  } else {
    // tasks per class: 0 pair (both directions of a draw), 1 x-small series, 2 (1-x)-small series, 3 saddle point of one
    // direction (the pair routine with the boundary test off), 4 rational
    int k[5], pos[5];
    k[0] = pair ? 1 : 0;
    k[1] = (r0 == 0) + (r1 == 0);
    k[2] = (r0 == 1) + (r1 == 1);
    k[3] = (r0 == 2 || r1 == 2) ? 1 : 0;
    k[4] = (r0 == 3) + (r1 == 3);
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const uint64_t m1 = __ballot(k[c] >= 1), m2 = __ballot(k[c] == 2);
      const int n = __popcll(m1) + __popcll(m2);
      int base = 0;
      if (lane == 0 && n) base = atomicAdd(&L.cnt[c], n);  // (LDS; the order of the waves does not matter: a task's result
      pos[c] = __shfl(base, 0, 64) + tq_mbcnt(m1) + tq_mbcnt(m2);  //  does not depend on its place in the queue)
    }
    auto put = [&](int c, int kk, float x, float al, int code) {
      *tq_bc_slot(L, c, kk) = make_float4(x, al, size, __int_as_float((tid << 2) | code));
    };
    if (k[0]) put(0, pos[0], t, c1, 0);
    if (k[3]) {
      put(3, pos[3], t, c1, (r0 == 2 ? 1 : 0) | (r1 == 2 ? 2 : 0));
      L.c0[tid] = c0;
    }
    {
      const float xf[2] = {t, 1.0f - t}, af[2] = {c1, c0};
      const int rr[2] = {r0, r1};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = rr[j];
        if (r == 0) put(1, pos[1]++, xf[j], af[j], j);
        if (r == 1) put(2, pos[2]++, xf[j], af[j], j);
        if (r == 3) put(4, pos[4]++, xf[j], af[j], j);
      }
    }
    L.res[2 * tid] = 0.0f;
    L.res[2 * tid + 1] = 0.0f;
  }
  __syncthreads();
  int n_c[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) n_c[c] = L.cnt[c];
  if (n_c[0] + n_c[1] + n_c[2] + n_c[3] + n_c[4] != 0) {
    // One loop per class, so that each regime's code and registers stand alone; the classes start on successive waves
    // (class c on the wave after the last one of class c-1), which spreads the ~1.4 evaluations per draw of a converged fit
    // evenly over the four waves.
    int wave0 = 0;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int n = n_c[c];
      const int slot = (((wave - wave0) & (TQ_BC_NT / 64 - 1)) << 6) | lane;
      for (int qc = slot; qc < n; qc += TQ_BC_NT) {
        const float4 rec = *tq_bc_slot(L, c, qc);
        const float x = rec.x, al = rec.y, sz = rec.z;
        const int dst = __float_as_int(rec.w) >> 2, code = __float_as_int(rec.w) & 3;
        const double total = sz;
        if (c == 0) {
          double ga = x, gb = al;
#ifndef TQ_DIAG_NO_RP  // (diagnostic builds, scripts/gpu_site_diag.sh: one regime's routine compiled out)
          tq_beta_grad_pair_mid((double)x, (double)al, total - (double)al, &ga, &gb);
#endif
          L.res[2 * dst] = (float)ga;
          L.res[2 * dst + 1] = (float)gb;
        } else if (c == 1) {
#ifdef TQ_DIAG_NO_R0
          L.res[2 * dst + code] = x;
#else
          L.res[2 * dst + code] = (float)tq_beta_grad_alpha_small((double)x, (double)al, total - (double)al);
#endif
        } else if (c == 2) {
#ifdef TQ_DIAG_NO_R1
          L.res[2 * dst + code] = x;
#else
          L.res[2 * dst + code] = -tq_beta_grad_beta_small_f(1.0f - x, sz - al, al);
#endif
        } else if (c == 3) {
          double ga = 0.0, gb = 0.0;
          if (!tq_beta_grad_pair_mid<true>((double)x, (double)al, total - (double)al, &ga, &gb)) {
            // (the two directions disagree about alpha, beta > 6 within rounding: the plain evaluation, as tq_beta_grad_pair_rest)
            const float c0q = L.c0[dst];
            ga = tq_beta_grad_alpha_mid((double)x, (double)al, total - (double)al);
            gb = tq_beta_grad_alpha_mid((double)(1.0f - x), (double)c0q, total - (double)c0q);
          }
          if (code & 1) L.res[2 * dst] = (float)ga;
          if (code & 2) L.res[2 * dst + 1] = (float)gb;
        } else {
#ifdef TQ_DIAG_NO_R3
          L.res[2 * dst + code] = x;
#else
          L.res[2 * dst + code] = tq_beta_grad_rational(x, al, sz);
#endif
        }
      }
      wave0 += (n + 63) >> 6;
    }
    __syncthreads();
    if (wave_mixed) {
      dd[0] = L.res[2 * tid];
      dd[1] = L.res[2 * tid + 1];
    }
  }
  if (live) {
    float terms[TQ_NSITE_TERMS];
    tq_affine_beta_site_terms(d.val, d.p0, d.p1, d.lo, d.hi, a.eps, terms, dd);
    tq_site_store(a, site, d, terms);
  }
}

// one site of one unit per lane; workgroups are uniform in the site (grid.y), AffineBeta sites go through the compaction
__device__ __forceinline__ void tq_sample_site_wg(const tq_cosmos_args& a, const int site, const int64_t i, const int64_t B) {
#ifndef TQ_DIAG_NO_COMPACT  // (diagnostic builds: the plain per-lane evaluation everywhere)
  if (site > a.K) {
    __shared__ TqBetaCompactLds s_bc;
    tq_site_beta_compact(a, site, i, i < B, s_bc);
  } else
#endif
  if (i < B) {
    tq_body_site(a, site, i);
  }
}

// grid.y = site: the site kind (Gamma / AffineBeta, which parameter rows) is uniform per workgroup
__global__ __launch_bounds__(256) void tq_sample_locals_kernel(const tq_cosmos_args a, const int64_t B, const int site_begin) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  tq_sample_site_wg(a, site_begin + (int)blockIdx.y, i, B);
}

// ---- per-unit terms ------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(TQ_UNIT_BLOCK) void tq_unit_kernel(const tq_cosmos_args a, const int64_t B) {
  __shared__ float s_part[TQ_UNIT_BLOCK / 64][TQ_MAX_NGSUM];
  const int64_t i = (int64_t)blockIdx.x * TQ_UNIT_BLOCK + threadIdx.x;
  const int nq = tq_num_gsum(a);
  float part[TQ_MAX_NGSUM];
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) part[j] = 0.0f;
  if (i < B) tq_body_unit<K>(a, i, part);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) {
    if (j < nq) {
      const float s = tq_wave_sum(part[j]);
      if (lane == 0) s_part[wave][j] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < nq) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < TQ_UNIT_BLOCK / 64; ++w) s += s_part[w][threadIdx.x];
    a.blk_part[(int64_t)blockIdx.x * nq + threadIdx.x] = s;
  }
}

// ---- per-AOI terms: one workgroup per (a, c), threads stride the frames ------------------------------------
__global__ __launch_bounds__(256) void tq_aoi_kernel(const tq_cosmos_args a, const int64_t B) {
  __shared__ float s_sum[4][2];
  const int ac = blockIdx.x;  // < nb * C
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ai = ac / a.C, c = ac % a.C;
  float s1 = 0.0f, s2 = 0.0f;
  for (int b = threadIdx.x; b < a.fb; b += 256) {
    const int64_t i = ((int64_t)ai * a.fb + b) * a.C + c;
    s1 += a.aoi_part[i];
    s2 += a.aoi_part[B + i];
  }
  s1 = tq_wave_sum(s1);
  s2 = tq_wave_sum(s2);
  if (lane == 0) {
    s_sum[wave][0] = s1;
    s_sum[wave][1] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float e;
    tq_body_aoi_finish(a, ai, c, (s_sum[0][0] + s_sum[1][0]) + (s_sum[2][0] + s_sum[3][0]),
                       (s_sum[0][1] + s_sum[1][1]) + (s_sum[2][1] + s_sum[3][1]), &e);
    a.aoi_part[2 * B + ac] = e;  // per-AOI prior part of the ELBO (row 2 is scratch, nb*C <= B)
  }
}

// one single-wave workgroup per global site: the fp64 special functions get the full register file
// (no spills, hence no scratch memory: a per-lane scratch request is sized by the runtime for the
// whole device and can push a dispatch onto the slow allocate-per-dispatch path)
__global__ __launch_bounds__(64) void tq_globals_grad_kernel(const tq_cosmos_args a, double* site_elbo) {
  const int s = blockIdx.x;
  if (threadIdx.x == 0) site_elbo[s] = tq_body_globals_grad(a, s);
}

__global__ __launch_bounds__(64) void tq_elbo_finish_kernel(const tq_cosmos_args a, const double* site_elbo) {
  if (threadIdx.x == 0) {
    double eg = 0.0;
    const int ns = tq_num_gsites(a);
    for (int j = 0; j < ns; ++j) eg += site_elbo[j];
    a.elbo_out[0] = a.gsum[TQ_GS_ELBO] + (double)a.global_weight * eg;
  }
}

__global__ __launch_bounds__(256) void tq_adam_kernel(const tq_cosmos_args a, const int64_t first, const int64_t total) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) tq_body_adam(a, j);
}

// lazy Adam: grid.x over the units of the batch (or of the dataset), grid.y = local parameter row
__global__ __launch_bounds__(256) void tq_adam_catchup_kernel(const tq_cosmos_args a, const int64_t n, const int all_units) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t u = all_units ? i : tq_decode_unit(a, i).u;
  tq_adam_replay(a, (int64_t)blockIdx.y * tq_num_units(a) + u, a.last_step[u] + 1, (int)a.step);
}

// single-GPU step: finish of the cross-unit sums + all global sites + total ELBO in ONE workgroup of 4 waves
// (one wave per SIMD, so the fp64 site code keeps the full register file); sites are taken round-robin
__device__ __forceinline__ double tq_wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// global sites (one lane of a wave per site, round-robin over the 4 waves) and the total ELBO from the finished sums
__device__ __forceinline__ void tq_globals_from_gsum_body(const tq_cosmos_args& a, double* s_e) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ns = tq_num_gsites(a);
  if (lane == 0)
    for (int s = wave; s < ns; s += 4) {
      s_e[s] = tq_body_globals_grad(a, s);
#if defined(TQ_MB_STAMPS) && TQ_MB_STAMPS_SITES == 1
      if (a.sync && s < 4) ((uint64_t*)(a.sync + 4))[24 + s] = __builtin_amdgcn_s_memrealtime();
#endif
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    double eg = 0.0;
    for (int j = 0; j < ns; ++j) eg += s_e[j];
    a.elbo_out[0] = a.gsum[TQ_GS_ELBO] + (double)a.global_weight * eg;
  }
}

// cross-unit sums in fp64 by ONE workgroup of 256 threads (s_w: its shared scratch): per-workgroup rows of the unit
// kernel + per-AOI ELBO parts -> gsum
__device__ __forceinline__ void tq_reduce_sums_body(const tq_cosmos_args& a, const int64_t nblk, const int64_t B,
                                                    double (*s_w)[TQ_MAX_NGSUM]) {
  const int nq = tq_num_gsum(a);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every thread walks the rows once, carrying all columns (nq <= 15); then shuffle + 4-way LDS sum
  double acc[TQ_MAX_NGSUM];
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) acc[j] = 0.0;
  for (int64_t r = threadIdx.x; r < nblk; r += 256) {
#pragma unroll
    for (int j = 0; j < TQ_MAX_NGSUM; ++j)
      if (j < nq) acc[j] += (double)a.blk_part[r * nq + j];
  }
  const int nac = a.nb * a.C;
  for (int r = threadIdx.x; r < nac; r += 256) acc[TQ_GS_ELBO] += (double)a.aoi_part[2 * B + r];
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) {
    if (j < nq) {
      const double s = tq_wave_sum_d(acc[j]);
      if (lane == 0) s_w[wave][j] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < nq) a.gsum[threadIdx.x] = s_w[0][threadIdx.x] + s_w[1][threadIdx.x] + s_w[2][threadIdx.x] + s_w[3][threadIdx.x];
}

// sums, then the global sites and the total ELBO (single-GPU steps: no all-reduce in between)
__device__ __forceinline__ void tq_reduce_globals_body(const tq_cosmos_args& a, const int64_t nblk, const int64_t B,
                                                       double (*s_w)[TQ_MAX_NGSUM], double* s_e, const bool with_globals = true) {
  tq_reduce_sums_body(a, nblk, B, s_w);
  __threadfence_block();
  __syncthreads();
  if (with_globals) tq_globals_from_gsum_body(a, s_e);
}

// ---- finish the cross-unit sums in fp64 (single workgroup; sharded runs all-reduce gsum after it) ------------------
__global__ __launch_bounds__(256) void tq_reduce_kernel(const tq_cosmos_args a, const int64_t nblk, const int64_t B) {
  __shared__ double s_w[4][TQ_MAX_NGSUM];
  tq_reduce_sums_body(a, nblk, B, s_w);
}

__global__ __launch_bounds__(256) void tq_reduce_globals_kernel(const tq_cosmos_args a, const int64_t nblk, const int64_t B) {
  __shared__ double s_w[4][TQ_MAX_NGSUM];
  __shared__ double s_e[TQ_NGSITES(TQ_MAXQ)];
  tq_reduce_globals_body(a, nblk, B, s_w, s_e);
}

// AOI-sharded runs: everything of a step that follows the all-reduce of gsum, in one single-workgroup launch -- global
// sites, total ELBO, Adam of the per-AOI / global parameters -- and, if `has_next`, the global draws of the next step.
__global__ __launch_bounds__(256) void tq_tail_reduced_kernel(const tq_cosmos_args a, const tq_cosmos_args next,
                                                              const int has_next) {
  __shared__ double s_e[TQ_NGSITES(TQ_MAXQ)];
  tq_globals_from_gsum_body(a, s_e);
  __syncthreads();
  const int64_t total = tq_num_params(a);
  const int64_t first = a.fuse_adam ? tq_aoi_base(a) : total;  // minibatch steps: the dense Adam is its own launch
  for (int64_t j = first + threadIdx.x; j < total; j += 256) tq_body_adam(a, j);
  if (has_next) {
    __threadfence();
    __syncthreads();
    const int ns = tq_num_gsites(next);
    if ((threadIdx.x & 63) == 0)
      for (int s = threadIdx.x >> 6; s < ns; s += 4) tq_body_sample_globals(next, s);
  }
}

// =============================================================================================================
// Per-unit kernel of full-batch steps (tq_cosmos_step_overlapped / tq_cosmos_step) with the per-AOI frame sums folded in.
//
// The units (f, c) of an AOI are contiguous, so a workgroup of 256 consecutive units touches at most TWO AOIs (when
// F * C >= 256): its row of partial sums carries, next to the cross-unit sums, the sums of
// d/d(background_mean_loc, background_std_loc) over its units of the first AOI (slot 0) and of the second (slot 1).  The
// single-workgroup tail adds the few rows that overlap an AOI itself, so there is no per-AOI kernel (5 us + a launch
// boundary at 400 000 units) and no aoi_part round trip (16 B per unit).  Workgroups stay 1 KiB-aligned in every
// parameter row (AOI-aligned workgroups start at n * F * C and straddle cache lines: 9 % slower, measured).
// Row layout: [2 slots][2 * TQ_MAXQ per-channel AOI partials][nq cross-unit sums]; fixed offsets keep every
// register-array index a compile-time constant.
// =============================================================================================================
#define TQ_ROWS_AOICOL (2 * TQ_MAXQ)
#define TQ_ROWS_GCOL (2 * TQ_ROWS_AOICOL)
#define TQ_ROWS_MAXCOL (TQ_ROWS_GCOL + TQ_MAX_NGSUM)

#include "tq_ksmogn_dev.h"
#include "tq_ksmogn_il2.h"

// (host) does this step use the rows layout?  TAPQIR_AMD_ROWS=0 keeps the flat layout + tq_aoi_kernel (A/B timing)
static bool tq_rows_layout(const tq_cosmos_args& a) {
  static const bool enabled = [] {
    const char* e = getenv("TAPQIR_AMD_ROWS");
    return !(e && e[0] == '0');
  }();
  return enabled && a.fuse_adam && !a.ndx && !a.fdx && a.nb == a.Nt && a.fb == a.F && a.F * a.C >= TQ_UNIT_BLOCK;
}

// Units per workgroup (= per row of partial sums) of the single-launch minibatch step: 16, one 16-lane group each -- or 20,
// the last four with a wave each, when that takes fewer rounds of pixel iterations on the chip's 256 CUs.  A workgroup
// keeps one wave per SIMD busy for 13 iterations of P = 14 (196 pixels on 16 lanes), 17 with 20 units (+ 4: 196 pixels on 64
// lanes); a CU that hosts two workgroups takes twice as long, and the default 10 x 512 minibatch is 320 workgroups of 16
// units -- 64 CUs with two, 26 iterations on the critical path -- but 256 of 20: 17.  With a single camera offset the phase is
// short, but every phase of a workgroup that shares its CU is slower: 49.5 -> 44.7 us per step with 20 (once the gain has its
// own flag; before that the tail workgroup next to a worker delayed everybody and 20 lost, 55.5 against 53.0).  A pure function
// of the batch geometry (TAPQIR_AMD_MB_UNITS = 16 / 20 overrides): the launch that runs the pending tail calls it again.
static int tq_mb_upr(const tq_cosmos_args& a) {
  const char* e = getenv("TAPQIR_AMD_MB_UNITS");  // (read at every call: tests switch it inside one process)
  const int forced = e ? atoi(e) : 0;
  if ((int64_t)a.fb * a.C < 20 || forced == 16) return 16;
  if (forced == 20) return 20;
  const int64_t B = tq_batch_units(a);
  const int64_t r16 = ((B + 15) / 16 + 255) / 256, r20 = ((B + 19) / 20 + 255) / 256;
  return 17 * r20 < 13 * r16 ? 20 : 16;
}
// has_prev code of a pending step for the kernels that run its tail
static int tq_prev_code(const tq_cosmos_args& prev) {
  if (prev.tail_kind == TQ_TAIL_ROWS16) return tq_mb_upr(prev) == 20 ? 7 : 4;
  return tq_rows_layout(prev) ? 3 : 1;
}
// units per row of a step with rows (codes 3 / 4): 16 (single-launch minibatch step), 64 (fused pixel + per-unit
// kernel), TQ_UNIT_BLOCK (tq_unit_rows_kernel)
__host__ __device__ __forceinline__ int tq_rows_upr(const tq_cosmos_args& a) {
  return a.tail_kind == TQ_TAIL_ROWS16 ? 16 : (a.pixel_mode == TQ_PIXEL_FUSED_UNIT ? 64 : TQ_UNIT_BLOCK);
}

template <int K>
__global__ __launch_bounds__(TQ_UNIT_BLOCK) void tq_unit_rows_kernel(const tq_cosmos_args a, const int64_t B) {
  __shared__ float s_part[TQ_UNIT_BLOCK / 64][TQ_ROWS_MAXCOL];
  const int64_t i = (int64_t)blockIdx.x * TQ_UNIT_BLOCK + threadIdx.x;
  const bool live = i < B;
  const uint32_t FC = (uint32_t)(a.F * a.C);
  const uint32_t n0 = ((uint32_t)blockIdx.x * TQ_UNIT_BLOCK) / FC;  // AOI of the workgroup's first unit
  const int nq = tq_num_gsum(a), ncol = TQ_ROWS_GCOL + nq;
  float part[TQ_MAX_NGSUM], aoi[TQ_ROWS_GCOL];
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) part[j] = 0.0f;
#pragma unroll
  for (int j = 0; j < TQ_ROWS_GCOL; ++j) aoi[j] = 0.0f;
  if (live) {
    float aoi2[2];
    tq_body_unit<K>(a, i, part, aoi2);
    const uint32_t n = (uint32_t)i / FC;
    const int c = (int)((uint32_t)i % (uint32_t)a.C);
    const int slot = n == n0 ? 0 : 1;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
      for (int q = 0; q < TQ_MAXQ; ++q) {
        const bool mine = sl == slot && q == c;
        aoi[sl * TQ_ROWS_AOICOL + 2 * q] = mine ? aoi2[0] : 0.0f;
        aoi[sl * TQ_ROWS_AOICOL + 2 * q + 1] = mine ? aoi2[1] : 0.0f;
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < TQ_ROWS_GCOL; ++j) {
    if ((j % TQ_ROWS_AOICOL) < 2 * a.C) {
      const float sum = tq_wave_sum(aoi[j]);
      if (lane == 0) s_part[wave][j] = sum;
    }
  }
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) {
    if (j < nq) {
      const float sum = tq_wave_sum(part[j]);
      if (lane == 0) s_part[wave][TQ_ROWS_GCOL + j] = sum;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < ncol) {
    const bool used = (int)threadIdx.x >= TQ_ROWS_GCOL || ((int)threadIdx.x % TQ_ROWS_AOICOL) < 2 * a.C;
    const float sum = used ? (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]) : 0.0f;
    a.blk_part[(int64_t)blockIdx.x * ncol + threadIdx.x] = sum;
  }
}

// ---- group rows: the tail's sums, spread over the otherwise idle workgroups of the sampling launch --------------------
// At c2 the fused launch leaves 6250 rows (one per wave).  The single-workgroup tail that adds them is a guest of the next
// step's sampling launch, and under that launch's memory traffic each of its ~17 dependent round trips (per-AOI frame
// sums: 16 rows per AOI, two AOIs per thread; cross-unit sums: 25 rows per thread) takes ~2.5 us: the sums alone kept it
// busy for 50 of its 84 us, which made it the last workgroup of the launch.  The grid row of that launch that holds the
// tail workgroup has B / 256 workgroups of which only the first did anything: now workgroup 1 + g of that row adds the rows
// of GROUP g (4096 units: 64 rows of 64 units or 16 of 256) -- one lane per row, one round trip -- and leaves a group row:
// the cross-unit sums in double and the per-AOI frame sums of the (at most 17) AOIs the group touches, published with
// write-through (`sc1`) stores, a drained store queue and an agent-scope counter (MI355X_MICROARCH.md, inter-workgroup
// visibility).  The tail workgroup polls the counter, then reads U / 4096 group rows with `sc1` loads: one round trip for
// the cross-unit sums, one for the per-AOI sums (stamps build: sums complete 9 us after its start instead of 50).  The
// reducers never wait, so the polling workgroup cannot deadlock.
#define TQ_GRP_UNITS 4096
#define TQ_GRP_AOIS (TQ_GRP_UNITS / TQ_UNIT_BLOCK + 1)   /* AOIs a group can touch (F * C >= TQ_UNIT_BLOCK) */
#define TQ_GGROW (2 * 16 + TQ_GRP_AOIS * 2 * TQ_MAXQ)    /* floats of a group row: 16 doubles, then 2 * TQ_MAXQ floats per AOI */
#define TQ_SYNC_LOST 63                                  /* workgroups that gave up waiting for a flag, ever (diagnostics; never observed) */
#define TQ_SYNC_GAIN 62                                  /* the gain of a minibatch launch (float bits), published with the first flag */
#define TQ_SYNC_FLAG2 61                                 /* second flag of a minibatch launch: the global draws after the gain */
#define TQ_SYNC_CLAIM 60                                 /* word that names the workgroup running the tail of a minibatch launch (tail_last) */
#define TQ_SYNC_GROUPS 40                                /* word of tq_cosmos_args.sync that counts the finished groups */
__host__ __device__ __forceinline__ int64_t tq_grp_count(int64_t B) { return (B + TQ_GRP_UNITS - 1) / TQ_GRP_UNITS; }
// group rows follow the rows in blk_part (16-byte aligned)
__host__ __device__ __forceinline__ int64_t tq_grp_base(int64_t nrows, int ncol) { return ((nrows * ncol + 3) / 4) * 4; }

// one wave: rows of group g of step `a` -> group row g (published); returns (lane 0) how many groups had been published before
__device__ __forceinline__ int tq_group_reduce_rows(const tq_cosmos_args& a, const int g) {
  const int lane = threadIdx.x & 63;
  const int nq = tq_num_gsum(a), ncol = TQ_ROWS_GCOL + nq;
  const int64_t B = tq_batch_units(a);
  const uint32_t UPR = (uint32_t)tq_rows_upr(a), RPG = TQ_GRP_UNITS / UPR;
  const int64_t nrows = (B + UPR - 1) / UPR;
  const int64_t r0 = (int64_t)g * RPG;
  const int nw = (int)((nrows - r0) < (int64_t)RPG ? (nrows - r0) : (int64_t)RPG);
  const bool have = lane < nw;
  const float* my = a.blk_part + (r0 + (have ? lane : 0)) * ncol;
  float s0[2 * TQ_MAXQ], s1[2 * TQ_MAXQ], col[TQ_MAX_NGSUM];
#pragma unroll
  for (int j = 0; j < 2 * TQ_MAXQ; ++j) {
    const bool used = j < 2 * a.C;
    s0[j] = (have && used) ? my[j] : 0.0f;
    s1[j] = (have && used) ? my[TQ_ROWS_AOICOL + j] : 0.0f;
  }
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) col[j] = (have && j < nq) ? my[TQ_ROWS_GCOL + j] : 0.0f;
  float* grow = a.blk_part + tq_grp_base(nrows, ncol) + (int64_t)g * TQ_GGROW;
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) {
    if (j < nq) {
      const double sum = tq_wave_sum_d((double)col[j]);
      if (lane == 0) __hip_atomic_store(&((double*)grow)[j], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  const uint32_t FC = (uint32_t)(a.fb * a.C);
  const uint32_t u_first = (uint32_t)g * TQ_GRP_UNITS;
  const uint32_t u_last = (uint32_t)(((int64_t)u_first + TQ_GRP_UNITS - 1 < B - 1) ? u_first + TQ_GRP_UNITS - 1 : B - 1);
  const uint32_t n_lo = u_first / FC, n_hi = u_last / FC;
  const uint32_t n0 = (u_first + UPR * (uint32_t)lane) / FC;  // AOI of this row's first unit (slot 0; slot 1 is the next AOI)
  for (uint32_t n = n_lo; n <= n_hi; ++n) {
    float* out = grow + 32 + (n - n_lo) * (2 * TQ_MAXQ);
#pragma unroll
    for (int j = 0; j < 2 * TQ_MAXQ; ++j) {
      if (j < 2 * a.C) {
        const float v = (n0 == n) ? s0[j] : ((n0 + 1 == n) ? s1[j] : 0.0f);
        const float sum = tq_wave_sum(v);
        if (lane == 0) __hip_atomic_store(&out[j], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the group row has left
  int ticket = 0;
  if (lane == 0) ticket = __hip_atomic_fetch_add(a.sync + TQ_SYNC_GROUPS, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return ticket;
}

// Fused pixel + per-unit kernel of full-batch steps (pixel_mode = TQ_PIXEL_FUSED_UNIT): a wave renders its tile of 64
// units (tq_il2_lane, the routine of tq_ksmogn_il2_kernel) and goes straight on to the per-unit terms + Adam of the same
// 64 units, lane for lane.  The pixel phase is bound by VALU issue (PMC: ~75 % busy) and the per-unit phase by HBM
// (4.3 TB/s of traffic at 33 % VALU busy): as two launches they run one after the other, here the waves of a SIMD are in
// different phases most of the time.  The pixel results go from one phase to the next in registers (56 B per unit
// neither written nor read), a launch boundary is gone, and the row of partial sums is per wave (rows of 64 units).
template <int K, int P>
__global__ __launch_bounds__(64, 2) void tq_pixel_unit_kernel(const tq_ksmogn_args k, const tq_cosmos_args a, const int64_t B) {
  float pixv[TQ_PIXOUT(K)];
#pragma unroll
  for (int j = 0; j < TQ_PIXOUT(K); ++j) pixv[j] = 0.0f;
  tq_il2_lane<K, P, true>(k, B, pixv);
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const uint32_t FC = (uint32_t)(a.F * a.C);
  const uint32_t n0 = ((uint32_t)blockIdx.x * 64u) / FC;  // AOI of the wave's first unit
  const int nq = tq_num_gsum(a), ncol = TQ_ROWS_GCOL + nq;
  float part[TQ_MAX_NGSUM], aoi[TQ_ROWS_GCOL];
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) part[j] = 0.0f;
#pragma unroll
  for (int j = 0; j < TQ_ROWS_GCOL; ++j) aoi[j] = 0.0f;
  if (i < B) {
    float aoi2[2];
    tq_body_unit<K, false, true>(a, i, part, aoi2, pixv);
    const uint32_t n = (uint32_t)i / FC;
    const int c = (int)((uint32_t)i % (uint32_t)a.C);
    const int slot = n == n0 ? 0 : 1;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
      for (int q = 0; q < TQ_MAXQ; ++q) {
        const bool mine = sl == slot && q == c;
        aoi[sl * TQ_ROWS_AOICOL + 2 * q] = mine ? aoi2[0] : 0.0f;
        aoi[sl * TQ_ROWS_AOICOL + 2 * q + 1] = mine ? aoi2[1] : 0.0f;
      }
    }
  }
  float* row = a.blk_part + (int64_t)blockIdx.x * ncol;
#pragma unroll
  for (int j = 0; j < TQ_ROWS_GCOL; ++j) {
    const bool used = (j % TQ_ROWS_AOICOL) < 2 * a.C;
    const float sum = used ? tq_wave_sum(aoi[j]) : 0.0f;
    if (threadIdx.x == 0) row[j] = sum;
  }
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) {
    if (j < nq) {
      const float sum = tq_wave_sum(part[j]);
      if (threadIdx.x == 0) row[TQ_ROWS_GCOL + j] = sum;
    }
  }
}

// (host) launch of the fused kernel; the caller has checked that the step qualifies (tq_fused_pixel_unit)
static bool tq_fused_pixel_unit(const tq_cosmos_args& a) {
  const int64_t B = tq_batch_units(a);
  return a.pixel_mode == TQ_PIXEL_FUSED_UNIT && tq_rows_layout(a) && !a.crosstalk && a.K <= 2 && a.O == 1 && a.pixstats &&
         a.images_il && (a.P == 14 || a.P == 20) && B >= a.il_min_units;
}
static int launch_pixel_unit(const tq_cosmos_args* a, void* stream);

// acc[j] += sum over this thread's rows (r = threadIdx.x, + 256, ...) of column j of the cross-unit sums.  Four rows are
// REQUESTED before any is added: with one row in flight at a time the 25 rows per thread of a c2-sized step with rows of
// 64 units were 25 memory latencies in sequence (~25 us, which made the tail workgroup the last one of the sampling
// launch it hides in).  Same order of additions per thread as the plain loop.
__device__ __forceinline__ void tq_rows_column_sums(const tq_cosmos_args& a, int64_t nrows, int nq, int ncol, double* acc) {
  for (int64_t r0 = threadIdx.x; r0 < nrows; r0 += 4 * 256) {
    float v[4][TQ_MAX_NGSUM];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = r0 + 256 * u;
#pragma unroll
      for (int j = 0; j < TQ_MAX_NGSUM; ++j) v[u][j] = (r < nrows && j < nq) ? a.blk_part[r * ncol + TQ_ROWS_GCOL + j] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < TQ_MAX_NGSUM; ++j)
        if (j < nq) acc[j] += (double)v[u][j];
    }
  }
}

// Tail of a step whose per-unit kernel wrote such rows (ONE workgroup of 256 threads): per-AOI sites from the rows that
// overlap the AOI, cross-unit sums in fp64, global sites and the total ELBO.
// UPR = units per row: TQ_UNIT_BLOCK (tq_unit_rows_kernel) or 16 (the single-launch minibatch step, whose rows hold `mb_upr`
// = 16 or 20 units: the host's tq_mb_upr)
template <int UPR_T>
__device__ __forceinline__ void tq_rows_reduce_globals_body(const tq_cosmos_args& a, double (*s_w)[TQ_MAX_NGSUM], double* s_e,
                                                            const int mb_upr = 16, const bool with_globals = true) {
  const int nq = tq_num_gsum(a), ncol = TQ_ROWS_GCOL + nq;
  const int64_t B = tq_batch_units(a);
  const uint32_t UPR = UPR_T == 16 ? (uint32_t)mb_upr : (uint32_t)tq_rows_upr(a);  // (one instance serves rows of 64 and of 256)
  const int64_t nrows = (B + UPR - 1) / UPR;
  const uint32_t FC = (uint32_t)(a.fb * a.C);  // units of one AOI of the batch
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc[TQ_MAX_NGSUM];
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) acc[j] = 0.0;
  // per-AOI sites: frame sums = sums over the rows that overlap the AOI; prior terms; gradient of the AOI parameters
  const int nac = a.nb * a.C;
  if constexpr (UPR_T == 16) {
    // rows of 16 units: an AOI of the minibatch spans fb C / 16 rows (32 at the default 10 x 512) and this workgroup is
    // the critical path of the step -- 16 lanes share the rows of one (AOI, channel), so the loads of an AOI are two
    // round trips instead of 32 in sequence
    const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;
    for (int ac0 = 0; ac0 < nac; ac0 += 16) {
      const int ac = ac0 + grp;
      const bool on = ac < nac;
      const uint32_t ai = on ? (uint32_t)ac / (uint32_t)a.C : 0u;
      const int c = on ? ac - (int)ai * a.C : 0;
      float s1 = 0.0f, s2 = 0.0f;
      if (on) {
        const uint32_t r_lo = (ai * FC) / UPR, r_hi = ((ai + 1) * FC - 1) / UPR;
        for (uint32_t r = r_lo + gl; r <= r_hi; r += 16) {
          const int slot = (r * UPR) / FC == ai ? 0 : 1;
          const float* row = a.blk_part + (int64_t)r * ncol + slot * TQ_ROWS_AOICOL + 2 * c;
          s1 += row[0];
          s2 += row[1];
        }
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 16);
        s2 += __shfl_xor(s2, o, 16);
      }
      if (on && gl == 0) {
        float e;
        tq_body_aoi_finish(a, (int)ai, c, s1, s2, &e);
        acc[TQ_GS_ELBO] += (double)e;
      }
    }
#ifdef TQ_MB_STAMPS
    if (threadIdx.x == 0) ((uint64_t*)(a.sync + 4))[13] = __builtin_amdgcn_s_memrealtime();
#endif
  } else {
    for (int ac = threadIdx.x; ac < nac; ac += 256) {
      const uint32_t ai = (uint32_t)ac / (uint32_t)a.C;  // position of the AOI in the batch
      const int c = ac - (int)ai * a.C;
      const uint32_t r_lo = (ai * FC) / UPR, r_hi = ((ai + 1) * FC - 1) / UPR;
      float s1 = 0.0f, s2 = 0.0f;
      for (uint32_t rb = r_lo; rb <= r_hi; rb += 4) {  // four rows requested before any is added (see tq_rows_column_sums)
        float p1[4], p2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t r = rb + u;
          const int slot = (r * UPR) / FC == ai ? 0 : 1;
          const float* row = a.blk_part + (int64_t)r * ncol + slot * TQ_ROWS_AOICOL + 2 * c;
          p1[u] = r <= r_hi ? row[0] : 0.0f;
          p2[u] = r <= r_hi ? row[1] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s1 += p1[u];
          s2 += p2[u];
        }
      }
      float e;
      tq_body_aoi_finish(a, (int)ai, c, s1, s2, &e);
      acc[TQ_GS_ELBO] += (double)e;
    }
  }
  if constexpr (UPR_T == 16) {  // a few hundred rows: the plain loop (and no extra registers in the minibatch kernel)
    for (int64_t r = threadIdx.x; r < nrows; r += 256) {
#pragma unroll
      for (int j = 0; j < TQ_MAX_NGSUM; ++j)
        if (j < nq) acc[j] += (double)a.blk_part[r * ncol + TQ_ROWS_GCOL + j];
    }
  } else {
    tq_rows_column_sums(a, nrows, nq, ncol, acc);
  }
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) {
    if (j < nq) {
      const double s = tq_wave_sum_d(acc[j]);
      if (lane == 0) s_w[wave][j] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < nq) a.gsum[threadIdx.x] = s_w[0][threadIdx.x] + s_w[1][threadIdx.x] + s_w[2][threadIdx.x] + s_w[3][threadIdx.x];
  __threadfence_block();
  __syncthreads();
#ifdef TQ_MB_STAMPS
  if (threadIdx.x == 0 && a.sync) ((uint64_t*)(a.sync + 4))[12] = __builtin_amdgcn_s_memrealtime();
#endif
  if (with_globals) tq_globals_from_gsum_body(a, s_e);
}

// Sums of a step from the group rows that other workgroups publish (ONE workgroup of 256 threads): WAIT: polls the counter of
// published groups first (the tail workgroup inside a sampling launch; returns false after ~2 s without them: never
// observed, the caller leaves a NaN loss) -- else the caller knows they are all there (the last workgroup of
// tq_group_sums_kernel).  Then per-AOI sites from the one or two groups that overlap the AOI and the cross-unit sums -> gsum.
template <bool WAIT>
__device__ __forceinline__ bool tq_groups_sums_body(const tq_cosmos_args& a, double (*s_w)[TQ_MAX_NGSUM]) {
  const int nq = tq_num_gsum(a), ncol = TQ_ROWS_GCOL + nq;
  const int64_t B = tq_batch_units(a);
  const uint32_t UPR = (uint32_t)tq_rows_upr(a);
  const int64_t nrows = (B + UPR - 1) / UPR;
  const int ngroups = (int)tq_grp_count(B);
  const float* grows = a.blk_part + tq_grp_base(nrows, ncol);
  const uint32_t FC = (uint32_t)(a.fb * a.C);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    int ok = 1;
    if (WAIT) {
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(a.sync + TQ_SYNC_GROUPS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ngroups) {
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
          ok = 0;
          break;
        }
      }
    }
    __hip_atomic_store(a.sync + TQ_SYNC_GROUPS, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
    s_ok = ok;
  }
  __syncthreads();
  if (!s_ok) return false;
  auto ld = [](const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto ldd = [](const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  double acc[TQ_MAX_NGSUM];
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) acc[j] = 0.0;
  const int nac = a.nb * a.C;
  for (int ac = threadIdx.x; ac < nac; ac += 256) {
    const uint32_t ai = (uint32_t)ac / (uint32_t)a.C;
    const int c = ac - (int)ai * a.C;
    const uint32_t g_lo = (ai * FC) / TQ_GRP_UNITS, g_hi = ((ai + 1) * FC - 1) / TQ_GRP_UNITS;
    float s1 = 0.0f, s2 = 0.0f;
    for (uint32_t g = g_lo; g <= g_hi; ++g) {
      const uint32_t m = ai - (g * TQ_GRP_UNITS) / FC;
      const float* p = grows + (int64_t)g * TQ_GGROW + 32 + m * (2 * TQ_MAXQ) + 2 * c;
      s1 += ld(p);
      s2 += ld(p + 1);
    }
    float e;
    tq_body_aoi_finish(a, (int)ai, c, s1, s2, &e);
    acc[TQ_GS_ELBO] += (double)e;
  }
  for (int g = threadIdx.x; g < ngroups; g += 256) {
    const double* p = (const double*)(grows + (int64_t)g * TQ_GGROW);
#pragma unroll
    for (int j = 0; j < TQ_MAX_NGSUM; ++j)
      if (j < nq) acc[j] += ldd(p + j);
  }
#ifdef TQ_MB_STAMPS
  if (threadIdx.x == 0 && a.sync) ((uint64_t*)(a.sync + 4))[9] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) {
    if (j < nq) {
      const double s = tq_wave_sum_d(acc[j]);
      if (lane == 0) s_w[wave][j] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < nq) a.gsum[threadIdx.x] = s_w[0][threadIdx.x] + s_w[1][threadIdx.x] + s_w[2][threadIdx.x] + s_w[3][threadIdx.x];
  return true;
}
// ... and the global sites + total ELBO (the tail workgroup of a sampling launch)
__device__ __forceinline__ bool tq_groups_reduce_globals_body(const tq_cosmos_args& a, double (*s_w)[TQ_MAX_NGSUM], double* s_e) {
  if (!tq_groups_sums_body<true>(a, s_w)) return false;
  __threadfence_block();
  __syncthreads();
#ifdef TQ_MB_STAMPS
  if (threadIdx.x == 0 && a.sync) ((uint64_t*)(a.sync + 4))[12] = __builtin_amdgcn_s_memrealtime();
#endif
  tq_globals_from_gsum_body(a, s_e);
  return true;
}

// AOI-sharded full-batch steps (and tq_cosmos_tail): rows -> per-AOI sites and gsum, what the all-reduce needs.  Workgroup g
// adds the rows of group g (one wave, one round trip) and publishes the group row; the workgroup whose ticket is the last
// one finishes the per-AOI sites and the cross-unit sums from the U / 4096 group rows (tq_rows_sums_kernel, one wave per AOI
// and a last workgroup that walked all 6250 rows, took 38 us at c2 -- on the critical path of every sharded step).
__global__ __launch_bounds__(256) void tq_group_sums_kernel(const tq_cosmos_args a) {
  __shared__ double s_w[4][TQ_MAX_NGSUM];
  __shared__ int s_last;
  if (threadIdx.x < 64) {
    const int ticket = tq_group_reduce_rows(a, (int)blockIdx.x);
    if (threadIdx.x == 0) s_last = ticket == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  tq_groups_sums_body<false>(a, s_w);
}


// AOI-sharded full-batch steps: rows -> per-AOI sites and gsum (what the all-reduce needs), nothing of the global
// sites.  One wave per (AOI, channel) adds the rows that overlap the AOI (a single workgroup would walk 400 AOIs x 17 rows
// of 64 units in sequence, on the critical path of the step); the LAST workgroup to finish -- a device-scope ticket
// after a release fence -- sums the rows and the per-AOI ELBO parts in fp64, in a fixed order.
__global__ __launch_bounds__(256) void tq_rows_sums_kernel(const tq_cosmos_args a) {
  __shared__ double s_w[4][TQ_MAX_NGSUM];
  __shared__ int s_last;
  const int nq = tq_num_gsum(a), ncol = TQ_ROWS_GCOL + nq;
  const int64_t B = tq_batch_units(a);
  const uint32_t UPR = (uint32_t)tq_rows_upr(a);
  const int64_t nrows = (B + UPR - 1) / UPR;
  const uint32_t FC = (uint32_t)(a.fb * a.C);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nac = a.nb * a.C;
  const int ac = (int)blockIdx.x * 4 + wave;
  if (ac < nac) {
    const uint32_t ai = (uint32_t)ac / (uint32_t)a.C;
    const int c = ac - (int)ai * a.C;
    const uint32_t r_lo = (ai * FC) / UPR, r_hi = ((ai + 1) * FC - 1) / UPR;
    float s1 = 0.0f, s2 = 0.0f;
    for (uint32_t r = r_lo + lane; r <= r_hi; r += 64) {
      const int slot = (r * UPR) / FC == ai ? 0 : 1;
      const float* row = a.blk_part + (int64_t)r * ncol + slot * TQ_ROWS_AOICOL + 2 * c;
      s1 += row[0];
      s2 += row[1];
    }
    s1 = tq_wave_sum(s1);
    s2 = tq_wave_sum(s2);
    if (lane == 0) {
      float e;
      tq_body_aoi_finish(a, (int)ai, c, s1, s2, &e);
      a.aoi_part[2 * B + ac] = e;  // (rows 0 and 1 of aoi_part belong to the flat layout; row 2 is scratch, nb*C <= B)
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(&a.sync[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  double acc[TQ_MAX_NGSUM];
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) acc[j] = 0.0;
  tq_rows_column_sums(a, nrows, nq, ncol, acc);
  for (int r = threadIdx.x; r < nac; r += 256) acc[TQ_GS_ELBO] += (double)a.aoi_part[2 * B + r];
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) {
    if (j < nq) {
      const double s = tq_wave_sum_d(acc[j]);
      if (lane == 0) s_w[wave][j] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < nq) a.gsum[threadIdx.x] = s_w[0][threadIdx.x] + s_w[1][threadIdx.x] + s_w[2][threadIdx.x] + s_w[3][threadIdx.x];
  if (threadIdx.x == 0) __hip_atomic_store(&a.sync[3], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed
}

__global__ __launch_bounds__(256) void tq_rows_reduce_globals_kernel(const tq_cosmos_args a, const int upr) {
  __shared__ double s_w[4][TQ_MAX_NGSUM];
  __shared__ double s_e[TQ_NGSITES(TQ_MAXQ)];
  if (upr <= 20) tq_rows_reduce_globals_body<16>(a, s_w, s_e, upr);
  else tq_rows_reduce_globals_body<TQ_UNIT_BLOCK>(a, s_w, s_e);
}

// Full-batch pipeline (tq_cosmos_step_overlapped): the local guide sampling of step t, with ONE extra workgroup (block (0, 0),
// dispatched first) that runs the single-workgroup tail of step t-1 -- cross-unit sums, global sites, total ELBO, Adam
// of the per-AOI / global parameters -- and then draws the global sites of step t from the updated parameters.  The
// sampling of the local sites reads local parameters only (already updated by the Adam fused into the unit kernel of
// step t-1), so the ~35 us latency chain of the tail hides behind the ~14 000 sampling workgroups of the same launch.
// Compiled for the occupancy of the SAMPLING path (five waves per SIMD, 96 registers; the fp64 code of the global sites
// spills ~800 registers to scratch at that cap, which the one tail workgroup can afford now that the other workgroups of
// its grid row add the rows for it: with the tail adding all rows itself it was the last workgroup of the launch and the
// kernel had to be built for three waves -- c2 step 0.248 -> 0.235 ms, 0.320 -> 0.288 ms in the regime of a converged fit).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void tq_sample_locals_tail_kernel(
    const tq_cosmos_args a, const tq_cosmos_args prev, const int has_prev, const int64_t B, const int site_begin) {
  // has_prev: 0 = nothing pending, 1 = the whole tail of `prev` (cross-unit sums first; 3 = the same from rows of 64 / 256 units
  // with the per-AOI sites folded in; 6 = the same with the rows added per group of 4096 units by the other workgroups of
  // this grid row), 2 = gsum of `prev` is complete (all-reduced by the caller): global sites onwards
  if (blockIdx.y == 0) {
    if (blockIdx.x != 0) {
      // has_prev == 6: workgroup 1 + g adds the rows of group g of `prev` for the tail workgroup (one wave; the others leave)
      if (has_prev == 6 && (int64_t)blockIdx.x <= tq_grp_count(tq_batch_units(prev)) && threadIdx.x < 64)
        tq_group_reduce_rows(prev, (int)blockIdx.x - 1);
      return;
    }
    __shared__ double s_w[4][TQ_MAX_NGSUM];
    __shared__ double s_e[TQ_NGSITES(TQ_MAXQ)];
#ifdef TQ_MB_STAMPS  // (diagnostic build, scripts/fb_tail_time.py: how long the tail workgroup of a full-batch step lives)
    if (threadIdx.x == 0 && a.sync) ((uint64_t*)(a.sync + 4))[8] = __builtin_amdgcn_s_memrealtime();
#endif
    if (has_prev) {
      const int64_t Bp = tq_batch_units(prev);
      if (has_prev == 3) tq_rows_reduce_globals_body<TQ_UNIT_BLOCK>(prev, s_w, s_e);
      else if (has_prev == 6) {
        if (!tq_groups_reduce_globals_body(prev, s_w, s_e) && threadIdx.x == 0) prev.elbo_out[0] = __builtin_nan("");
      } else if (has_prev == 1) tq_reduce_globals_body(prev, (Bp + TQ_UNIT_BLOCK - 1) / TQ_UNIT_BLOCK, Bp, s_w, s_e);
      else tq_globals_from_gsum_body(prev, s_e);
      __syncthreads();
#ifdef TQ_MB_STAMPS
      if (threadIdx.x == 0 && a.sync) ((uint64_t*)(a.sync + 4))[7] = __builtin_amdgcn_s_memrealtime();
#endif
      const int64_t total = tq_num_params(prev);
      for (int64_t j = tq_aoi_base(prev) + threadIdx.x; j < total; j += 256) tq_body_adam(prev, j);
      __threadfence();
      __syncthreads();
    }
#ifdef TQ_MB_STAMPS
    if (threadIdx.x == 0 && a.sync) ((uint64_t*)(a.sync + 4))[13] = __builtin_amdgcn_s_memrealtime();
#endif
    const int ns = tq_num_gsites(a);
    if ((threadIdx.x & 63) == 0)
      for (int s = threadIdx.x >> 6; s < ns; s += 4) tq_body_sample_globals(a, s);
#ifdef TQ_MB_STAMPS
    __syncthreads();
    if (threadIdx.x == 0 && a.sync) ((uint64_t*)(a.sync + 4))[11] = __builtin_amdgcn_s_memrealtime();
#endif
    return;
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  tq_sample_site_wg(a, site_begin + (int)blockIdx.y - 1, i, B);
#ifdef TQ_MB_STAMPS
  if (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && threadIdx.x == 0 && a.sync)
    ((uint64_t*)(a.sync + 4))[10] = __builtin_amdgcn_s_memrealtime();  // (about) the last sampling workgroup
#endif
}


// =============================================================================================================
// Single-launch minibatch step (tq_cosmos_minibatch_step).
//
// The reference's default operating point (10 AOIs x 512 frames, main.py:1428-1431) is 5120 units: 80 waves of work for
// a chip with 4096 wave slots.  As five launches (lazy-Adam catch-up, site draws + previous tail, likelihood, per-unit,
// per-AOI) a step costs five launch latencies on the device (70 us) and about as much on the host, which becomes the
// bottleneck.  Here ONE launch runs a step; a workgroup owns U = 16 or 20 units (tq_mb_upr) through
// all phases:
//   tail workgroup: tail of the PREVIOUS step (cross-unit sums, per-AOI sites, global sites, ELBO, Adam of the per-AOI /
//                   global parameters) and the global draws of this step, the GAIN's chain first: flag 1 (device-scope
//                   release) when the gain is drawn, flag 2 when the other draws are; then the next step's subsample
//                   (tq_draw_subsample).  Ticket 0, or the block dispatched last if it claims the role (tail_last);
//   phase 1       : lazy-Adam catch-up of the U units' local parameters, then their 9 x U guide-site draws;
//   (wait 1)      : one lane polls flag 1 (the likelihood needs this step's gain);
//   phase 2       : the 16-lanes-per-unit likelihood routine of tq_ksmogn_kernel (tq_ksmogn_tile_at); of 20 units the
//                   last four with a wave each;
//   (wait 2)      : flag 2 (the per-unit terms need the tables of pi, lamda, proximity: set long before);
//   phase 3       : per-unit ELBO terms, gradients and Adam (one lane per unit), row of partial sums with the per-AOI
//                   frame sums folded in (rows of U units, tq_rows_reduce_globals_body<16>).
// Phases hand data over through the step workspace in global memory; a workgroup lives on one CU, whose L1 its waves
// share, so a workgroup barrier orders those accesses.  The tail workgroup never waits for a workgroup that may not be
// resident yet, so the waiting workgroups cannot deadlock whatever the dispatch order.  The tail of THIS step runs in the next launch (or in
// tq_cosmos_tail).
// =============================================================================================================
// The next step's subsample, drawn by the tail workgroup of a minibatch launch: `take` of `n` indices without replacement =
// the indices of the `take` smallest of n Philox keys (stream: seed, step, site; ties broken by the index).  The law of
// randperm(n)[:take] (pyro.plate's subsample, cosmos.py:194-208) up to the order of the selected indices, which no sum depends
// on.  256 threads, n <= TQ_SUBSAMPLE_MAX = 2048 (eight keys per thread, in registers); `hist` holds 2048 + 8 int32 words.
//
// Selection by radix instead of a sort (a bitonic sort of 1024 keys in LDS is 55 barrier-separated stages, ~2.5 us of every
// step): a histogram of the top 11 bits of the 43-bit composite (key << 11 | index), a scan over its bins to the bin that
// holds the take-th smallest, and -- only if that bin is not taken whole -- the same again on the next 11 bits inside it
// (random keys: the boundary bin holds one or two elements, so one or two levels).  Then the selected indices are
// compacted in (thread, slot) order through a second scan: the output does not depend on the timing of any atomic.
__device__ __forceinline__ int tq_block_exscan(int v, int* s_w) {  // exclusive prefix sum over the 256 threads; s_w: 4 words
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  __syncthreads();  // (the previous use of s_w has been read)
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w)
    if (w < wave) base += s_w[w];
  return base + inc - v;
}
__device__ __forceinline__ void tq_draw_subsample(int* hist, uint64_t seed, uint32_t step, uint32_t site, int n, int take,
                                                  int32_t* out) {
  constexpr int PER = TQ_SUBSAMPLE_MAX / 256;
  int* s_w = hist + 2048;      // 4 words of the scans
  int* s_bnd = hist + 2048 + 4;  // boundary bin, elements below it, elements in it
  const int tid = threadIdx.x;
  uint64_t c[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid + 256 * j;
    c[j] = ~0ull;
    if (i < n) {
      TqPhilox s;
      tq_philox_init(&s, seed, step, site, (uint64_t)i);
      c[j] = ((uint64_t)tq_philox_next(&s) << 11) | (uint32_t)i;
    }
  }
  uint64_t path = 0, T = 0;
  int need = take;
  for (int level = 0; level < 4; ++level) {
    const int shift = level == 0 ? 32 : (level == 1 ? 21 : (level == 2 ? 10 : 0));
    const int width = level == 3 ? 10 : 11;
    for (int b = tid; b < 2048; b += 256) hist[b] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j)
      if (tid + 256 * j < n && (c[j] >> (shift + width)) == path) atomicAdd(&hist[(int)((c[j] >> shift) & ((1u << width) - 1u))], 1);
    __syncthreads();
    int cnt[8], local = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      cnt[q] = hist[8 * tid + q];
      local += cnt[q];
    }
    int run = tq_block_exscan(local, s_w);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (run < need && run + cnt[q] >= need) {
        s_bnd[0] = 8 * tid + q;
        s_bnd[1] = run;
        s_bnd[2] = cnt[q];
      }
      run += cnt[q];
    }
    __syncthreads();
    const int b = s_bnd[0], below = s_bnd[1], inbin = s_bnd[2];
    path = (path << width) | (uint64_t)b;
    need -= below;
    if (inbin == need) {  // the boundary bin is taken whole (always at the last level: composites are distinct)
      T = (path + 1) << shift;
      break;
    }
    __syncthreads();  // (s_bnd is rewritten at the next level)
  }
  int mine = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) mine += (tid + 256 * j < n && c[j] < T) ? 1 : 0;
  int at = tq_block_exscan(mine, s_w);
#pragma unroll
  for (int j = 0; j < PER; ++j)
    if (tid + 256 * j < n && c[j] < T) out[at++] = (int32_t)(tid + 256 * j);
  __syncthreads();  // (hist is reused by the next draw)
}
#define TQ_SITE_SUBSAMPLE_N 0xA00u
#define TQ_SITE_SUBSAMPLE_F 0xA01u

// Lazy-Adam replay of ONE element by G neighbouring lanes (the last, thinly filled pass of the catch-up phase: 32 of 288
// elements at K = 2, which cost the workgroup a second full pass of ~136 dependent steps on one wave).  With zero gradient
// the increment of step s0 + k depends on (m0 beta1^k, v0 beta2^k) and the step's bias factors only, not on the parameter:
// lane `part` starts from the moments after k0 = part * ceil(n / G) steps (closed form), adds up the increments of its own
// steps, the G sums are added and the parameter moves once.  Against the step-by-step form the sum is rounded once instead
// of at every step (a few ulp of the parameter) and no increment is dropped as negligible.  Steps older than the bias
// table take the plain replay on the first lane.
template <int G>
__device__ __forceinline__ void tq_adam_replay_split(const tq_cosmos_args& a, int64_t j, int s0, int s1, const float* tab, int T0,
                                                     float p, float m, float v, int part) {
  const bool valid = j >= 0 && s0 <= s1;
  const bool direct = valid && s0 < T0;
  const int n = valid ? s1 - s0 + 1 : 0;
  float dp = 0.0f;
  if (valid && !direct) {
    const int L = (n + G - 1) / G;
    const int k0 = part * L;
    const int k1 = k0 + L < n ? k0 + L : n;
    if (k0 < k1) {
      float mm = m * (float)tq_powi((double)a.beta1, k0), vv = v * (float)tq_powi((double)a.beta2, k0);
      const float* t = tab + 2 * (s0 + k0 - T0);
#pragma unroll 4
      for (int k = k0; k < k1; ++k, t += 2) {
        mm = a.beta1 * mm;
        vv = a.beta2 * vv;
        dp += t[0] * mm * TQ_FRCP(TQ_FSQRT(vv) * t[1] + a.adam_eps);
      }
    }
  }
#pragma unroll
  for (int o = 1; o < G; o <<= 1) dp += __shfl_xor(dp, o, 64);
  if (direct) {
    if (part == 0) tq_adam_replay_tab_given(a, j, s0, s1, tab, T0, p, m, v);
  } else if (valid && part == 0) {
    a.params[j] = p - dp;
    a.exp_avg[j] = m * (float)tq_powi((double)a.beta1, n);
    a.exp_avg_sq[j] = v * (float)tq_powi((double)a.beta2, n);
  }
}

#ifdef TQ_MB_STAMPS
// (diagnostic) where a workgroup runs: XCC (4 bits) | SE, SH, CU of HW_ID (8 bits) | block (10 bits) | ticket (10 bits)
__device__ __forceinline__ unsigned long long tq_where(unsigned block, int ticket) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return ((unsigned long long)(xcc & 15) << 28) | (((hw >> 8) & 0xff) << 20) | ((block & 1023) << 10) | ((unsigned)ticket & 1023);
}
#endif
template <int K, bool ONE, int U>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void tq_minibatch_kernel(
    const tq_cosmos_args a, const tq_cosmos_args prev, const int has_prev, const tq_ksmogn_args k, const int64_t B,
    const int tail_last) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_ticket, s_ok, s_role;
  __shared__ float s_part[4][TQ_ROWS_MAXCOL];
  const int tid = threadIdx.x;
#ifdef TQ_MB_STAMPS
  uint64_t tq_tloc[8];  // (every workgroup keeps its own stamps too: maxima over the grid at stamp 5, words 16..21)
#define TQ_STAMP(n)                                                                   \
  if (tid == 0) {                                                                     \
    tq_tloc[n] = __builtin_amdgcn_s_memrealtime();                                    \
    if (blockIdx.x == TQ_MB_STAMPS) ((uint64_t*)(a.sync + 4))[n] = tq_tloc[n];        \
    if (n == 5) {                                                                     \
      unsigned long long* mx = (unsigned long long*)(a.sync + 4) + 16;                \
      atomicMax(mx + 0, (unsigned long long)(tq_tloc[1] - tq_tloc[0]));               \
      atomicMax(mx + 1, (unsigned long long)(tq_tloc[2] - tq_tloc[1]));               \
      atomicMax(mx + 2, (unsigned long long)(tq_tloc[3] - tq_tloc[2]));               \
      atomicMax(mx + 3, (unsigned long long)(tq_tloc[4] - tq_tloc[3]));               \
      atomicMax(mx + 4, (unsigned long long)(tq_tloc[5] - tq_tloc[4]));               \
      atomicMax(mx + 5, (unsigned long long)(tq_tloc[5] - tq_tloc[0]));               \
      atomicMax(mx + 6, ((unsigned long long)(tq_tloc[5] - tq_tloc[0]) << 32) | tq_where(blockIdx.x, s_ticket)); \
    }                                                                                 \
  }
#define TQ_STAMP2(n) if (tid == 0 && blockIdx.x == TQ_MB_STAMPS) ((uint64_t*)(a.sync + 4))[24 + n] = __builtin_amdgcn_s_memrealtime();
#define TQ_TAIL_STAMP(n)                                                                     \
  if (tid == 0) {                                                                            \
    ((uint64_t*)(a.sync + 4))[n] = __builtin_amdgcn_s_memrealtime();                         \
    if (n == 8) ((uint64_t*)(a.sync + 4))[23] = tq_where(blockIdx.x, s_ticket);              \
  }
#else
#define TQ_STAMP(n)
#define TQ_STAMP2(n)
#define TQ_TAIL_STAMP(n)
#endif
  TQ_STAMP(0)
  if (tid == 0) s_ticket = __hip_atomic_fetch_add(&a.sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  TQ_STAMP(6)
  const int ticket = s_ticket;
  const int flag_value = a.sync_value;
  // every workgroup counts itself out exactly once; the last one re-arms the ticket counter for the next launch
  auto count_out = [&]() {
    const int done = __hip_atomic_fetch_add(&a.sync[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (int)gridDim.x - 1) {
      __hip_atomic_store(&a.sync[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.sync[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.sync[TQ_SYNC_CLAIM], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  if (ticket >= (int)gridDim.x) {
    // the counter did not start from zero (an earlier launch was torn down before it could re-arm it): this launch has
    // no valid work split.  Touch nothing, leave a NaN loss, and re-arm so that the next launch is whole again.
    if (tid == 0) {
      a.elbo_out[0] = __builtin_nan("");
      count_out();
    }
    return;
  }
  // Who runs the tail.  Ticket 0 by default: that workgroup is resident whatever else the chip is doing, and it waits for
  // nobody.  With `tail_last` (a grid of 256 k + 1 workgroups: every CU hosts k of them and ONE hosts k + 1) the LAST block
  // of the grid -- dispatched last, so the one that doubles up on a CU -- claims the tail if it gets there within a few
  // microseconds: the tail is short, and the workgroup it shares the CU with keeps its SIMDs to itself for the long
  // likelihood phase (two workers on one CU take twice as long there and ARE the critical path of the launch).  The
  // ticket-0 workgroup then takes over the units of the claimer.  It polls the claim for a bounded time only and
  // claims the tail itself when nothing arrives: progress never depends on a workgroup that is not resident yet.
  int work = ticket - 1;
  bool is_tail = ticket == 0;
  if (tail_last) {
    if (tid == 0) {
      int role = ticket == 0 ? -1 : ticket - 1;  // -1: the tail
      if (blockIdx.x == gridDim.x - 1 && ticket != 0) {
        int expected = 0;
        if (__hip_atomic_compare_exchange_strong(&a.sync[TQ_SYNC_CLAIM], &expected, ticket + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT))
          role = -1;
      } else if (ticket == 0 && blockIdx.x != gridDim.x - 1) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        int c;
        while ((c = __hip_atomic_load(&a.sync[TQ_SYNC_CLAIM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 &&
               __builtin_amdgcn_s_memrealtime() - t0 < 600ull)  // 6 us of the 100 MHz clock
          __builtin_amdgcn_s_sleep(2);
        if (c == 0 && !__hip_atomic_compare_exchange_strong(&a.sync[TQ_SYNC_CLAIM], &c, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT)) {
          // (lost the race at the last moment: c now holds the claimer's ticket + 1)
        }
        if (c != 0) role = c - 2;  // the units the claimer would have had
      }
      s_role = role;
    }
    __syncthreads();
    work = s_role;
    is_tail = work < 0;
  }
  if (is_tail) {  // the extra workgroup of the grid: owns no units
    __shared__ double s_w[4][TQ_MAX_NGSUM];
    __shared__ double s_e[TQ_NGSITES(TQ_MAXQ)];
    TQ_TAIL_STAMP(8)
    // The workers need the GAIN of this step before their likelihood phase and the other global draws (tables of pi, lamda,
    // proximity) only in the per-unit phase after it: the gain's chain -- gradient of its site, Adam of its two
    // parameters, the draw -- runs on wave 0 by itself and is published first (sync[1]); the other sites' gradients
    // (7-8 us each against 3.5) run beside it on waves 1..3, and their Adam and draws follow under a second flag
    // (sync[TQ_SYNC_FLAG2]) that is long set when a worker gets to it.
    const int lane = tid & 63, wave = tid >> 6;
    if (has_prev) {
      const int64_t Bp = tq_batch_units(prev);
      if (has_prev == 3) tq_rows_reduce_globals_body<TQ_UNIT_BLOCK>(prev, s_w, s_e, 16, false);
      else if (has_prev == 4 || has_prev == 7) tq_rows_reduce_globals_body<16>(prev, s_w, s_e, has_prev == 7 ? 20 : 16, false);
      else tq_reduce_globals_body(prev, (Bp + TQ_UNIT_BLOCK - 1) / TQ_UNIT_BLOCK, Bp, s_w, s_e, false);
      // (the bodies end with gsum stored, a workgroup-scope fence and a barrier)
      if (lane == 0) {
        const int nsp = tq_num_gsites(prev);
        if (wave == 0) {
          s_e[0] = tq_body_globals_grad(prev, 0);
          const int64_t gb = tq_global_base(prev);  // [0] gain_loc [1] gain_beta (tq_globals.h)
          tq_body_adam(prev, gb);
          tq_body_adam(prev, gb + 1);
        } else {
          for (int sg = wave; sg < nsp; sg += 3) s_e[sg] = tq_body_globals_grad(prev, sg);
        }
      }
    }
    if (wave == 0) {
      if (lane == 0) {
        tq_body_sample_globals(a, 0);
        // publish: the gain once more in a sync word (what the workers read: wait_flag), then the storing lane drains,
        // releases at device scope and sets the flag
        __hip_atomic_store(&a.sync[TQ_SYNC_GAIN], __float_as_int(((const TqGlobals*)a.globals)->gain), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&a.sync[1], flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    TQ_TAIL_STAMP(9)
    if (has_prev) {
      if (tid == 0) {  // total ELBO of the previous step (as tq_globals_from_gsum_body)
        const int nsp = tq_num_gsites(prev);
        double eg = 0.0;
        for (int j = 0; j < nsp; ++j) eg += s_e[j];
        prev.elbo_out[0] = prev.gsum[TQ_GS_ELBO] + (double)prev.global_weight * eg;
      }
      const int64_t total = tq_num_params(prev), gb = tq_global_base(prev);
      for (int64_t j = tq_aoi_base(prev) + tid; j < total; j += 256)
        if (j != gb && j != gb + 1) tq_body_adam(prev, j);
      __threadfence_block();
      __syncthreads();
    }
    TQ_TAIL_STAMP(10)
    const int ns = tq_num_gsites(a);
    if (lane == 0)
      for (int sg = 1 + wave; sg < ns; sg += 4) tq_body_sample_globals(a, sg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&a.sync[TQ_SYNC_FLAG2], flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    TQ_TAIL_STAMP(11)
    // the NEXT step's subsample (nobody waits for it: the next launch reads it)
    if (a.next_ndx && a.nb < a.Nt) {
      tq_draw_subsample((int*)smem, a.seed, a.step + 1, TQ_SITE_SUBSAMPLE_N, a.Nt, a.nb, a.next_ndx);
      __syncthreads();
    }
    if (a.next_fdx && a.fb < a.F) tq_draw_subsample((int*)smem, a.seed, a.step + 1, TQ_SITE_SUBSAMPLE_F, a.F, a.fb, a.next_fdx);
    if (tid == 0) count_out();
    return;
  }
  // ---- phase 1: catch-up + site draws of this workgroup's U units (work index = ticket - 1) ----
  const int64_t wblk = work;
  const int64_t u0 = wblk * U;
  const int64_t u_end = u0 + U < B ? u0 + U : B;
  constexpr int NL = TQ_NLOCAL(K), NS = 1 + 4 * K;
  TQ_STAMP(7)
  if (a.last_step) {
    // per-step bias-correction factors of the last TQ_BIAS_TABLE_STEPS steps, shared by every element of the workgroup
    __shared__ float s_bias[2 * TQ_BIAS_TABLE_STEPS];
    const int s1 = (int)a.step;
    const int T0 = s1 - (TQ_BIAS_TABLE_STEPS - 1) > 1 ? s1 - (TQ_BIAS_TABLE_STEPS - 1) : 1;
    // the chain of dependent loads of every element of this thread (subsample index -> unit -> last step -> values) is
    // issued first and overlaps with the table build
    constexpr int NPASS = (NL * U + 255) / 256;
    // a thin last pass is shared out: G lanes per element (tq_adam_replay_split)
    constexpr int XLAST = NL * U - 256 * (NPASS - 1);
#ifdef TQ_NO_REPLAY_SPLIT
    constexpr int G = 1;
#else
    constexpr int G = NPASS == 1 ? 1 : (XLAST <= 32 ? 8 : (XLAST <= 64 ? 4 : (XLAST <= 128 ? 2 : 1)));
#endif
    int64_t ej[NPASS];
    int es0[NPASS];
    float ep[NPASS], em[NPASS], ev[NPASS];
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const bool split = G > 1 && q == NPASS - 1;
      const int e = split ? 256 * q + tid / G : tid + 256 * q;
      const int64_t i = u0 + (e % U);
      ej[q] = -1;
      es0[q] = s1 + 1;
      ep[q] = em[q] = ev[q] = 0.0f;
      if (e < NL * U && i < B) {
        const int64_t u = tq_decode_unit(a, i).u;
        ej[q] = (int64_t)(e / U) * tq_num_units(a) + u;
        es0[q] = a.last_step[u] + 1;
        ep[q] = a.params[ej[q]];
        em[q] = a.exp_avg[ej[q]];
        ev[q] = a.exp_avg_sq[ej[q]];
      }
    }
    {
      double pw1 = tq_powi(a.beta1_d, T0 + tid), pw2 = tq_powi(a.beta2_d, T0 + tid);
      const double b1_256 = tq_powi(a.beta1_d, 256), b2_256 = tq_powi(a.beta2_d, 256);
      for (int e = tid; e < TQ_BIAS_TABLE_STEPS; e += 256) {  // same expressions as tq_adam_bias_entry
        if (T0 + e <= s1) {
          s_bias[2 * e] = a.lr * TQ_FRCP((float)(1.0 - pw1));
          s_bias[2 * e + 1] = TQ_FRCP(TQ_FSQRT((float)(1.0 - pw2)));
        }
        pw1 *= b1_256;
        pw2 *= b2_256;
      }
    }
    TQ_STAMP2(0)
    __syncthreads();
    TQ_STAMP2(1)
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      if (G > 1 && q == NPASS - 1) tq_adam_replay_split<G>(a, ej[q], es0[q], s1, s_bias, T0, ep[q], em[q], ev[q], tid % G);
      else if (ej[q] >= 0) tq_adam_replay_tab_given(a, ej[q], es0[q], s1, s_bias, T0, ep[q], em[q], ev[q]);
      if (q == 0) { TQ_STAMP2(2) }
    }
    TQ_STAMP2(3)
    __syncthreads();
  }
  TQ_STAMP(1)
  if constexpr (K <= 3 && (K + 1) * U <= 64) {
    // one KIND of site per wave -- wave 0 the K+1 Gamma sites (background, heights), waves 1..3 the width / x / y sites --
    // so that no wave runs the Gamma code and then the Beta code (with its regimes) for different lanes
    const int w = tid >> 6, l = tid & 63;
    const int nl = (w == 0 ? K + 1 : K) * U;
    const int site = (w == 0 ? 0 : K + 1 + (w - 1) * K) + (l / U);
    const int64_t i = u0 + (l % U);
    if (l < nl && i < B) tq_body_site(a, site, i);
  } else {
    for (int e = tid; e < NS * U; e += 256) {
      const int64_t i = u0 + (e % U);
      if (i < B) tq_body_site(a, e / U, i);
    }
  }
  __syncthreads();
  TQ_STAMP(2)
  // ---- wait for the gain of this step (bounded: ~2 s of the 100 MHz wall clock) ----
  __shared__ float s_gain;
  auto wait_flag = [&](int word) {
    if (tid == 0) {
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
      int ok = 1;
      while (__hip_atomic_load(&a.sync[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != flag_value) {
        __builtin_amdgcn_s_sleep(16);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
          ok = 0;
          break;
        }
      }
      // No acquire fence: at device scope it invalidates the CU's vector cache AND this XCD's L2 for every workgroup on
      // them.  What the tail workgroup publishes is read so that no stale copy can answer instead:
      //   the gain        from a sync word, with a device-scope load (here);
      //   TqGlobals       with plain (scalar) loads in the per-unit phase, after the second flag: no workgroup touches the
      //                   struct's cache lines earlier in the launch (the gain comes from the sync word for that reason), and
      //                   a launch starts with clean caches;
      //   per-AOI params  with device-scope loads in tq_body_unit (their first line also holds the end of the last
      //                   local-parameter row, which a replay may have read).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (ok && word == 1) s_gain = __int_as_float(__hip_atomic_load(&a.sync[TQ_SYNC_GAIN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
  };
  auto step_lost = [&]() {  // never observed: leave a visible trace (NaN loss) instead of reading half-written tables
    if (tid == 0) {
      // The step is lost.  Its row of partial sums carries a NaN ELBO, so the tail of this step (run by the next launch or
      // by tq_cosmos_tail) reports a NaN loss whichever workgroup was late, and Model.run rolls back to its last
      // checkpoint (model.py:220-232); CosmosEngine.reset_adam_clock zeroes the sync words on that path.
      a.blk_part[wblk * (TQ_ROWS_GCOL + tq_num_gsum(a)) + TQ_ROWS_GCOL + TQ_GS_ELBO] = __builtin_nanf("");
      a.elbo_out[0] = __builtin_nan("");
      __hip_atomic_fetch_add(&a.sync[TQ_SYNC_LOST], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (never reset: a soak run reads it)
      count_out();  // still counted: the ticket counter is re-armed for the launches that follow
    }
  };
  if (!wait_flag(1)) {
    step_lost();
    return;
  }
  TQ_STAMP(3)
  // ---- phase 2: likelihood of the U units (reads the draws of phase 1 and the gain): sixteen of them with 16 lanes each,
  // and of U = 20 the last four with a wave each (tq_mb_upr: why 20)
  tq_ksmogn_args kw = k;
  kw.gain = &s_gain;  // (the copy read at the flag)
  tq_ksmogn_tile_at<K, ONE, true, 16, 16, false>(kw, B, u0, u_end, smem);
  if constexpr (U > 16) {
    static_assert(U == 20, "16 units at 16 lanes + 4 at 64");
    tq_ksmogn_tile_at<K, ONE, true, 64, 16, true>(kw, B, u0 + 16, u_end, smem);
  }
  __syncthreads();
  // ---- the other global draws (tables of the per-unit terms): set long ago, unless the tail workgroup started late ----
  if (!wait_flag(TQ_SYNC_FLAG2)) {
    step_lost();
    return;
  }
  TQ_STAMP(4)
  // ---- phase 3: per-unit terms + Adam, one lane per unit; row of partial sums ----
  const int nq = tq_num_gsum(a), ncol = TQ_ROWS_GCOL + nq;
  float part[TQ_MAX_NGSUM], aoi[TQ_ROWS_GCOL];
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) part[j] = 0.0f;
#pragma unroll
  for (int j = 0; j < TQ_ROWS_GCOL; ++j) aoi[j] = 0.0f;
  // (units 16..19 of U = 20: the second lane of the first four 16-lane groups)
  const int64_t i = u0 + (tid >> 4) + 16 * (tid & 15);
  if ((tid & 15) < (U + 15) / 16 && i < u_end) {
    float aoi2[2];
    tq_body_unit<K, false, false, true>(a, i, part, aoi2);
    const uint32_t FC = (uint32_t)(a.fb * a.C);
    const int c = (int)((uint32_t)i % (uint32_t)a.C);
    const int slot = (uint32_t)i / FC == (uint32_t)u0 / FC ? 0 : 1;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
      for (int q = 0; q < TQ_MAXQ; ++q) {
        const bool mine = sl == slot && q == c;
        aoi[sl * TQ_ROWS_AOICOL + 2 * q] = mine ? aoi2[0] : 0.0f;
        aoi[sl * TQ_ROWS_AOICOL + 2 * q + 1] = mine ? aoi2[1] : 0.0f;
      }
    }
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < TQ_ROWS_GCOL; ++j) {
    if ((j % TQ_ROWS_AOICOL) < 2 * a.C) {
      const float sum = tq_wave_sum(aoi[j]);
      if (lane == 0) s_part[wave][j] = sum;
    }
  }
#pragma unroll
  for (int j = 0; j < TQ_MAX_NGSUM; ++j) {
    if (j < nq) {
      const float sum = tq_wave_sum(part[j]);
      if (lane == 0) s_part[wave][TQ_ROWS_GCOL + j] = sum;
    }
  }
  __syncthreads();
  if (tid < ncol) {
    const bool used = tid >= TQ_ROWS_GCOL || (tid % TQ_ROWS_AOICOL) < 2 * a.C;
    const float sum = used ? (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]) : 0.0f;
    a.blk_part[wblk * ncol + tid] = sum;
  }
  TQ_STAMP(5)
  // the last workgroup to get here re-arms the ticket counter for the next launch (the flag holds the step number)
  if (tid == 0) count_out();
}


// ---------------------------------------------------------------------------------------------------------
static int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[200];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    tq_set_error(buf);
    return TQ_ERR_LAUNCH;
  }
  return TQ_OK;
}

static int check_args(const tq_cosmos_args* a, const char* who) {
  if (!a || !a->params || !a->globals || !a->gbase) {
    tq_set_error("tq_cosmos_*: NULL args/params/globals/gbase");
    return TQ_ERR_ARG;
  }
  if (a->K < 1 || a->K > TQ_MAX_K || a->P < 2 || a->P > TQ_MAX_P || a->C < 1 || a->C > TQ_MAXQ || a->nb < 1 ||
      a->fb < 1 || a->nb > a->Nt || a->fb > a->F || a->O < 1) {
    tq_set_error("tq_cosmos_*: unsupported K/P/C or inconsistent batch geometry");
    return TQ_ERR_ARG;
  }
  if (tq_num_units(*a) >= ((int64_t)1 << 31)) {
    tq_set_error("tq_cosmos_*: Nt*F*C must be below 2^31 (unit indices are 32-bit on the device)");
    return TQ_ERR_ARG;
  }
  if (a->images_by_slot && (a->crosstalk || !a->ndx)) {
    tq_set_error("tq_cosmos_*: images_by_slot (a streamed window of AOIs) needs the AOI list ndx of the batch; cosmos model only");
    return TQ_ERR_ARG;
  }
  if (a->crosstalk && (a->C != 2 || a->K > 2)) {
    tq_set_error("tq_cosmos_*: the crosstalk model is implemented for Q = C = 2 and K <= 2");
    return TQ_ERR_ARG;
  }
  (void)who;
  return TQ_OK;
}

extern "C" int64_t tq_globals_size(void) { return (int64_t)sizeof(TqGlobals); }
extern "C" int64_t tq_gbase_size(void) { return (int64_t)sizeof(TqGlobalBase); }
extern "C" int64_t tq_cosmos_nblk(int64_t B) { return (B + TQ_UNIT_BLOCK - 1) / TQ_UNIT_BLOCK; }
extern "C" int64_t tq_cosmos_param_count(int32_t Nt, int32_t F, int32_t C, int32_t K) {
  return (int64_t)TQ_NLOCAL(K) * Nt * F * C + 2 * (int64_t)Nt * C + TQ_NGLOBAL(C);
}
extern "C" int64_t tq_crosstalk_param_count(int32_t Nt, int32_t F, int32_t C, int32_t K) {
  return (int64_t)TQ_NLOCAL(K) * Nt * F * C + 2 * (int64_t)Nt * C + TQ_NGLOBAL_X(C, 1);
}

extern "C" int tq_cosmos_sample_globals(const tq_cosmos_args* a, void* stream) {
  if (int rc = check_args(a, "sample_globals")) return rc;
  hipLaunchKernelGGL(tq_sample_globals_kernel, dim3(tq_num_gsites(*a)), dim3(64), 0, (hipStream_t)stream, *a);
  return check_launch("tq_sample_globals_kernel");
}

extern "C" int tq_cosmos_sample_locals(const tq_cosmos_args* a, void* stream) {
  if (int rc = check_args(a, "sample_locals")) return rc;
  if (!a->lat || !a->site) {
    tq_set_error("tq_cosmos_sample_locals: lat or site is NULL");
    return TQ_ERR_ARG;
  }
  const int64_t B = tq_batch_units(*a);
  hipLaunchKernelGGL(tq_sample_locals_kernel, dim3((unsigned)((B + 255) / 256), (unsigned)(1 + 4 * a->K)), dim3(256), 0,
                     (hipStream_t)stream, *a, B, 0);
  return check_launch("tq_sample_locals_kernel");
}

// argument block of the likelihood kernel of a cosmos step
static tq_ksmogn_args cosmos_ksmogn_args(const tq_cosmos_args* a) {
  const int K = a->K, M = 1 << K;
  const int64_t B = tq_batch_units(*a), U = tq_num_units(*a);
  tq_ksmogn_args k = {};
  k.images = a->images; k.images_il = a->images_il; k.xy = a->xy; k.ndx = a->ndx; k.fdx = a->fdx;
  k.nb_full = a->Nt;
  k.pixstats = a->pixstats;
  k.stats_stride = U;
  k.il_min_units = a->il_min_units;
  k.pixel_mode = a->pixel_mode;
  k.background = a->lat;
  k.height = a->lat + (int64_t)1 * B;
  k.width = a->lat + (int64_t)(1 + K) * B;
  k.x = a->lat + (int64_t)(1 + 2 * K) * B;
  k.y = a->lat + (int64_t)(1 + 3 * K) * B;
  k.gain = &((const TqGlobals*)a->globals)->gain;
  k.offset_samples = a->offset_samples; k.offset_logits = a->offset_logits;
  k.gout = nullptr;
  k.m_logit = a->params;
  k.m_kstride = U;
  k.aoi_mask = a->aoi_mask;
  k.ll = a->pix;
  k.g_background = a->pix + (int64_t)M * B;
  k.g_gain = a->pix + (int64_t)(M + 1) * B;
  k.g_height = a->pix + (int64_t)(M + 2) * B;
  k.g_width = a->pix + (int64_t)(M + 2 + K) * B;
  k.g_x = a->pix + (int64_t)(M + 2 + 2 * K) * B;
  k.g_y = a->pix + (int64_t)(M + 2 + 3 * K) * B;
  k.nb = a->nb; k.fb = a->fb; k.C = a->C; k.F = a->F; k.P = a->P; k.K = K; k.O = a->O;
  k.scale = a->scale;
  k.images_by_slot = a->images_by_slot;
  return k;
}

// pixel kernel of a step: fused render + log-likelihood + pathwise gradients, Dice weights from m_probs
static int launch_likelihood(const tq_cosmos_args* a, void* stream) {
  const int K = a->K, M = 1 << K;
  const int64_t B = tq_batch_units(*a), U = tq_num_units(*a);
  tq_ksmogn_args k = cosmos_ksmogn_args(a);
  if (a->crosstalk) {
    // one data site per AOI-frame, all dyes in every channel: per-dye marginal likelihoods go where the cosmos
    // per-unit routine expects ll, two more row groups follow the cosmos block of pix
    tq_xtalk_args x = {};
    x.images = a->images; x.xy = a->xy; x.ndx = a->ndx; x.fdx = a->fdx;
    x.images_il = a->images_il;  // crosstalk: interleaved (C, P, P) tiles per AOI-frame (tq_images_interleave_n)
    x.pixstats = a->pixstats; x.nb_full = a->Nt; x.il_min_units = a->il_min_units;
    x.background = k.background; x.height = k.height; x.width = k.width; x.x = k.x; x.y = k.y;
    x.gain = k.gain;
    x.alpha = &((const TqGlobals*)a->globals)->alpha[0][0];
    x.offset_samples = a->offset_samples; x.offset_logits = a->offset_logits;
    x.gout = nullptr; x.m_logit = a->params; x.m_kstride = U; x.aoi_mask = a->aoi_mask;
    x.ll_joint = nullptr; x.ll = a->pix;
    x.ell_excess = a->pix + (int64_t)(M + 2 + 4 * K) * B;
    x.g_alpha = a->pix + (int64_t)(M + 3 + 4 * K) * B;
    x.g_background = k.g_background; x.g_gain = k.g_gain;
    x.g_height = k.g_height; x.g_width = k.g_width; x.g_x = k.g_x; x.g_y = k.g_y;
    x.nb = a->nb; x.fb = a->fb; x.C = a->C; x.F = a->F; x.P = a->P; x.K = K; x.O = a->O;
    x.scale = a->scale;
    return tq_ksmogn_crosstalk_log_prob(&x, stream);
  }
  return tq_ksmogn_log_prob(&k, stream);
}

static int launch_pixel_unit(const tq_cosmos_args* a, void* stream) {
  const int64_t B = tq_batch_units(*a);

  const tq_ksmogn_args k = cosmos_ksmogn_args(a);
  const dim3 grid((unsigned)((B + 63) / 64)), block(64);
  hipStream_t st = (hipStream_t)stream;
  if (a->K == 1) {
    if (a->P == 14) hipLaunchKernelGGL((tq_pixel_unit_kernel<1, 14>), grid, block, 0, st, k, *a, B);
    else hipLaunchKernelGGL((tq_pixel_unit_kernel<1, 20>), grid, block, 0, st, k, *a, B);
  } else {
    if (a->P == 14) hipLaunchKernelGGL((tq_pixel_unit_kernel<2, 14>), grid, block, 0, st, k, *a, B);
    else hipLaunchKernelGGL((tq_pixel_unit_kernel<2, 20>), grid, block, 0, st, k, *a, B);
  }
  return check_launch("tq_pixel_unit_kernel");
}

// per-AOI sites + gsum of a step with rows, for callers that all-reduce gsum before the global sites (tq_rows_sums_kernel)
static int launch_rows_sums(const tq_cosmos_args* a, hipStream_t st) {
  if (!a->sync || !a->aoi_part || !a->gsum) {
    tq_set_error("tq_cosmos_elbo_grads: the rows layout needs sync, aoi_part and gsum");
    return TQ_ERR_ARG;
  }
  static const bool groups = [] {
    const char* e = getenv("TAPQIR_AMD_GROUPS");
    return !(e && e[0] == '0');
  }();
  if (groups) {
    hipLaunchKernelGGL(tq_group_sums_kernel, dim3((unsigned)tq_grp_count(tq_batch_units(*a))), dim3(256), 0, st, *a);
    return check_launch("tq_group_sums_kernel");
  }
  hipLaunchKernelGGL(tq_rows_sums_kernel, dim3((unsigned)((a->nb * a->C + 3) / 4)), dim3(256), 0, st, *a);
  return check_launch("tq_rows_sums_kernel");
}

// rows: AOI-aligned per-unit kernel whose tail also finishes the per-AOI sites (tq_unit_rows_kernel; full-batch steps
// that finish with tq_cosmos_tail or inside the next tq_cosmos_step_overlapped)
static int elbo_grads_impl(const tq_cosmos_args* a, void* stream, bool finish_sums, bool rows = false) {
  if (int rc = check_args(a, "elbo_grads")) return rc;
  if (!a->images || !a->xy || !a->is_ontarget || !a->offset_samples || !a->offset_logits || !a->grad || !a->lat ||
      !a->site || !a->pix || (!rows && !a->aoi_part) || !a->blk_part || !a->gsum) {
    tq_set_error("tq_cosmos_elbo_grads: NULL required pointer");
    return TQ_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const int K = a->K;
  const int64_t B = tq_batch_units(*a);
  if (a->pixel_mode == TQ_PIXEL_FUSED_UNIT) {  // pixel + per-unit kernel in one launch (rows of 64 units)
    if (!rows || !tq_fused_pixel_unit(*a)) {
      tq_set_error("tq_cosmos: pixel_mode = TQ_PIXEL_FUSED_UNIT needs a full-batch cosmos step with fused Adam, K <= 2, one "
                   "offset value, P in {14, 20}, the interleaved images and the pixel statistics");
      return TQ_ERR_ARG;
    }
    if (int rc = launch_pixel_unit(a, stream)) return rc;
    return finish_sums ? launch_rows_sums(a, st) : TQ_OK;
  }
  // 1. pixel kernel
  if (int rc = launch_likelihood(a, stream)) return rc;
  // 2. per-unit sites
  if (rows) {
    const dim3 grid((unsigned)tq_cosmos_nblk(B)), block(TQ_UNIT_BLOCK);
    switch (K) {
      case 1: hipLaunchKernelGGL((tq_unit_rows_kernel<1>), grid, block, 0, st, *a, B); break;
      case 2: hipLaunchKernelGGL((tq_unit_rows_kernel<2>), grid, block, 0, st, *a, B); break;
      case 3: hipLaunchKernelGGL((tq_unit_rows_kernel<3>), grid, block, 0, st, *a, B); break;
      default: hipLaunchKernelGGL((tq_unit_rows_kernel<4>), grid, block, 0, st, *a, B); break;
    }
    if (int rc = check_launch("tq_unit_rows_kernel")) return rc;
    return finish_sums ? launch_rows_sums(a, st) : TQ_OK;
  }
  const int64_t nblk = tq_cosmos_nblk(B);
  const dim3 grid((unsigned)nblk), block(TQ_UNIT_BLOCK);
  switch (K) {
    case 1: hipLaunchKernelGGL((tq_unit_kernel<1>), grid, block, 0, st, *a, B); break;
    case 2: hipLaunchKernelGGL((tq_unit_kernel<2>), grid, block, 0, st, *a, B); break;
    case 3: hipLaunchKernelGGL((tq_unit_kernel<3>), grid, block, 0, st, *a, B); break;
    default: hipLaunchKernelGGL((tq_unit_kernel<4>), grid, block, 0, st, *a, B); break;
  }
  if (int rc = check_launch("tq_unit_kernel")) return rc;
  // 3. per-AOI sites
  hipLaunchKernelGGL(tq_aoi_kernel, dim3((unsigned)(a->nb * a->C)), dim3(256), 0, st, *a, B);
  if (int rc = check_launch("tq_aoi_kernel")) return rc;
  // 4. cross-unit sums
  if (!finish_sums) return TQ_OK;
  hipLaunchKernelGGL(tq_reduce_kernel, dim3(1), dim3(256), 0, st, *a, nblk, B);
  return check_launch("tq_reduce_kernel");
}

extern "C" int tq_cosmos_elbo_grads(const tq_cosmos_args* a, void* stream) {
  // full-batch steps with the Adam of the local parameters fused in take the rows layout here too (the per-AOI sites are
  // finished by tq_rows_sums_kernel, which also leaves gsum ready for the caller's all-reduce)
  return elbo_grads_impl(a, stream, true, a && tq_rows_layout(*a) && a->sync && a->aoi_part);
}

extern "C" int tq_cosmos_pixel_unit(const tq_cosmos_args* a, void* stream) {
  if (int rc = check_args(a, "pixel_unit")) return rc;
  if (a->pixel_mode != TQ_PIXEL_FUSED_UNIT) {
    tq_set_error("tq_cosmos_pixel_unit: pixel_mode must be TQ_PIXEL_FUSED_UNIT");
    return TQ_ERR_ARG;
  }
  return elbo_grads_impl(a, stream, false, true);
}

extern "C" int tq_cosmos_globals_grad(const tq_cosmos_args* a, void* stream) {
  if (int rc = check_args(a, "globals_grad")) return rc;
  if (!a->grad || !a->gsum || !a->elbo_out) {
    tq_set_error("tq_cosmos_globals_grad: NULL required pointer");
    return TQ_ERR_ARG;
  }
  // per-site ELBO parts go through the tail of the gsum buffer (gsum has 3+3Q used entries; the
  // caller allocates TQ_GSUM_LEN doubles)
  double* site_elbo = a->gsum + tq_num_gsum(*a);
  hipLaunchKernelGGL(tq_globals_grad_kernel, dim3(tq_num_gsites(*a)), dim3(64), 0, (hipStream_t)stream, *a, site_elbo);
  if (int rc = check_launch("tq_globals_grad_kernel")) return rc;
  hipLaunchKernelGGL(tq_elbo_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *a, (const double*)site_elbo);
  return check_launch("tq_elbo_finish_kernel");
}

extern "C" int tq_cosmos_adam(const tq_cosmos_args* a, void* stream) {
  if (int rc = check_args(a, "adam")) return rc;
  if (!a->grad || !a->exp_avg || !a->exp_avg_sq) {
    tq_set_error("tq_cosmos_adam: NULL required pointer");
    return TQ_ERR_ARG;
  }
  const int64_t total = tq_num_params(*a);
  // with fuse_adam the local block was already updated by the per-unit kernel
  const int64_t first = a->fuse_adam ? tq_aoi_base(*a) : 0;
  int64_t nblk = (total - first + 255) / 256;
  if (nblk > 256 * 16) nblk = 256 * 16;  // grid-stride: 16 workgroups per CU
  hipLaunchKernelGGL(tq_adam_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, *a, first, total);
  return check_launch("tq_adam_kernel");
}

extern "C" int tq_cosmos_adam_catchup(const tq_cosmos_args* a, int32_t all_units, void* stream) {
  if (int rc = check_args(a, "adam_catchup")) return rc;
  if (!a->last_step || !a->exp_avg || !a->exp_avg_sq) {
    tq_set_error("tq_cosmos_adam_catchup: NULL required pointer (last_step, exp_avg, exp_avg_sq)");
    return TQ_ERR_ARG;
  }
  const int64_t n = all_units ? tq_num_units(*a) : tq_batch_units(*a);
  const dim3 grid((unsigned)((n + 255) / 256), (unsigned)TQ_NLOCAL(a->K));
  hipLaunchKernelGGL(tq_adam_catchup_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a, n, (int)all_units);
  return check_launch("tq_adam_catchup_kernel");
}

// ---- whole steps ---------------------------------------------------------------------------------------------
static int launch_reduce_globals(const tq_cosmos_args* a, hipStream_t st) {
  if (!a->grad || !a->gsum || !a->elbo_out) {
    tq_set_error("tq_cosmos_step: NULL required pointer");
    return TQ_ERR_ARG;
  }
  // no all-reduce on this path: sums, (per-AOI sites,) global sites and the total ELBO finish in one launch
  if (a->tail_kind == TQ_TAIL_ROWS16 || tq_rows_layout(*a)) {
    hipLaunchKernelGGL(tq_rows_reduce_globals_kernel, dim3(1), dim3(256), 0, st, *a, a->tail_kind == TQ_TAIL_ROWS16 ? tq_mb_upr(*a) : TQ_UNIT_BLOCK);
    return check_launch("tq_rows_reduce_globals_kernel");
  }
  const int64_t B = tq_batch_units(*a);
  hipLaunchKernelGGL(tq_reduce_globals_kernel, dim3(1), dim3(256), 0, st, *a, tq_cosmos_nblk(B), B);
  return check_launch("tq_reduce_globals_kernel");
}

extern "C" int tq_cosmos_tail(const tq_cosmos_args* a, void* stream) {
  if (int rc = check_args(a, "tail")) return rc;
  if (a->tail_kind != TQ_TAIL_ROWS16 && tq_rows_layout(*a) && a->fuse_adam && a->sync && a->aoi_part) {
    // rows of 64 or 256 units: the per-AOI frame sums span up to F C / 64 rows each -- one wave per AOI
    // (tq_rows_sums_kernel) instead of one workgroup walking all of them, then the global sites + tail Adam
    if (int rc = launch_rows_sums(a, (hipStream_t)stream)) return rc;
    return tq_cosmos_tail_reduced(a, nullptr, stream);
  }
  if (int rc = launch_reduce_globals(a, (hipStream_t)stream)) return rc;
  return tq_cosmos_adam(a, stream);
}

extern "C" int tq_cosmos_tail_reduced(const tq_cosmos_args* a, const tq_cosmos_args* next, void* stream) {
  if (int rc = check_args(a, "tail_reduced")) return rc;
  if (next)
    if (int rc = check_args(next, "tail_reduced (next)")) return rc;
  if (!a->grad || !a->gsum || !a->elbo_out || !a->exp_avg || !a->exp_avg_sq) {
    tq_set_error("tq_cosmos_tail_reduced: NULL required pointer");
    return TQ_ERR_ARG;
  }
  if (next && !a->fuse_adam) {
    tq_set_error("tq_cosmos_tail_reduced: the next step's global draws need this step's Adam to be complete (fuse_adam steps only)");
    return TQ_ERR_ARG;
  }
  hipLaunchKernelGGL(tq_tail_reduced_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *a, next ? *next : *a, next ? 1 : 0);
  if (int rc = check_launch("tq_tail_reduced_kernel")) return rc;
  return a->fuse_adam ? TQ_OK : tq_cosmos_adam(a, stream);
}

extern "C" int tq_cosmos_step(const tq_cosmos_args* a, void* stream) {
  if (int rc = check_args(a, "step")) return rc;
  if (int rc = tq_cosmos_sample_globals(a, stream)) return rc;
  if (int rc = tq_cosmos_sample_locals(a, stream)) return rc;
  if (int rc = elbo_grads_impl(a, stream, false, tq_rows_layout(*a))) return rc;
  return tq_cosmos_tail(a, stream);
}

extern "C" int tq_cosmos_step_overlapped(const tq_cosmos_args* a, const tq_cosmos_args* prev, void* stream) {
  if (int rc = check_args(a, "step_overlapped")) return rc;
  if (prev)
    if (int rc = check_args(prev, "step_overlapped (prev)")) return rc;
  if (!a->fuse_adam || (prev && !prev->fuse_adam)) {
    tq_set_error("tq_cosmos_step_overlapped: full-batch steps with fuse_adam only (the next step's local sampling must not depend on the pending tail)");
    return TQ_ERR_ARG;
  }
  if (!a->lat || !a->site || !a->grad || !a->gsum || !a->elbo_out || !a->exp_avg || !a->exp_avg_sq ||
      (prev && (!prev->grad || !prev->gsum || !prev->elbo_out || !prev->exp_avg || !prev->exp_avg_sq || !prev->blk_part ||
                (!tq_rows_layout(*prev) && !prev->aoi_part)))) {
    tq_set_error("tq_cosmos_step_overlapped: NULL required pointer");
    return TQ_ERR_ARG;
  }
  const int64_t B = tq_batch_units(*a);
  if (prev && prev->tail_kind == TQ_TAIL_ROWS16) {
    // (a pending single-launch minibatch step: its rows-of-16 tail is not carried by the sampling launch, whose register
    // allocation every extra tail variant burdens)
    if (int rc = tq_cosmos_tail(prev, stream)) return rc;
    prev = nullptr;
  }
  int code = prev ? tq_prev_code(*prev) : 0;
  {
    // rows of 64 / 256 units: the idle workgroups of the launch's first grid row add them per group (tq_group_reduce_rows)
    static const bool groups = [] {
      const char* e = getenv("TAPQIR_AMD_GROUPS");
      return !(e && e[0] == '0');
    }();
    const int64_t gx = (B + 255) / 256;
    if (groups && code == 3 && prev->sync && tq_grp_count(tq_batch_units(*prev)) + 1 <= gx) code = 6;
  }
  hipLaunchKernelGGL(tq_sample_locals_tail_kernel, dim3((unsigned)((B + 255) / 256), (unsigned)(2 + 4 * a->K)), dim3(256), 0,
                     (hipStream_t)stream, *a, prev ? *prev : *a, code, B, 0);
  if (int rc = check_launch("tq_sample_locals_tail_kernel")) return rc;
  return elbo_grads_impl(a, stream, false, tq_rows_layout(*a));
}


extern "C" int64_t tq_cosmos_blk_floats(int32_t Nt, int32_t F, int32_t C, int32_t crosstalk, int64_t B) {
  const int64_t ncol = TQ_ROWS_GCOL + TQ_NGSUM_X(C, crosstalk);
  const int64_t full = (((int64_t)Nt * F * C + 63) / 64) * ncol;                             // full-batch rows of 256 or 64 units
  const int64_t mini = ((B + TQ_UNITS_PER_BLOCK - 1) / TQ_UNITS_PER_BLOCK) * ncol;          // single-launch minibatch step: rows of 16
  const int64_t flat = tq_cosmos_nblk(B) * TQ_NGSUM_X(C, crosstalk);
  const int64_t grp = 4 + tq_grp_count((int64_t)Nt * F * C) * TQ_GGROW;                      // group rows behind the full-batch rows
  const int64_t most = full > mini ? (full > flat ? full : flat) : (mini > flat ? mini : flat);
  return most + grp;
}


extern "C" int tq_cosmos_minibatch_step(const tq_cosmos_args* a, const tq_cosmos_args* prev, void* stream) {
  if (int rc = check_args(a, "minibatch_step")) return rc;
  if (prev)
    if (int rc = check_args(prev, "minibatch_step (prev)")) return rc;
  if (a->crosstalk || !a->fuse_adam || (prev && !prev->fuse_adam) || (int64_t)a->fb * a->C < TQ_UNITS_PER_BLOCK) {
    tq_set_error("tq_cosmos_minibatch_step: cosmos steps with fuse_adam and at least 16 units per AOI only");
    return TQ_ERR_ARG;
  }
  if (!a->images || !a->xy || !a->is_ontarget || !a->offset_samples || !a->offset_logits || !a->lat || !a->site || !a->pix ||
      !a->blk_part || !a->gsum || !a->elbo_out || !a->exp_avg || !a->exp_avg_sq || !a->grad || !a->sync ||
      (prev && (!prev->grad || !prev->gsum || !prev->elbo_out || !prev->exp_avg || !prev->exp_avg_sq || !prev->blk_part ||
                (tq_prev_code(*prev) == 1 && !prev->aoi_part)))) {
    tq_set_error("tq_cosmos_minibatch_step: NULL required pointer");
    return TQ_ERR_ARG;
  }
  const int64_t B = tq_batch_units(*a);
  const tq_ksmogn_args k = cosmos_ksmogn_args(a);
  const bool one = a->O == 1 && a->pixstats;
  // one workgroup per 16 (or 20: tq_mb_upr) units + the one that runs the tail and the global draws
  const int upr = tq_mb_upr(*a);
  const dim3 grid((unsigned)((B + upr - 1) / upr) + 1), block(256);
  const char* tl = getenv("TAPQIR_AMD_MB_TAIL_LAST");
  const int tail_last = !(tl && tl[0] == '0') && grid.x > 1 && (grid.x - 1) % 256 == 0 && grid.x <= 513;  // (<= 2 workgroups per CU: all resident)
  size_t lds = sizeof(float) * tq_tile16_lds_floats(a->P, a->K, a->O);
  if (a->next_ndx || a->next_fdx) {
    if ((a->next_ndx && a->Nt > TQ_SUBSAMPLE_MAX) || (a->next_fdx && a->F > TQ_SUBSAMPLE_MAX)) {
      tq_set_error("tq_cosmos_minibatch_step: next_ndx / next_fdx need Nt, F <= TQ_SUBSAMPLE_MAX");
      return TQ_ERR_ARG;
    }
    const size_t sel = sizeof(int) * (2048 + 8);  // tq_draw_subsample: histogram of 2048 bins + scan / boundary words
    if (lds < sel) lds = sel;
  }
  const int code = prev ? tq_prev_code(*prev) : 0;
  const tq_cosmos_args& pv = prev ? *prev : *a;
  hipStream_t st = (hipStream_t)stream;
#define TQ_MB_LAUNCH(KK)                                                                                                 \
  if (upr == 20) {                                                                                                       \
    if (one) hipLaunchKernelGGL((tq_minibatch_kernel<KK, true, 20>), grid, block, lds, st, *a, pv, code, k, B, tail_last);         \
    else hipLaunchKernelGGL((tq_minibatch_kernel<KK, false, 20>), grid, block, lds, st, *a, pv, code, k, B, tail_last);            \
  } else {                                                                                                               \
    if (one) hipLaunchKernelGGL((tq_minibatch_kernel<KK, true, 16>), grid, block, lds, st, *a, pv, code, k, B, tail_last);         \
    else hipLaunchKernelGGL((tq_minibatch_kernel<KK, false, 16>), grid, block, lds, st, *a, pv, code, k, B, tail_last);            \
  }
  switch (a->K) {
    case 1: TQ_MB_LAUNCH(1) break;
    case 2: TQ_MB_LAUNCH(2) break;
    case 3: TQ_MB_LAUNCH(3) break;
    default: TQ_MB_LAUNCH(4) break;
  }
#undef TQ_MB_LAUNCH
  return check_launch("tq_minibatch_kernel");
}

// AOI-sharded pipeline: the local sites [site_begin, site_begin + site_count) of `a`; with `prev` (whose gsum the caller
// has all-reduced) the launch also carries, as one extra workgroup, everything of `prev` after the all-reduce and the
// global draws of `a` (tq_cosmos_tail_reduced(prev, a)).  A sharded host samples the first sites of step t+1 while the
// all-reduce of step t is in flight, waits for it, and calls this with `prev` for the remaining sites.
extern "C" int tq_cosmos_sample_locals_range(const tq_cosmos_args* a, int32_t site_begin, int32_t site_count,
                                             const tq_cosmos_args* prev, void* stream) {
  if (int rc = check_args(a, "sample_locals_range")) return rc;
  if (prev)
    if (int rc = check_args(prev, "sample_locals_range (prev)")) return rc;
  if (!a->lat || !a->site || site_begin < 0 || site_count < 1 || site_begin + site_count > 1 + 4 * a->K) {
    tq_set_error("tq_cosmos_sample_locals_range: bad site range or NULL lat/site");
    return TQ_ERR_ARG;
  }
  if (prev && (!prev->fuse_adam || !prev->grad || !prev->gsum || !prev->elbo_out || !prev->exp_avg || !prev->exp_avg_sq)) {
    tq_set_error("tq_cosmos_sample_locals_range: prev must be a full-batch (fuse_adam) step with grad/gsum/elbo_out/moments");
    return TQ_ERR_ARG;
  }
  const int64_t B = tq_batch_units(*a);
  const unsigned gx = (unsigned)((B + 255) / 256);
  if (prev) {
    hipLaunchKernelGGL(tq_sample_locals_tail_kernel, dim3(gx, (unsigned)(site_count + 1)), dim3(256), 0, (hipStream_t)stream, *a,
                       *prev, 2, B, (int)site_begin);
    return check_launch("tq_sample_locals_tail_kernel");
  }
  hipLaunchKernelGGL(tq_sample_locals_kernel, dim3(gx, (unsigned)site_count), dim3(256), 0, (hipStream_t)stream, *a, B,
                     (int)site_begin);
  return check_launch("tq_sample_locals_kernel");
}

// ---- posterior read-out (cosmos.compute_probs) -------------------------------------------------------------
__global__ __launch_bounds__(64) void tq_probs_globals_kernel(const tq_probs_args a) {
  if (threadIdx.x == 0) tq_body_probs_globals(a, blockIdx.x, blockIdx.y);
}

template <int K>
__global__ __launch_bounds__(256) void tq_probs_kernel(const tq_probs_args a, const int64_t U) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u < U) tq_body_probs_unit<K>(a, u);
}

extern "C" int tq_cosmos_probs(const tq_probs_args* a, void* stream) {
  if (!a || !a->params || !a->is_ontarget || !a->globals_p || !a->gbase_p || !a->z_probs || !a->theta_probs ||
      (!a->draw && !a->xy_given)) {
    tq_set_error("tq_cosmos_probs: NULL required pointer");
    return TQ_ERR_ARG;
  }
  if (a->K < 1 || a->K > TQ_MAX_K || a->C < 1 || a->C > TQ_MAXQ || a->particles < 1 || a->particles > 65535) {
    tq_set_error("tq_cosmos_probs: unsupported K/C/particles");
    return TQ_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(tq_probs_globals_kernel, dim3(TQ_NGSITES(a->C), a->particles), dim3(64), 0, st, *a);
  if (int rc = check_launch("tq_probs_globals_kernel")) return rc;
  const int64_t U = (int64_t)a->Nt * a->F * a->C;
  const dim3 grid((unsigned)((U + 255) / 256)), block(256);
  switch (a->K) {
    case 1: hipLaunchKernelGGL((tq_probs_kernel<1>), grid, block, 0, st, *a, U); break;
    case 2: hipLaunchKernelGGL((tq_probs_kernel<2>), grid, block, 0, st, *a, U); break;
    case 3: hipLaunchKernelGGL((tq_probs_kernel<3>), grid, block, 0, st, *a, U); break;
    default: hipLaunchKernelGGL((tq_probs_kernel<4>), grid, block, 0, st, *a, U); break;
  }
  return check_launch("tq_probs_kernel");
}
