/* tapqir_hip.h -- C ABI of libtapqir_hip.so, the MI355X (gfx950) implementation of the
 * cosmos SVI hot path of Tapqir.
 *
 * The reference has no FFI for this path (it is 100 % Python): the seam is
 * the Python class contract tapqir.models.Model / tapqir.distributions.KSMOGN.  Each entry
 * point below names the reference code it replaces; the Python host (tapqir_amd/) binds them
 * through ctypes (INTEGRATION.md shows the stub a maintainer would add to the reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller, contiguous, row-major, float32
 *     unless stated; nothing is allocated, freed or synchronised inside the library
 *   - every call is asynchronous on the caller's `stream` (a hipStream_t passed as void*)
 *   - return value: 0 = launched, nonzero = TQ_ERR_*; tq_last_error() gives the text
 *   - "unit" = one (AOI n, frame f, channel c) image of P x P pixels.  Minibatch units are
 *     numbered i = (a * fb + b) * C + c with n = ndx[a], f = fdx[b] (ndx/fdx may be NULL =
 *     identity); B = nb * fb * C.  Dataset-sized arrays use u = (n * F + f) * C + c.
 *   - spot arrays are K-major: [K][B] (minibatch) or [K][Nt*F*C] (dataset), as the reference's
 *     (K, Nt, F, Q) parameters (tapqir/models/cosmos.py:481-598)
 *   - enumerated spot-presence combinations are indexed mi in [0, 2^K), bit k of mi = m_k
 */
#ifndef TAPQIR_HIP_H
#define TAPQIR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TQ_OK 0
#define TQ_ERR_ARG 1     /* invalid argument (shape, NULL pointer, unsupported K/P) */
#define TQ_ERR_LAUNCH 2  /* hipLaunch failed (text in tq_last_error) */

#define TQ_MAX_K 4
#define TQ_MAX_P 32
#define TQ_GSUM_LEN 32   /* doubles in tq_cosmos_args.gsum */

int tq_version(void);
const char* tq_last_error(void);

/* ---------------------------------------------------------------------------------------
 * KSMOGN image likelihood, forward and (optionally) backward in one pass.
 * Replaces: tapqir/distributions/ksmogn.py:146-238 (gaussians -> image -> concentration ->
 * log_prob; the KeOps Genred LogSumExp of 188-216) + tapqir/distributions/util.py:15-64
 * (gaussian_spots) + their autograd backward.
 *
 *   ll[mi][i] = sum_pixels log sum_o w_o Gamma(D - delta_o; (b + sum_{k in mi} spot_k)/g, 1/g)
 *
 * If the g_* outputs are non-NULL the kernel also returns the gradient of
 *   sum_mi gout[mi][i] * ll[mi][i]
 * with respect to background, height, width, x, y (per unit) and gain (per-unit partials,
 * to be summed by the caller).  The upstream weights are either given (`gout`) or built in
 * the kernel as the Dice weights of the cosmos guide:
 *   gout[mi][i] = scale * mask[n] * prod_k (m_k ? sigmoid(u_k) : sigmoid(-u_k)),
 *   u_k = m_logit[k * m_kstride + u]   (unconstrained `m_probs`, cosmos.py:419-424).
 * ------------------------------------------------------------------------------------- */
typedef struct {
  const float* images;         /* (Nt, F, C, P, P) */
  const float* images_il;      /* the same images in the 64-unit interleaved layout of tq_images_interleave, or
                                  NULL.  Used for contiguous batches (ndx == fdx == NULL, nb == nb_full, fb == F) */
  const float* pixstats;       /* [3][stats_stride] per-unit data statistics of tq_image_stats (dataset-indexed), or
                                  NULL.  With O == 1 they enable the single-offset formulation (tq_pixel.h) */
  const float* xy;             /* (Nt, F, C, 2) target locations (x, y) */
  const int32_t* ndx;          /* [nb] AOI indices or NULL */
  const int32_t* fdx;          /* [fb] frame indices or NULL */
  const float* background;     /* [B]    sampled background b */
  const float* height;         /* [K][B] sampled h */
  const float* width;          /* [K][B] sampled w */
  const float* x;              /* [K][B] sampled x */
  const float* y;              /* [K][B] sampled y */
  const float* gain;           /* [1]    sampled gain g (device scalar) */
  const float* offset_samples; /* [O] */
  const float* offset_logits;  /* [O] log weights */
  const float* gout;           /* [2^K][B] upstream weights, or NULL */
  const float* m_logit;        /* (K, Nt, F, C) unconstrained m_probs, used when gout == NULL */
  const uint8_t* aoi_mask;     /* [Nt] or NULL (all ones) */
  float* ll;                   /* [2^K][B] out */
  float* g_background;         /* [B]    out or NULL (forward only) */
  float* g_height;             /* [K][B] out */
  float* g_width;              /* [K][B] out */
  float* g_x;                  /* [K][B] out */
  float* g_y;                  /* [K][B] out */
  float* g_gain;               /* [B]    out: per-unit partial of d/d gain */
  int64_t m_kstride;           /* = Nt*F*C */
  int64_t stats_stride;        /* = Nt*F*C (row stride of pixstats) */
  int32_t nb, fb, C, F;        /* minibatch and dataset geometry */
  int32_t P, K, O;
  int32_t nb_full;             /* Nt of the dataset behind images_il */
  int32_t il_min_units;        /* use the interleaved kernel only for batches of at least this many units */
  float scale;                 /* plate scale (Nt/nb)(F/fb), used with m_logit */
  int32_t pixel_mode;          /* backward pass on the interleaved layout (K <= 2, one offset): 1 = persistent waves that walk
                                  over the tiles with every global read an LDS-DMA request, 0 = one wave per tile.  Same
                                  results; which is faster depends on the box, so the host times both once */
  int32_t images_by_slot;      /* 1: `images` holds only the AOIs of THIS batch, AOI number ai of the batch at
                                  images[ai * F * C * P * P] (a window of a data set that does not fit the device: the host
                                  streams the AOIs of a batch in, dataset.py:140-151 does the same per minibatch); every other
                                  array stays dataset-indexed.  Gathered batches (ndx given) of the 16-lane kernel */
} tq_ksmogn_args;

int tq_ksmogn_log_prob(const tq_ksmogn_args* a, void* stream);

/* ---------------------------------------------------------------------------------------
 * KSMOGN likelihood of the crosstalk model (Q = C = 2 dyes / channels).
 * Replaces: the `alpha is not None` branch of tapqir/distributions/ksmogn.py:93-105, 146-165 as used by
 * tapqir/models/crosstalk.py:262-281: one observation per AOI-frame with event shape (C, P, P),
 *   image_c = b_c + sum_q alpha_qc sum_k m_qk h_qk N(x_qk + tx_c, y_qk + ty_c; w_qk),
 * for every joint spot-presence combination (bit q*K + k of the combination index = m_qk).
 * Units follow the cosmos layout: unit g*C + c of AOI-frame g carries background b_c and the spots of dye
 * q = c.  Outputs: the joint log-likelihoods (`ll_joint`), and/or, with `m_logit`, the per-dye marginals
 *   ll[mq][g*C + q] = sum_{m_-q} q(m_-q) ll_joint(m_q, m_-q)
 * that the cosmos per-unit routine consumes, plus `ell_excess` (the part of the ELBO those marginals count
 * more than once).  Backward as in tq_ksmogn_log_prob, with the extra output g_alpha.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  const float* images;         /* (Nt, F, C, P, P) */
  const float* images_il;      /* tq_images_interleave_n(images, ., Nt*F, C*P*P): one (C, P, P) tile per AOI-frame, or NULL.
                                  With pixstats, O == 1 and a contiguous batch of at least il_min_units AOI-frames the
                                  packed lane-per-AOI-frame kernel runs */
  const float* pixstats;       /* [3][Nt*F*C] tq_image_stats of the (n, f, c) tiles, or NULL */
  const float* xy;             /* (Nt, F, C, 2) */
  const int32_t* ndx;          /* [nb] or NULL */
  const int32_t* fdx;          /* [fb] or NULL */
  const float* background;     /* [B]    B = nb*fb*C, unit g*C + c */
  const float* height;         /* [K][B] unit g*C + q */
  const float* width;
  const float* x;
  const float* y;
  const float* gain;           /* [1] */
  const float* alpha;          /* [Q*C] crosstalk fractions alpha[q][c] */
  const float* offset_samples; /* [O] */
  const float* offset_logits;  /* [O] */
  const float* gout;           /* [2^(QK)][nb*fb] upstream weights of the joint combinations, or NULL */
  const float* m_logit;        /* (K, Nt, F, Q) unconstrained m_probs: Dice weights and per-dye marginals */
  const uint8_t* aoi_mask;     /* [Nt] or NULL */
  float* ll_joint;             /* [2^(QK)][nb*fb] out or NULL */
  float* ll;                   /* [2^K][B] out or NULL (needs m_logit) */
  float* ell_excess;           /* [B] out or NULL */
  float* g_background;         /* [B] out or NULL (forward only) */
  float* g_height;             /* [K][B] */
  float* g_width;
  float* g_x;
  float* g_y;
  float* g_gain;               /* [B] per-unit partial (whole AOI-frame in unit c = 0) */
  float* g_alpha;              /* [Q][B]: row q, unit g*C + c = d/d alpha[q][c] */
  int64_t m_kstride;           /* = Nt*F*Q */
  int32_t nb, fb, C, F;
  int32_t P, K, O;
  int32_t nb_full;             /* Nt of the dataset behind images_il */
  int32_t il_min_units;        /* AOI-frames from which the packed kernel is used */
  float scale;
} tq_xtalk_args;

int tq_ksmogn_crosstalk_log_prob(const tq_xtalk_args* a, void* stream);

/* Tile-interleaved copy of the image tensor for the contiguous-batch kernel: units in blocks of 64,
 * pixels in groups of 4, so that a wave64 reads one contiguous 1 KiB row per load:
 *   out[((u / 64) * npix4 + q) * 64 + (u % 64)] = float4{ pixels 4q..4q+3 of unit u },  npix4 = ceil(P*P/4)
 * `out` must hold tq_interleaved_floats(U, P) floats. */
int64_t tq_interleaved_floats(int64_t U, int32_t P);
int tq_images_interleave(const float* images, float* images_il, int64_t U, int32_t P, void* stream);
/* the same for tiles of `npix` floats (crosstalk: one (C, P, P) tile per AOI-frame) */
int64_t tq_interleaved_floats_n(int64_t U, int32_t npix);
int tq_images_interleave_n(const float* images, float* images_il, int64_t U, int32_t npix, void* stream);

/* Per-unit data statistics for a single camera offset `offset[0]` (device scalar): with v = D - offset,
 *   pixstats[0][u] = sum_pix v,  pixstats[1][u] = sum_pix ln v,  pixstats[2][u] = #pixels with v <= 0
 * (a unit with such a pixel has log-likelihood -inf, ksmogn.py:226).  pixstats holds 3*U floats. */
int tq_image_stats(const float* images, const float* offset, float* pixstats, int64_t U, int32_t P, void* stream);

/* ---------------------------------------------------------------------------------------
 * One SVI step of the cosmos model = what `self.svi.step()` does at
 * tapqir/models/model.py:212 (pyro SVI + TraceEnum_ELBO(max_plate_nesting=3) over
 * cosmos.model / cosmos.guide, tapqir/models/cosmos.py:82-462, then pyro.optim.Adam,
 * model.py:169-171), split into stages so that a data-parallel host can put ONE all-reduce
 * of `gsum` between tq_cosmos_elbo_grads and tq_cosmos_globals_grad.
 *
 * Parameter storage.  All unconstrained parameters live in one flat float32 buffer
 *   params = [ local  : (8K+2) rows x U ]   U = Nt*F*C, row order:
 *                m_probs[k], h_loc[k], h_beta[k], w_mean[k], w_size[k], x_mean[k], y_mean[k],
 *                size[k]  (each K rows, k-major = the reference's (K,Nt,F,Q) tensors), b_loc, b_beta
 *            [ per-AOI: 2 rows x Nt*C ]     background_mean_loc, background_std_loc
 *            [ global : 4 + 5Q ]            gain_loc, gain_beta, proximity_loc, proximity_size,
 *                                           lamda_loc[Q], lamda_beta[Q], pi_mean[Q][2], pi_size[Q]
 * with the torch `transform_to(constraint)` maps of cosmos.py:471-598 (exp / sigmoid-affine /
 * exp+2 / softmax).  grad, exp_avg, exp_avg_sq have the same layout; `grad` receives
 * d ELBO / d param (Adam then descends on -ELBO).
 * ------------------------------------------------------------------------------------- */
typedef struct {
  /* dataset (device) */
  const float* images;         /* (Nt, F, C, P, P) */
  const float* images_il;      /* interleaved copy (tq_images_interleave) or NULL */
  const float* pixstats;       /* [3][Nt*F*C] data statistics (tq_image_stats) when O == 1, else NULL */
  const float* xy;             /* (Nt, F, C, 2) */
  const uint8_t* is_ontarget;  /* [Nt] */
  const uint8_t* aoi_mask;     /* [Nt] or NULL */
  const int32_t* ndx;          /* [nb] or NULL */
  const int32_t* fdx;          /* [fb] or NULL */
  const float* offset_samples; /* [O] (host merges duplicate samples) */
  const float* offset_logits;  /* [O] */
  /* parameters and optimiser state (device, flat, see above) */
  float* params;
  float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  /* workspace (device) */
  float* lat;                  /* [1+4K][B]        latent draws: b, h[K], w[K], x[K], y[K] */
  float* site;                 /* [5][(1+4K)*B]    per-site guide terms (log q, its derivatives w.r.t. the concentrations,
                                                   implicit reparameterisation gradients), same site order as lat */
  float* pix;                  /* [2^K+2+4K][B]    ll[2^K], g_b, g_gain, g_h[K], g_w[K], g_x[K], g_y[K]
                                                   (crosstalk: 1+Q more rows: ell_excess, g_alpha[Q]) */
  float* aoi_part;             /* [3][B]           per-unit d/d(bg mean, bg std) partials; row 2 = scratch */
  float* blk_part;             /* tq_cosmos_blk_floats() floats: per-workgroup partial sums, [nblk][3+3Q] (crosstalk:
                                                   3+3Q+Q*Q), or -- full-batch steps run by tq_cosmos_step /
                                                   tq_cosmos_step_overlapped -- [nblk][16 + 3+3Q(+Q*Q)] with the
                                                   per-AOI frame sums of the (at most two) AOIs a workgroup touches in front */
  double* gsum;                /* [TQ_GSUM_LEN]    cross-unit sums: d/d gain, d/d cs, ELBO, (d/d rho, a, c)[Q] in the
                                                   first 3+3Q entries (the part a data-parallel host
                                                   all-reduces); the tail is scratch of the library */
  void* globals;               /* TqGlobals  (tq_globals_size() bytes) */
  void* gbase;                 /* TqGlobalBase (tq_gbase_size() bytes): base draws of the global sites */
  double* elbo_out;            /* [1] ELBO of the step */
  /* geometry */
  int32_t Nt, F, C, P, K, O;
  int32_t nb, fb;
  int32_t n_offset;            /* global index of local AOI 0 (AOI sharding: RNG streams use global ids) */
  int32_t draw_globals;        /* 1: draw the global base variates; 0: use the contents of gbase */
  int32_t draw_locals;         /* 1: draw b, h, w, x, y; 0: use the contents of lat (parity tests) */
  int32_t il_min_units;        /* contiguous batches of at least this many units use the interleaved pixel kernel */
  float scale_n;               /* Nt_global / nb_global */
  float scale;                 /* scale_n * F / fb */
  float global_weight;         /* weight of the global ELBO part on this rank (1 on exactly one rank for reporting) */
  /* model constants */
  float eps;                   /* finfo(model dtype).eps */
  float width_min, width_max, height_std, background_mean_std, background_std_std;
  float gain_std, lamda_rate, proximity_rate;
  /* optimiser */
  float lr, beta1, beta2, adam_eps;
  float bias_correction1, bias_correction2;  /* 1 - beta^t for this step */
  int32_t zero_grad;           /* 1: Adam clears grad after use (minibatch mode keeps grad dense-zero) */
  int32_t fuse_adam;           /* 1: tq_cosmos_elbo_grads applies Adam to the local parameters of each unit of the
                                  batch right where their gradient is formed (no gradient round trip through HBM) and
                                  tq_cosmos_adam then updates only the per-AOI and global tail.  Full batches, or
                                  minibatches with `last_step` (lazy Adam, below) */
  int32_t crosstalk;           /* 1: the crosstalk model (tapqir/models/crosstalk.py; Q = C = 2, K <= 2): one data site per
                                  AOI-frame whose channel c sees every dye's spots scaled by alpha[q][c]; the global block
                                  of the parameter buffer grows by alpha_mean[Q][2], alpha_size[Q] (4+8Q entries), gsum
                                  by d/d alpha[q][c] (3+3Q+Q*Q entries) */
  /* RNG */
  uint64_t seed;
  uint32_t step;               /* number of completed steps: Philox key of this step's draws; this step is Adam step `step + 1` */
  /* lazy Adam of minibatch steps (tq_cosmos_adam_catchup) */
  int32_t* last_step;          /* [Nt*F*C] Adam step at which the local parameters of a unit were last updated, or NULL */
  double beta1_d, beta2_d;     /* the Adam betas in double: 1 - beta^s of the replayed steps is formed like the host's */
  int32_t pixel_mode;          /* 0 / 1: passed on to tq_ksmogn_args.pixel_mode; TQ_PIXEL_FUSED_UNIT (2): see tq_cosmos_pixel_unit */
  int32_t tail_kind;           /* how the pending tail of THIS step finds its partial sums: TQ_TAIL_AUTO (what tq_cosmos_step /
                                  _step_overlapped / _elbo_grads wrote) or TQ_TAIL_ROWS16 (the step ran as
                                  tq_cosmos_minibatch_step); the host sets it on the struct it later passes as `prev` / to
                                  tq_cosmos_tail */
  int32_t* sync;               /* [TQ_SYNC_WORDS] zero-initialised device words, or NULL: [0..2] workgroup tickets, flag and completion count of
                                  tq_cosmos_minibatch_step; [3] completion ticket of the per-AOI sums of a full-batch step with
                                  rows (tq_cosmos_elbo_grads / tq_cosmos_tail; without `sync` those take the flat layout / the
                                  single-workgroup tail); [40] count of the row groups added inside the sampling launch of
                                  tq_cosmos_step_overlapped (without `sync` its tail workgroup adds all rows itself); [60..62] of
                                  tq_cosmos_minibatch_step: which workgroup runs the tail, the second flag (global draws after the
                                  gain) and the gain of the launch.  Every launch leaves the counters re-armed. */
  int32_t sync_value;          /* value the flag takes in this launch: any value different from the previous launch's on the
                                  same `sync` words (a launch counter of the host) */
  int32_t images_by_slot;      /* 1: `images` is a window holding the AOIs of this batch in batch order (tq_ksmogn_args.images_by_slot):
                                  data sets larger than the device memory, whose AOIs the host streams in group by group */
  int32_t* next_ndx;           /* tq_cosmos_minibatch_step: [nb] or NULL, and */
  int32_t* next_fdx;           /* [fb] or NULL -- the launch also draws the NEXT step's subsample, `randperm(Nt)[:nb]` /
                                  `randperm(F)[:fb]` of pyro.plate (cosmos.py:194-208), as the nb / fb smallest of Nt / F Philox
                                  keys (stream: seed, step + 1; a radix selection, no sort) in its tail workgroup, after the flags are published: a host that
                                  passes them as ndx / fdx of the next call never draws, stages or copies an index
                                  (Nt, F <= TQ_SUBSAMPLE_MAX) */
} tq_cosmos_args;
#define TQ_SUBSAMPLE_MAX 2048

#define TQ_TAIL_AUTO 0
#define TQ_TAIL_ROWS16 1
#define TQ_SYNC_WORDS 64
#define TQ_PIXEL_FUSED_UNIT 2      /* tq_cosmos_args.pixel_mode: pixel + per-unit kernel of a full-batch step in one launch */

int64_t tq_globals_size(void);
int64_t tq_gbase_size(void);
int64_t tq_cosmos_nblk(int64_t B);              /* rows of blk_part (flat layout) */
int64_t tq_cosmos_blk_floats(int32_t Nt, int32_t F, int32_t C, int32_t crosstalk, int64_t B);  /* floats blk_part must hold */
int64_t tq_cosmos_param_count(int32_t Nt, int32_t F, int32_t C, int32_t K);
int64_t tq_crosstalk_param_count(int32_t Nt, int32_t F, int32_t C, int32_t K);  /* + alpha_mean, alpha_size */

/* guide draws: globals (gain, pi, lamda, proximity) + derived tables; then, one lane per
 * (unit, site), b, h, w, x, y (cosmos.py:342-368, 408-462; torch Gamma/Beta rsample + pyro
 * AffineBeta clamp) together with log q of the draw, its derivatives and the implicit
 * reparameterisation gradient of the draw -- everything about a guide site that depends on that
 * site alone */
int tq_cosmos_sample_globals(const tq_cosmos_args* a, void* stream);
int tq_cosmos_sample_locals(const tq_cosmos_args* a, void* stream);
/* likelihood + all per-unit / per-AOI ELBO terms and gradients; fills grad (local + AOI parts) and gsum */
int tq_cosmos_elbo_grads(const tq_cosmos_args* a, void* stream);
/* global sites: chain gsum through the tables to the unconstrained global parameters; writes elbo_out */
int tq_cosmos_globals_grad(const tq_cosmos_args* a, void* stream);
/* dense Adam over the whole flat buffer (torch.optim.Adam semantics, model.py:169-171) */
int tq_cosmos_adam(const tq_cosmos_args* a, void* stream);
/* all of the above back to back */
int tq_cosmos_step(const tq_cosmos_args* a, void* stream);
/* Lazy Adam for minibatch steps.  torch.optim.Adam updates EVERY element at every step; for the units outside the
 * minibatch the gradient is zero and the update (m *= beta1, v *= beta2, p -= lr_s m / (sqrt(v / bc2_s) + eps)) depends on
 * the element's own state and the step number only, so it can be replayed later from registers instead of streaming the
 * whole parameter / moment buffers through HBM every step (28 B per element: 35 us at config c2, 1.1 ms at c3).
 * tq_cosmos_adam_catchup replays, for every local parameter of the units of the batch (all_units = 0) or of the whole
 * dataset (all_units = 1), the zero-gradient steps last_step[u] + 1 .. a->step.  A minibatch step with fuse_adam then
 * runs it before its local sampling; tq_cosmos_elbo_grads applies step a->step + 1 and records it in last_step.  Call it
 * with all_units = 1 before anything reads the buffers (it does not write last_step: the caller re-bases the clock). */
int tq_cosmos_adam_catchup(const tq_cosmos_args* a, int32_t all_units, void* stream);
/* Full-batch pipeline (fuse_adam steps).  The single-workgroup TAIL of a step -- cross-unit sums, global sites, total
 * ELBO, Adam of the per-AOI and global parameters (~35 us of latency on one CU) -- does not have to finish before the
 * NEXT step samples its local guide sites (they read local parameters only, already updated by the fused Adam).
 * tq_cosmos_step_overlapped(a, prev) runs [local sampling of a + tail of prev + global draws of a] in ONE launch (the
 * tail occupies one extra workgroup of the ~14 000), then the likelihood, per-unit and per-AOI kernels of `a`, and
 * leaves the tail of `a` pending: pass `a` as `prev` of the next call, or finish it with tq_cosmos_tail(a) before
 * reading elbo_out / the per-AOI and global parameters.  prev == NULL: nothing pending (first step). */
int tq_cosmos_step_overlapped(const tq_cosmos_args* a, const tq_cosmos_args* prev, void* stream);
int tq_cosmos_tail(const tq_cosmos_args* a, void* stream);
/* With a->pixel_mode = TQ_PIXEL_FUSED_UNIT, tq_cosmos_step / tq_cosmos_step_overlapped run the likelihood and the
 * per-unit terms + Adam of a full-batch step as ONE launch: a wave renders its tile of 64 units and goes on to the
 * per-unit routine of the same units with the pixel results in registers (the pixel phase is bound by VALU issue, the
 * per-unit phase by HBM: waves in different phases overlap).  Full-batch cosmos steps with fuse_adam, K <= 2, one offset
 * value, P in {14, 20}, images_il and pixstats given, F * C >= 256; anything else is refused (TQ_ERR_ARG).  Same results
 * as the two-launch form up to the order of the fp32 partial sums (rows of 64 units instead of 256).
 * tq_cosmos_pixel_unit is that launch alone (a stage export like tq_cosmos_elbo_grads: draws and global tables of `a`
 * must be in place; it applies the Adam step of the local parameters and leaves the rows for the tail). */
int tq_cosmos_pixel_unit(const tq_cosmos_args* a, void* stream);
/* Minibatch steps (the reference's default operating point is 10 AOIs x 512 frames = 5120 units, main.py:1428-1431) in
 * ONE launch: every workgroup takes 16 units (20 when that saves a round of workgroups on the chip's CUs) through lazy-Adam catch-up, guide-site draws, likelihood and per-unit terms + Adam; one more workgroup runs
 * the pending tail of `prev` (or nothing) and the global draws of `a` -- the gain first, which the others wait for before
 * their likelihood phase, the rest before their per-unit phase.  Same arithmetic, same RNG streams and same results as
 * tq_cosmos_adam_catchup + tq_cosmos_step_overlapped; the tail of `a` stays pending: pass `a` (with tail_kind =
 * TQ_TAIL_ROWS16) as `prev` of the next step of either kind or to tq_cosmos_tail.  cosmos model, fuse_adam,
 * fb * C >= 16; `sync` must point to TQ_SYNC_WORDS zero-initialised int32 (zeroed again by the host after a launch that was
 * torn down). */
int tq_cosmos_minibatch_step(const tq_cosmos_args* a, const tq_cosmos_args* prev, void* stream);

/* AOI-sharded runs: everything that follows the all-reduce of gsum (tq_cosmos_globals_grad + tq_cosmos_adam) in one
 * single-workgroup launch, plus -- if `next` is given (full-batch steps) -- the global draws of the next step
 * (tq_cosmos_sample_globals(next)), which need the global parameters this call updates. */
int tq_cosmos_tail_reduced(const tq_cosmos_args* a, const tq_cosmos_args* next, void* stream);
/* AOI-sharded pipeline: draw the local sites [site_begin, site_begin + site_count) of `a` (site order as in `lat`).
 * With `prev` != NULL -- a full-batch step whose gsum the caller has all-reduced -- the same launch carries, as one extra
 * workgroup, tq_cosmos_tail_reduced(prev, a): a sharded host draws the first sites of step t+1 while the all-reduce of
 * step t is in flight, waits for it, and passes `prev` with the remaining sites, so that neither the collective nor the
 * single-workgroup tail is exposed. */
int tq_cosmos_sample_locals_range(const tq_cosmos_args* a, int32_t site_begin, int32_t site_count,
                                  const tq_cosmos_args* prev, void* stream);


/* ---------------------------------------------------------------------------------------
 * Posterior read-out: Monte-Carlo estimate (`particles` joint guide draws) of p(z | .) and p(theta = k | .).
 * Replaces cosmos.compute_probs (tapqir/models/cosmos.py:609-672): for every particle the
 * global latents (pi, lamda, proximity) and the spot positions x_k, y_k are drawn from the guide,
 * r(z, theta | m) = softmax_{z,theta}[log p(z) p(theta|z) prod_k p(m_k|theta) (p(x_k|theta) p(y_k|theta))^{m_k}],
 * R(z, theta) = sum_m prod_k q(m_k) r(z, theta | m);  z_probs = mean_particles sum_theta R,
 * theta_probs[k] = mean_particles R(z=1, theta=k+1).  Off-target AOIs are left at zero, as in the reference.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  const float* params;         /* flat unconstrained parameters (layout of tq_cosmos_args) */
  const uint8_t* is_ontarget;  /* [Nt] */
  void* globals_p;             /* [particles] TqGlobals (tq_globals_size() bytes each), workspace */
  void* gbase_p;               /* [particles] TqGlobalBase (tq_gbase_size() bytes each); input when draw == 0 */
  const float* xy_given;       /* [particles][2K][U] spot positions (x_0..x_{K-1}, y_0..y_{K-1}) when draw == 0, else NULL */
  float* z_probs;              /* (Nt, F, Q, 2) out */
  float* theta_probs;          /* (K, Nt, F, Q) out */
  int32_t Nt, F, C, P, K;
  int32_t particles;
  int32_t draw;                /* 1: draw latents from the guide; 0: use gbase_p / xy_given (parity tests) */
  float eps;
  uint64_t seed;
} tq_probs_args;

int tq_cosmos_probs(const tq_probs_args* a, void* stream);

/* ---------------------------------------------------------------------------------------
 * KSMOGN.rsample (tapqir/distributions/ksmogn.py:171-185), the sampler behind tapqir/utils/simulate.py:108-122:
 *   out[i, j, ic] = max(Gamma(mu / g, 1 / g), tiny) + offset_samples[odx],  odx ~ Categorical(offset_logits) per pixel,
 *   mu = background + sum_k height_k N(ic; x_k + tx, w_k) N(j; y_k + ty, w_k).
 * `height` carries the presence indicators (m_k h_k) and, for the crosstalk image of channel c, the fractions alpha_qc
 * (then K = Q K' spots per unit).  Draws are distributionally (not bitwise) those of torch: Philox4x32-10 +
 * Marsaglia-Tsang, one stream per pixel keyed by (seed, unit, pixel).
 * ------------------------------------------------------------------------------------- */
typedef struct {
  const float* height;         /* [K][B] */
  const float* width;          /* [K][B] */
  const float* x;              /* [K][B] */
  const float* y;              /* [K][B] */
  const float* xy;             /* [B][2] target locations */
  const float* background;     /* [B] */
  const float* gain;           /* [1] */
  const float* offset_samples; /* [O] */
  const float* offset_logits;  /* [O] */
  float* out;                  /* [B][P][P] */
  int64_t B;
  int32_t P, K, O;
  uint64_t seed;
} tq_rsample_args;

int tq_ksmogn_rsample(const tq_rsample_args* a, void* stream);

/* ---------------------------------------------------------------------------------------
 * Post-fit statistics of every unit (n, f, c): signal-to-noise ratio of each spot and chi2 of the fitted image.
 * Replaces snr_and_chi2 (tapqir/utils/stats.py:29-86) and the per-AOI host loop around it (stats.py:166-182):
 *   N_k      = N(ic; x_k + tx, w_k) N(j; y_k + ty, w_k)      with the posterior MEANS of the spot parameters
 *   snr[k]   = sum_ij (D - b - offset_mean) N_k / sqrt(offset_var + b gain)
 *   chi2     = mean_ij (D - (b + sum_k h_k N_k) - offset_mean)^2 / (b + sum_k h_k N_k)
 * ------------------------------------------------------------------------------------- */
typedef struct {
  const float* images;         /* (Nt, F, C, P, P) */
  const float* xy;             /* (Nt, F, C, 2) */
  const float* height;         /* [K][U] posterior means, U = Nt*F*C */
  const float* width;          /* [K][U] */
  const float* x;              /* [K][U] */
  const float* y;              /* [K][U] */
  const float* background;     /* [U] */
  float* snr;                  /* [K][U] out */
  float* chi2;                 /* [U] out */
  int64_t U;
  int32_t P, K;
  float gain, offset_mean, offset_var;
} tq_snr_args;

int tq_snr_chi2(const tq_snr_args* a, void* stream);

/* ---------------------------------------------------------------------------------------
 * Input side: AOI extraction from raw Glimpse frames.
 * Replaces the per-frame / per-AOI loop of read_glimpse (tapqir/imscroll/glimpse_reader.py:358-392), the frame
 * decode of GlimpseDataset.__getitem__ (168-186: big-endian int16 + 2^15) and the offset-region value counts
 * (362-369), for one chunk of `nf` consecutive frames of one colour channel:
 *
 *   shiftx = rint(x - 0.5 (P - 1)),  shifty = rint(y - 0.5 (P - 1))       (Python round(): half to even, 374-375)
 *   images[n, f0 + f, c, i, j]   = frame_f[shifty + i, shiftx + j] + 32768  (376-378)
 *   target_xy[n, f0 + f, c, :]   = (x - shiftx, y - shifty)                 (379-384)
 *   offset_hist[v]              += #{pixels == v in frame_f[offset_y : +offset_P, offset_x : +offset_P]}  (362-369)
 *
 * Integer work, bit-exact.  A crop that leaves the frame (the reference fails there with a numpy broadcasting
 * ValueError) is skipped and counted in status[0]; the caller raises.  status[1] receives the smallest extracted
 * pixel value (atomic min: initialise to INT32_MAX), which the offset post-processing needs (419-421).
 * ------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t* frames;   /* nf frames of H*W big-endian int16: the bytes of the .glimpse file, unmodified */
  const double* raw_xy;    /* (N, nf, 2) drift-corrected target positions (x, y) in frame pixels, float64 (347-350);
                              16-byte aligned */
  int32_t* images;         /* (N, F, C, P, P) out; only channel c of frames [f0, f0 + nf) is written */
  double* target_xy;       /* (N, F, C, 2) out */
  int64_t* offset_hist;    /* [65536] accumulated; NULL = no offset region in this call */
  int32_t* status;         /* [2]: {crops outside the frame (accumulated), min pixel (atomic min)} */
  int32_t H, W;            /* frame height, width (header.mat) */
  int32_t N, F, C, P;
  int32_t c, f0, nf;
  int32_t offset_x, offset_y, offset_P;
} tq_glimpse_args;

int tq_glimpse_extract(const tq_glimpse_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif
