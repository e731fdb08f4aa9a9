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
  const dim3 grid((unsigned)((Bg + 15) / 16)), block(256);
  hipStream_t st = (hipStream_t)stream;
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
