// tq_aux.hip -- KSMOGN.rsample and the post-fit SNR / chi2 statistics on the device (bodies in tq_aux.h).
// Neither is on the SVI step path: the first generates synthetic data (tapqir/utils/simulate.py through
// KSMOGN.rsample, ksmogn.py:171-185), the second runs once per fit (tapqir/utils/stats.py:29-86, 166-193, where the
// reference loops over the AOIs on the host).  Both are bound by the image bytes they touch (4 P^2 per unit).
#include <hip/hip_runtime.h>

#include "tq_aux.h"

void tq_set_error(const char* msg);

static int aux_launch_status(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[200];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    tq_set_error(buf);
    return TQ_ERR_LAUNCH;
  }
  return TQ_OK;
}

// one lane per pixel; consecutive lanes write consecutive pixels of the (B, P, P) output
__global__ __launch_bounds__(256) void tq_rsample_kernel(const tq_rsample_args a, const int64_t total, const int npix) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int64_t i = t / npix;
  tq_body_rsample(a, i, (int)(t - i * npix));
}

// one lane per unit (a unit's tile is re-read from L2 / L1 by the same lane; the kernel runs once per fit)
__global__ __launch_bounds__(256) void tq_snr_chi2_kernel(const tq_snr_args a) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u < a.U) tq_body_snr_chi2(a, u);
}

extern "C" int tq_ksmogn_rsample(const tq_rsample_args* a, void* stream) {
  if (!a || !a->height || !a->width || !a->x || !a->y || !a->xy || !a->background || !a->gain || !a->offset_samples ||
      !a->offset_logits || !a->out) {
    tq_set_error("tq_ksmogn_rsample: NULL required pointer");
    return TQ_ERR_ARG;
  }
  if (a->B < 1 || a->P < 2 || a->P > TQ_MAX_P || a->K < 1 || a->K > 2 * TQ_MAX_K || a->O < 1) {
    tq_set_error("tq_ksmogn_rsample: unsupported B/P/K/O");
    return TQ_ERR_ARG;
  }
  const int npix = a->P * a->P;
  const int64_t total = a->B * npix;
  hipLaunchKernelGGL(tq_rsample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a, total, npix);
  return aux_launch_status("tq_rsample_kernel");
}

extern "C" int tq_snr_chi2(const tq_snr_args* a, void* stream) {
  if (!a || !a->images || !a->xy || !a->height || !a->width || !a->x || !a->y || !a->background || !a->snr || !a->chi2) {
    tq_set_error("tq_snr_chi2: NULL required pointer");
    return TQ_ERR_ARG;
  }
  if (a->U < 1 || a->P < 2 || a->P > TQ_MAX_P || a->K < 1 || a->K > TQ_MAX_K) {
    tq_set_error("tq_snr_chi2: unsupported U/P/K");
    return TQ_ERR_ARG;
  }
  hipLaunchKernelGGL(tq_snr_chi2_kernel, dim3((unsigned)((a->U + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
  return aux_launch_status("tq_snr_chi2_kernel");
}
