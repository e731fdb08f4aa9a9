// AOI extraction from raw Glimpse frames (tapqir/imscroll/glimpse_reader.py:358-392).
//
// Byte/integer work bound by HBM: per extracted AOI-frame 2 P^2 bytes of big-endian int16 are gathered from the
// frame (P-pixel row segments at an arbitrary corner) and 4 P^2 bytes of int32 are streamed out, contiguous over
// (frame, pixel) for one AOI and one channel.  One lane per output pixel; consecutive lanes walk one crop row by
// row, so a wave reads ~4.5 row segments of 28 B and writes 256 contiguous bytes.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tapqir_hip.h"

void tq_set_error(const char* msg);

#define TQ_HIST_BINS 65536
#define TQ_HIST_WINDOW 8192  // LDS window of the per-frame offset histogram (camera offsets spread over a few hundred ADU)

// glimpse_reader.py:181-186: np.fromfile(dtype=">i2") + 2**15
__device__ __forceinline__ int tq_glimpse_pixel(const uint8_t* __restrict__ frames, int64_t idx) {
  const uint16_t raw = *reinterpret_cast<const uint16_t*>(frames + 2 * idx);  // little-endian load of a big-endian value
  const int16_t be = (int16_t)(uint16_t)((raw << 8) | (raw >> 8));
  return (int)be + 32768;
}

struct TqCropWindow {
  double x, y, rx, ry;
  int64_t sx, sy;
  bool inside;
};

// Python round(float): to nearest, ties to even = rint in the default rounding mode (glimpse_reader.py:374-375)
__device__ __forceinline__ TqCropWindow tq_crop_window(const tq_glimpse_args& a, int64_t r) {
  TqCropWindow w;
  const double2 p = *reinterpret_cast<const double2*>(a.raw_xy + 2 * r);
  w.x = p.x;
  w.y = p.y;
  const double half = 0.5 * (double)(a.P - 1);
  const bool finite = fabs(w.x) < 1e9 && fabs(w.y) < 1e9;
  w.rx = finite ? rint(w.x - half) : -1.0;
  w.ry = finite ? rint(w.y - half) : -1.0;
  w.sx = (int64_t)w.rx;
  w.sy = (int64_t)w.ry;
  w.inside = finite && w.sx >= 0 && w.sy >= 0 && w.sx + a.P <= a.W && w.sy + a.P <= a.H;
  return w;
}

__device__ __forceinline__ void tq_crop_finish(const tq_glimpse_args& a, int vmin, int bad) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    vmin = min(vmin, __shfl_xor(vmin, o, 64));
    bad += __shfl_xor(bad, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    // one atomic per wave on a single address would serialise the whole grid in L2: look first, the running minimum
    // settles after a few waves
    if (vmin < __atomic_load_n(&a.status[1], __ATOMIC_RELAXED)) atomicMin(&a.status[1], vmin);
    if (bad) atomicAdd(&a.status[0], bad);
  }
}

// Any P.  grid: x over (frame of the chunk, pixel), y = AOI; one lane per pixel.
__global__ __launch_bounds__(256) void tq_glimpse_crop_kernel(const tq_glimpse_args a, const unsigned per_aoi) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  const int P = a.P, PP = P * P;
  int vmin = INT_MAX;
  int bad = 0;
  if (t < per_aoi) {
    const int f = (int)(t / (unsigned)PP);
    const int pix = (int)(t - (unsigned)f * (unsigned)PP);
    const int64_t n = blockIdx.y;
    const TqCropWindow w = tq_crop_window(a, n * a.nf + f);
    const int64_t u = (n * a.F + (a.f0 + f)) * a.C + a.c;
    if (w.inside) {
      const int i = pix / P, j = pix - i * P;
      const int v = tq_glimpse_pixel(a.frames, ((int64_t)f * a.H + (w.sy + i)) * a.W + (w.sx + j));
      a.images[u * PP + pix] = v;
      vmin = v;
    } else if (pix == 0) {
      bad = 1;
    }
    if (pix == 0) {  // glimpse_reader.py:379-384
      a.target_xy[2 * u] = w.x - w.rx;
      a.target_xy[2 * u + 1] = w.y - w.ry;
    }
  }
  tq_crop_finish(a, vmin, bad);
}

// Even P (so that pairs do not straddle rows): one lane per pair of pixels of a row, 8-byte stores.  PC = P at compile
// time (index arithmetic by constants) or 0 = any even P.
template <int PC>
__global__ __launch_bounds__(256) void tq_glimpse_crop_pair_kernel(const tq_glimpse_args a, const unsigned per_aoi) {
  static_assert(PC % 2 == 0, "pairs must not straddle rows");
  const int P = PC ? PC : a.P;
  const int HP = P / 2, NPAIR = P * HP;
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  int vmin = INT_MAX;
  int bad = 0;
  if (t < per_aoi) {
    const int f = (int)(t / (unsigned)NPAIR);
    const int pair = (int)(t - (unsigned)f * (unsigned)NPAIR);
    const int64_t n = blockIdx.y;
    const TqCropWindow w = tq_crop_window(a, n * a.nf + f);
    const int64_t u = (n * a.F + (a.f0 + f)) * a.C + a.c;
    if (w.inside) {
      const int i = pair / HP, j = 2 * (pair - i * HP);
      const int64_t src = ((int64_t)f * a.H + (w.sy + i)) * a.W + (w.sx + j);
      const int v0 = tq_glimpse_pixel(a.frames, src), v1 = tq_glimpse_pixel(a.frames, src + 1);
      *reinterpret_cast<int2*>(a.images + u * (P * P) + 2 * pair) = make_int2(v0, v1);
      vmin = min(v0, v1);
    } else if (pair == 0) {
      bad = 1;
    }
    if (pair == 0) {
      a.target_xy[2 * u] = w.x - w.rx;
      a.target_xy[2 * u + 1] = w.y - w.ry;
    }
  }
  tq_crop_finish(a, vmin, bad);
}

// Value counts of the offset region of every frame (glimpse_reader.py:362-369).  One workgroup per frame: counts go
// to an LDS window anchored at the frame's smallest offset pixel and are flushed with one 64-bit atomic per occupied
// bin; values above the window (hot pixels) go to the global histogram directly.
__global__ __launch_bounds__(256) void tq_glimpse_hist_kernel(const tq_glimpse_args a, const int rows, const int cols) {
  __shared__ unsigned s_hist[TQ_HIST_WINDOW];
  __shared__ int s_min;
  const int f = blockIdx.x, tid = threadIdx.x;
  const int npix = rows * cols;
  if (tid == 0) s_min = INT_MAX;
  for (int b = tid; b < TQ_HIST_WINDOW; b += 256) s_hist[b] = 0u;
  __syncthreads();
  int vmin = INT_MAX;
  for (int p = tid; p < npix; p += 256) {
    const int i = p / cols, j = p - i * cols;
    vmin = min(vmin, tq_glimpse_pixel(a.frames, ((int64_t)f * a.H + (a.offset_y + i)) * a.W + (a.offset_x + j)));
  }
  if (vmin != INT_MAX) atomicMin(&s_min, vmin);
  __syncthreads();
  const int base = s_min;
  unsigned long long* hist = reinterpret_cast<unsigned long long*>(a.offset_hist);
  for (int p = tid; p < npix; p += 256) {
    const int i = p / cols, j = p - i * cols;
    const int v = tq_glimpse_pixel(a.frames, ((int64_t)f * a.H + (a.offset_y + i)) * a.W + (a.offset_x + j));
    const int d = v - base;
    if (d < TQ_HIST_WINDOW) atomicAdd(&s_hist[d], 1u);
    else atomicAdd(&hist[v], 1ull);
  }
  __syncthreads();
  for (int b = tid; b < TQ_HIST_WINDOW; b += 256) {
    const unsigned cnt = s_hist[b];
    if (cnt) atomicAdd(&hist[base + b], (unsigned long long)cnt);
  }
}

extern "C" int tq_glimpse_extract(const tq_glimpse_args* a, void* stream) {
  if (!a || !a->frames || !a->status || (a->N > 0 && (!a->raw_xy || !a->images || !a->target_xy))) {
    tq_set_error("tq_glimpse_extract: NULL required pointer");
    return TQ_ERR_ARG;
  }
  if (a->H <= 0 || a->W <= 0 || a->N < 0 || a->F <= 0 || a->C <= 0 || a->P < 1 || a->P > a->H || a->P > a->W ||
      a->c < 0 || a->c >= a->C || a->nf <= 0 || a->f0 < 0 || a->f0 + (int64_t)a->nf > a->F ||
      (a->offset_hist && (a->offset_x < 0 || a->offset_y < 0 || a->offset_P < 0))) {
    tq_set_error("tq_glimpse_extract: inconsistent frame / AOI / chunk geometry");
    return TQ_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (a->N > 0) {
    const bool pairs = a->P % 2 == 0;
    const int64_t per_aoi = (int64_t)a->nf * a->P * a->P / (pairs ? 2 : 1);
    if ((int64_t)a->nf * a->P * a->P >= 0x7fffffffLL || a->N > 65535) {
      tq_set_error("tq_glimpse_extract: chunk too large (nf * P^2 < 2^31 and N <= 65535 per call)");
      return TQ_ERR_ARG;
    }
    const dim3 grid((unsigned)((per_aoi + 255) / 256), (unsigned)a->N), block(256);
    if (a->P == 14)  // the AOI size of tapqir's defaults (main.py:1428)
      hipLaunchKernelGGL(tq_glimpse_crop_pair_kernel<14>, grid, block, 0, st, *a, (unsigned)per_aoi);
    else if (pairs) hipLaunchKernelGGL(tq_glimpse_crop_pair_kernel<0>, grid, block, 0, st, *a, (unsigned)per_aoi);
    else hipLaunchKernelGGL(tq_glimpse_crop_kernel, grid, block, 0, st, *a, (unsigned)per_aoi);
  }
  if (a->offset_hist) {
    // numpy slicing clips the offset region to the frame (glimpse_reader.py:362-365)
    const int rows = max(0, min(a->offset_P, a->H - a->offset_y)), cols = max(0, min(a->offset_P, a->W - a->offset_x));
    if (rows > 0 && cols > 0)
      hipLaunchKernelGGL(tq_glimpse_hist_kernel, dim3((unsigned)a->nf), dim3(256), 0, st, *a, rows, cols);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[200];
    snprintf(buf, sizeof(buf), "tq_glimpse_extract: %s", hipGetErrorString(e));
    tq_set_error(buf);
    return TQ_ERR_LAUNCH;
  }
  return TQ_OK;
}
