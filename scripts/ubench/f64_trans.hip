// Micro-benchmark (diagnostic; not part of the product): issue cost and accuracy of the fp64 hardware seeds
// v_rcp_f64 / v_rsq_f64 / v_sqrt_f64 on gfx950, next to the fp32-seeded Newton forms of tq_math.h (tq_drcp, tq_dsqrt).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>

#define REP8(x) x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  double d0 = 1.0 + threadIdx.x * 1e-3, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
    } else if (MODE == 1) {
      REP8(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
    } else if (MODE == 2) {
      REP8(asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3\n v_sqrt_f64 %4, %4\n v_sqrt_f64 %5, %5\n v_sqrt_f64 %6, %6\n v_sqrt_f64 %7, %7\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
    } else {
      REP8(asm volatile("v_fma_f64 %0, %1, %2, %3\n v_fma_f64 %1, %2, %3, %4\n v_fma_f64 %2, %3, %4, %5\n v_fma_f64 %3, %4, %5, %6\n"
                        "v_fma_f64 %4, %5, %6, %7\n v_fma_f64 %5, %6, %7, %0\n v_fma_f64 %6, %7, %0, %1\n v_fma_f64 %7, %0, %1, %2\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
}

__global__ void acc(double* err) {  // worst relative error of the raw seeds over a sweep of arguments
  double worst_rcp = 0, worst_rsq = 0, worst_sqrt = 0;
  for (int i = 0; i < 4096; ++i) {
    const double x = 0.37 + 1.61803398875 * (threadIdx.x * 4096 + i) * 1e-3;
    double r, q, s;
    asm volatile("v_rcp_f64 %0, %1" : "=v"(r) : "v"(x));
    asm volatile("v_rsq_f64 %0, %1" : "=v"(q) : "v"(x));
    asm volatile("v_sqrt_f64 %0, %1" : "=v"(s) : "v"(x));
    worst_rcp = fmax(worst_rcp, fabs(r * x - 1.0));
    worst_rsq = fmax(worst_rsq, fabs(q * q * x - 1.0) * 0.5);
    worst_sqrt = fmax(worst_sqrt, fabs(s * s / x - 1.0) * 0.5);
  }
  err[3 * threadIdx.x] = worst_rcp;
  err[3 * threadIdx.x + 1] = worst_rsq;
  err[3 * threadIdx.x + 2] = worst_sqrt;
}

template <int MODE>
void run(const char* name, double* out) {
  const int blocks = 512, iters = 1000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-12s waves/SIMD=2  %.2f ns per instr per SIMD-slot\n", name, ms * 1e6 / (64.0 * iters * 2));
}

int main() {
  double *out, *err;
  hipMalloc(&out, 512 * 256 * sizeof(double));
  hipMalloc(&err, 3 * 64 * sizeof(double));
  run<3>("v_fma_f64", out);
  run<0>("v_rcp_f64", out);
  run<1>("v_rsq_f64", out);
  run<2>("v_sqrt_f64", out);
  hipLaunchKernelGGL(acc, dim3(1), dim3(64), 0, 0, err);
  double h[3 * 64];
  hipMemcpy(h, err, sizeof(h), hipMemcpyDeviceToHost);
  double w[3] = {0, 0, 0};
  for (int i = 0; i < 64; ++i)
    for (int j = 0; j < 3; ++j) w[j] = fmax(w[j], h[3 * i + j]);
  printf("worst relative error of the raw result: v_rcp_f64 %.3g (2^%.1f)  v_rsq_f64 %.3g (2^%.1f)  v_sqrt_f64 %.3g (2^%.1f)\n", w[0], log2(w[0]),
         w[1], log2(w[1]), w[2], log2(w[2]));
  return 0;
}
