// Micro-benchmark of VALU issue rates on gfx950 (diagnostic; not part of the product): cycles per wave64
// instruction for v_fma_f32, v_pk_fma_f32, v_rcp_f32, v_log_f32, v_exp_f32, dependent vs independent chains,
// at 1..4 resident waves per SIMD, plus the shader clock under load (s_memtime vs the 100 MHz wall clock).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  const float c = 0.999f, d = 1e-3f;
  const f2 pc = {c, c}, pd = {d, d};
  const f2 sc = {0.5f, 0.25f};
  double d0 = a0, d1 = a1 * 0.5, d2 = a2 * 0.25, d3 = a3 * 0.125, d4 = 1.0 + a4 * 1e-3, d5 = a5, d6 = a6 * 0.5, d7 = a7;
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // independent v_fma_f32 x8
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
    } else if (MODE == 1) {  // dependent v_fma_f32
      REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(c), "v"(d));)
    } else if (MODE == 2) {  // independent v_pk_fma_f32 x8
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                        "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));)
    } else if (MODE == 3) {  // dependent v_pk_fma_f32
      REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n" : "+v"(p0) : "v"(pc), "v"(pd));)
    } else if (MODE == 4) {  // independent v_rcp_f32 x8
      REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 5) {  // independent v_log_f32 / v_exp_f32 alternating x8
      REP8(asm volatile("v_log_f32 %0, %0\n v_exp_f32 %1, %1\n v_log_f32 %2, %2\n v_exp_f32 %3, %3\n v_log_f32 %4, %4\n v_exp_f32 %5, %5\n v_log_f32 %6, %6\n v_exp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 6) {  // mix: 1 trans + 3 pk, independent (does the transcendental overlap with the packed ops?)
      REP8(asm volatile("v_rcp_f32 %0, %0\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n"
                        "v_rcp_f32 %1, %1\n v_pk_fma_f32 %7, %7, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));)
    } else if (MODE == 7) {  // dependent v_rcp_f32
      REP64(asm volatile("v_rcp_f32 %0, %0\n" : "+v"(a0));)
    } else if (MODE == 8) {  // v_pk_mul_f32 + v_pk_add_f32 independent x8 (4 + 4)
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n"
                        "v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n v_pk_mul_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));)
    } else if (MODE == 9) {  // v_pk_fma_f32, three DISTINCT register operands each, results not fed back (RF read bandwidth)
      REP8(asm volatile("v_pk_fma_f32 %0, %1, %2, %3\n v_pk_fma_f32 %1, %2, %3, %4\n v_pk_fma_f32 %2, %3, %4, %5\n v_pk_fma_f32 %3, %4, %5, %6\n"
                        "v_pk_fma_f32 %4, %5, %6, %7\n v_pk_fma_f32 %5, %6, %7, %0\n v_pk_fma_f32 %6, %7, %0, %1\n v_pk_fma_f32 %7, %0, %1, %2\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
    } else if (MODE == 10) {  // v_pk_mul_f32 / v_pk_add_f32 with two distinct register operands
      REP8(asm volatile("v_pk_mul_f32 %0, %1, %2\n v_pk_add_f32 %1, %2, %3\n v_pk_mul_f32 %2, %3, %4\n v_pk_add_f32 %3, %4, %5\n"
                        "v_pk_mul_f32 %4, %5, %6\n v_pk_add_f32 %5, %6, %7\n v_pk_mul_f32 %6, %7, %0\n v_pk_add_f32 %7, %0, %1\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
    } else if (MODE == 11) {  // v_fma_f32 with three distinct register operands
      REP8(asm volatile("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %1, %2, %3, %4\n v_fma_f32 %2, %3, %4, %5\n v_fma_f32 %3, %4, %5, %6\n"
                        "v_fma_f32 %4, %5, %6, %7\n v_fma_f32 %5, %6, %7, %0\n v_fma_f32 %6, %7, %0, %1\n v_fma_f32 %7, %0, %1, %2\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 12) {  // v_pk_fma_f32 accumulate form: acc = a * b + acc with distinct a, b per instruction
      REP8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %5, %6, %1\n v_pk_fma_f32 %2, %6, %7, %2\n v_pk_fma_f32 %3, %7, %4, %3\n"
                        "v_pk_fma_f32 %0, %6, %4, %0\n v_pk_fma_f32 %1, %7, %5, %1\n v_pk_fma_f32 %2, %4, %6, %2\n v_pk_fma_f32 %3, %5, %7, %3\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
    } else if (MODE == 14) {  // v_fma_f64, three distinct operands
      REP8(asm volatile("v_fma_f64 %0, %1, %2, %3\n v_fma_f64 %1, %2, %3, %4\n v_fma_f64 %2, %3, %4, %5\n v_fma_f64 %3, %4, %5, %6\n"
                        "v_fma_f64 %4, %5, %6, %7\n v_fma_f64 %5, %6, %7, %0\n v_fma_f64 %6, %7, %0, %1\n v_fma_f64 %7, %0, %1, %2\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
    } else if (MODE == 15) {  // v_mul_f64 / v_add_f64
      REP8(asm volatile("v_mul_f64 %0, %1, %2\n v_add_f64 %1, %2, %3\n v_mul_f64 %2, %3, %4\n v_add_f64 %3, %4, %5\n"
                        "v_mul_f64 %4, %5, %6\n v_add_f64 %5, %6, %7\n v_mul_f64 %6, %7, %0\n v_add_f64 %7, %0, %1\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
    } else if (MODE == 16) {  // v_cvt_f64_f32 / v_cvt_f32_f64 round trips
      REP8(asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f32_f64 %9, %1\n v_cvt_f64_f32 %2, %8\n v_cvt_f32_f64 %9, %3\n"
                        "v_cvt_f64_f32 %4, %8\n v_cvt_f32_f64 %9, %5\n v_cvt_f64_f32 %6, %8\n v_cvt_f32_f64 %9, %7\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(a0), "+v"(a1));)
    } else if (MODE == 17) {  // dependent v_fma_f64
      REP64(asm volatile("v_fma_f64 %0, %0, %1, %2\n" : "+v"(d0) : "v"(d1), "v"(d2));)
    } else if (MODE == 13) {  // v_pk_fma_f32 with an SGPR-pair operand: a * s + c
      REP8(asm volatile("v_pk_fma_f32 %0, %1, %8, %2\n v_pk_fma_f32 %1, %2, %8, %3\n v_pk_fma_f32 %2, %3, %8, %4\n v_pk_fma_f32 %3, %4, %8, %5\n"
                        "v_pk_fma_f32 %4, %5, %8, %6\n v_pk_fma_f32 %5, %6, %8, %7\n v_pk_fma_f32 %6, %7, %8, %0\n v_pk_fma_f32 %7, %0, %8, %1\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(sc));)
    }
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    cyc[0] = t1 - t0;
    cyc[1] = w1 - w0;
  }
}

template <int MODE>
void run(const char* name, int wavesPerSimd, float* out, long long* cyc) {
  // blocks of 256 threads = 4 waves = 1 per SIMD of a CU; wavesPerSimd blocks per CU x 256 CUs fill the chip
  const int blocks = 256 * wavesPerSimd, iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[2];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double n = 64.0 * iters;  // instructions per wave
  printf("%-34s waves/SIMD=%d  %.2f memtime-ticks/instr/wave  memtime rate %.0f MHz  kernel %.3f ms -> %.2f ns per instr per SIMD-slot\n",
         name, wavesPerSimd, h[0] / n, 100.0 * h[0] / h[1], ms, ms * 1e6 / (n * wavesPerSimd));
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
  hipMalloc(&cyc, 16);
  for (int w : {2, 4}) {
    run<0>("v_fma_f32 independent", w, out, cyc);
    run<1>("v_fma_f32 dependent", w, out, cyc);
    run<2>("v_pk_fma_f32 independent", w, out, cyc);
    run<3>("v_pk_fma_f32 dependent", w, out, cyc);
    run<8>("v_pk_mul/add_f32 independent", w, out, cyc);
    run<4>("v_rcp_f32 independent", w, out, cyc);
    run<7>("v_rcp_f32 dependent", w, out, cyc);
    run<5>("v_log/exp_f32 independent", w, out, cyc);
    run<6>("1 rcp + 3 pk_fma independent", w, out, cyc);
    run<9>("v_pk_fma_f32 3 distinct operands", w, out, cyc);
    run<12>("v_pk_fma_f32 acc += a*b distinct", w, out, cyc);
    run<13>("v_pk_fma_f32 a*sgpr+c", w, out, cyc);
    run<10>("v_pk_mul/add_f32 2 distinct", w, out, cyc);
    run<11>("v_fma_f32 3 distinct operands", w, out, cyc);
    run<14>("v_fma_f64 3 distinct operands", w, out, cyc);
    run<17>("v_fma_f64 dependent", w, out, cyc);
    run<15>("v_mul/add_f64", w, out, cyc);
    run<16>("v_cvt f64<->f32", w, out, cyc);
  }
  return 0;
}
