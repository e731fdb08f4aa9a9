// Where does the dispatcher put the workgroups of a launch that fits the chip in one round?  Workgroups of 256 threads with the
// register footprint of tq_minibatch_kernel (256 VGPRs: two waves per SIMD), each alive ~40 us, record their XCC and CU.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/wg_placement.hip -o scripts/ubench/wg_placement && scripts/ubench/wg_placement 257 321
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(uint32_t* out, uint64_t* t, float* sink) {
  __shared__ float pad[2316];
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  float acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = (float)(threadIdx.x + i);
  while (__builtin_amdgcn_s_memrealtime() - t0 < 4000) {  // 40 us of the 100 MHz clock
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += acc[i];
  pad[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
    t[2 * blockIdx.x] = t0;
    t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    sink[blockIdx.x] = pad[(int)s & 255];
  }
}

int main(int argc, char** argv) {
  for (int ai = 1; ai < argc; ++ai) {
    const int n = atoi(argv[ai]);
    uint32_t* out;
    uint64_t* t;
    float* sink;
    hipMalloc(&out, 8 * n);
    hipMalloc(&t, 16 * n);
    hipMalloc(&sink, 4 * n);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, out, t, sink);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(2 * n);
    std::vector<uint64_t> ht(2 * n);
    hipMemcpy(h.data(), out, 8 * n, hipMemcpyDeviceToHost);
    hipMemcpy(ht.data(), t, 16 * n, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> per_cu;
    uint64_t tmin = ~0ull, tmax = 0;
    for (int b = 0; b < n; ++b) {
      const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 15;
      const uint32_t cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
      if (ht[2 * b] < tmin) tmin = ht[2 * b];
      if (ht[2 * b + 1] > tmax) tmax = ht[2 * b + 1];
    }
    int hist[8] = {0};
    for (auto& kv : per_cu) hist[kv.second < 7 ? kv.second : 7]++;
    printf("grid %d: %zu distinct CUs; CUs with 1 / 2 / 3 / 4 workgroups: %d / %d / %d / %d; first start -> last end %.1f us\n", n,
           per_cu.size(), hist[1], hist[2], hist[3], hist[4], (tmax - tmin) / 100.0);
    if (getenv("VERBOSE"))
      for (int b = 0; b < n; ++b)
        printf("  wg %3d xcc %u hw %08x start %.1f\n", b, h[2 * b + 1] & 15, h[2 * b], (ht[2 * b] - tmin) / 100.0);
    hipFree(out); hipFree(t); hipFree(sink);
  }
  return 0;
}
