// Micro-benchmark: the rate a bare stream of v_mfma_f32_32x32x16_f16 sustains on the whole chip (every CU, 1 or 2 waves per SIMD, NACC
// independent accumulators, milliseconds long so that the clock settles under the load), timed with HIP events.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma16_peak.hip -o tools/ubench/mfma16_peak.bin && tools/ubench/mfma16_peak.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int WPS, int RANDOM>
__global__ void __launch_bounds__(256, WPS) k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  f16x8 x[4], y[2];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  auto rnd = [&]() { h = h * 1664525u + 1013904223u; return RANDOM ? (_Float16)(((int)(h >> 9) & 0xffff) * (1.0f / 32768.0f) - 1.0f) : (_Float16)(1.0f + (h >> 30)); };
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) x[i][e] = rnd();
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) y[i][e] = rnd();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int a = m % NACC;
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[(m >> 1) & 3], y[m & 1], acc[a], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int WPS, int RANDOM> void run(float* out, int cus) {
  const int iters = 40000, blocks = cus * WPS;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, WPS, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, 1000);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, WPS, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 32 * 32 * 16 * 16.0 * iters * 4 * blocks;
  const double cyc_at_24 = ms * 1e-3 * 2.4e9 / (16.0 * iters * WPS);
  printf("%s operands, accumulators %d, waves/SIMD %d: %.2f ms, %.0f TFLOP/s, %.1f cycles per MFMA and SIMD at 2.4 GHz\n", RANDOM ? "random" : "small-integer", NACC, WPS, ms, flops / ms * 1e-9, cyc_at_24);
}

int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  printf("%s, %d CUs, clock %d kHz\n", pr.name, cus, pr.clockRate);
  float* out; hipMalloc(&out, 4 << 20);
  run<8, 1, 0>(out, cus); run<8, 2, 0>(out, cus); run<8, 1, 1>(out, cus); run<8, 2, 1>(out, cus); run<8, 2, 1>(out, cus); run<8, 2, 0>(out, cus);
  return 0;
}
