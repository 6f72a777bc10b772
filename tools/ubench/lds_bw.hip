// Micro-benchmark: LDS read bandwidth per CU by instruction width and wave count (conflict-free, lane-linear addresses).
// One block per CU; every wave issues N reads of its own 64 x WIDTH bytes window; prints bytes per cycle per CU (2.4 GHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>   // 0: ds_read_b128, 1: ds_read_b64, 2: ds_read_b32, 3: ds_read2_b64 (two 8-byte halves 16 B apart = the same 16 bytes), 4: ds_write_b128
__global__ void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int W = KIND == 0 || KIND == 3 || KIND == 4 ? 16 : KIND == 1 ? 8 : 4;
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + wave * 4096 + lane * W;
  f32x4 a[8] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned ad = base + ((u & 3) * 1024);
      if (KIND == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(a[u]) : "v"(ad));
      else if (KIND == 1) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(ad)); a[u][0] = t[0]; a[u][1] = t[1]; }
      else if (KIND == 2) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(ad)); a[u][0] = t; }
      else if (KIND == 3) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1" : "=v"(a[u]) : "v"(ad));
      else asm volatile("ds_write_b128 %0, %1" :: "v"(ad), "v"(a[u]) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0.f;
  for (int u = 0; u < 8; ++u) s += a[u][0] + a[u][1] + a[u][2] + a[u][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> void run(float* out, int threads) {
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 100 * 1024, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 100 * 1024, 0, out, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int W = KIND == 0 || KIND == 3 || KIND == 4 ? 16 : KIND == 1 ? 8 : 4;
  const double bytes = (double)iters * 8 * threads * W;         // per CU
  printf("kind %d (%2d B/lane) %d waves/CU: %.1f B per cycle per CU (at 2.4 GHz)\n", KIND, W, threads / 64, bytes / (ms * 1e-3 * 2.4e9));
}

int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  for (int t : {256, 512, 1024}) { run<0>(out, t); run<1>(out, t); run<2>(out, t); run<3>(out, t); run<4>(out, t); }
  return 0;
}
