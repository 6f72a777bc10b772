// Micro-benchmark: how many independent VALU / LDS-read fillers hide in the gap between two v_mfma_f32_32x32x2_f32 of ONE wave per
// SIMD (4 waves per CU, one block per CU)?  Prints cycles per MFMA (s_memtime) for F fillers per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int F, int KIND, int THREADS>   // KIND 0: v_add_f32 fillers, 1: ds_read_b128, 2: v_pk_add_f32, 3: s_add_u32, 4: F v_add_f32 in ONE gap of 16
__global__ void __launch_bounds__(THREADS, 1) k(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = threadIdx.x * 1e-3f, y = 1.0f;
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = x + i;
  f32x4 q[4] = {};
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pk[8]; for (int i = 0; i < 8; ++i) pk[i] = f32x2{x + i, x - i};
  f32x2 py = {1.f, 2.f};
  unsigned sc = 0;
  const unsigned ldsb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  const float* gsrc = out + 65536;
  const float* lp = lds + (threadIdx.x & 63) * 4;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < (KIND == 4 ? (m == 0 ? 16 * F : 0) : F); ++i) {
        if (KIND == 0 || KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(m * F + i) & 15]) : "v"(y));
        else if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[(m * F + i) & 7]) : "v"(py));
        else if (KIND == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
        else if (KIND == 5) {   // direct-to-LDS piece (1 KB per wave), as in wino_conv.hip
          unsigned keep; const unsigned dst = __builtin_amdgcn_readfirstlane(ldsb + 1024u * ((m * F + i) & 15) + 4096u * (threadIdx.x >> 6) * 4);
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"((threadIdx.x & 63) * 16), "s"(gsrc + 256 * ((m * F + i) & 15)), "s"(dst) : "memory");
        }
        else if (KIND == 6) { f32x4 t; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(t) : "v"((threadIdx.x & 63) * 16), "s"(gsrc + 256 * ((m * F + i) & 15)) : "memory"); q[i & 3] = t; }
        else if (KIND == 7) asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(size_t)lp & 0xffff), "v"(q[i & 3]) : "memory");
        else { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)lp & 0xffff)); q[i & 3] = t; }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == 1 || KIND == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (KIND == 5 || KIND == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 16; ++i) s += f[i];
  for (int i = 0; i < 4; ++i) s += q[i][0];
  for (int i = 0; i < 8; ++i) s += pk[i][0] + pk[i][1];
  s += (float)sc;
  out[blockIdx.x * THREADS + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int F, int KIND, int THREADS = 256> void run(float* out, long long* cyc) {
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<F, KIND, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<F, KIND, THREADS>), dim3(256), dim3(THREADS), 100 * 1024, 0, out, cyc, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<F, KIND, THREADS>), dim3(256), dim3(THREADS), 100 * 1024, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("kind %d threads %d fillers/MFMA %2d: %.1f ns per MFMA per wave, %.1f TF/s (256 CUs), counter %.1f per MFMA\n", KIND, THREADS, F,
         ms * 1e6 / (iters * 16.0), 256.0 * (THREADS / 64) * 16 * iters * 4096 / (ms * 1e-3) / 1e12, (double)c / (iters * 16.0));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4 + (1 << 20)); hipMalloc(&cyc, 8);
  run<0, 0>(out, cyc); run<1, 0>(out, cyc); run<2, 0>(out, cyc); run<4, 0>(out, cyc); run<6, 0>(out, cyc); run<8, 0>(out, cyc);
  run<12, 0>(out, cyc); run<16, 0>(out, cyc);
  run<1, 1>(out, cyc); run<2, 1>(out, cyc); run<4, 1>(out, cyc);
  run<2, 2>(out, cyc); run<4, 2>(out, cyc); run<8, 2>(out, cyc);
  run<2, 3>(out, cyc); run<4, 3>(out, cyc); run<8, 3>(out, cyc);
  run<2, 4>(out, cyc); run<4, 4>(out, cyc); run<8, 4>(out, cyc);
  run<1, 5>(out, cyc); run<2, 5>(out, cyc);
  run<1, 6>(out, cyc); run<2, 6>(out, cyc);
  run<1, 7>(out, cyc); run<2, 7>(out, cyc);
  run<0, 0, 512>(out, cyc); run<2, 0, 512>(out, cyc); run<4, 0, 512>(out, cyc); run<8, 0, 512>(out, cyc);
  return 0;
}
