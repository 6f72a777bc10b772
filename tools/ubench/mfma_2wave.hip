// Micro-benchmark: do the vector-ALU instructions / LDS reads of a SECOND wave on the same SIMD hide behind the fp32 MFMAs of the
// first one?  (tools/ubench/mfma_fill.hip: with ONE wave per SIMD every VALU instruction beside v_mfma_f32_32x32x2_f32 costs its
// issue cycles.)  Blocks of 512 threads = two waves per SIMD, one block per CU (100 KB of LDS).
//   ROLE 0  both waves of a SIMD run MFMA + F fillers per MFMA (symmetric: the two-waves-per-SIMD kernel shape)
//   ROLE 1  waves 0-3 run MFMAs only, waves 4-7 run F*16 fillers per 16-MFMA period of the others (asymmetric, same duration)
// Prints ns per MFMA per SIMD and the chip-wide rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int F, int KIND, int ROLE, int BITS16>   // KIND 0: v_add_f32, 1: ds_read_b128, 2: v_pk_add_f32;  BITS16: 32x32x16 f16 MFMAs instead
__global__ void __launch_bounds__(512, 1) k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = threadIdx.x * 1e-3f, y = 1.0f;
  f16x8 hx, hy; for (int i = 0; i < 8; ++i) { hx[i] = (_Float16)(x + i); hy[i] = (_Float16)1.0f; }
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = x + i;
  f32x4 q[4] = {};
  f32x2 pk[8]; for (int i = 0; i < 8; ++i) pk[i] = f32x2{x + i, x - i};
  f32x2 py = {1.f, 2.f};
  const float* lp = lds + (threadIdx.x & 63) * 4;
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = ROLE == 0 || wave < 4, do_fill = ROLE == 0 || wave >= 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (do_mfma) {
        if (BITS16) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, acc[m & 3], 0, 0, 0);
        else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[m & 3], 0, 0, 0);
      }
      if (do_fill) {
#pragma unroll
        for (int i = 0; i < F; ++i) {
          if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(m * F + i) & 15]) : "v"(y));
          else if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[(m * F + i) & 7]) : "v"(py));
          else { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)lp & 0xffff)); q[i & 3] = t; }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 16; ++i) s += f[i];
  for (int i = 0; i < 4; ++i) s += q[i][0];
  for (int i = 0; i < 8; ++i) s += pk[i][0] + pk[i][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int F, int KIND, int ROLE, int BITS16 = 0> void run(float* out) {
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<F, KIND, ROLE, BITS16>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<F, KIND, ROLE, BITS16>), dim3(256), dim3(512), 100 * 1024, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<F, KIND, ROLE, BITS16>), dim3(256), dim3(512), 100 * 1024, 0, out, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (ROLE == 0 ? 2.0 : 1.0) * 16.0 * iters;          // MFMAs issued per SIMD
  const double flop = BITS16 ? 32768.0 : 4096.0;
  printf("%s role %d kind %d fillers/MFMA %2d: %.1f ns per MFMA per SIMD, %.1f TF/s (256 CUs)\n", BITS16 ? "f16x16" : "f32x2 ", ROLE, KIND, F,
         ms * 1e6 / mfma_per_simd, 1024.0 * mfma_per_simd * flop / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0, 0, 0>(out); run<1, 0, 0>(out); run<2, 0, 0>(out); run<4, 0, 0>(out); run<8, 0, 0>(out);
  run<2, 2, 0>(out); run<4, 2, 0>(out);
  run<1, 1, 0>(out); run<2, 1, 0>(out);
  run<0, 0, 1>(out); run<2, 0, 1>(out); run<4, 0, 1>(out); run<8, 0, 1>(out); run<12, 0, 1>(out);
  run<4, 2, 1>(out); run<2, 1, 1>(out);
  run<0, 0, 0, 1>(out); run<2, 0, 0, 1>(out); run<4, 0, 0, 1>(out); run<8, 0, 0, 1>(out);
  run<0, 0, 1, 1>(out); run<2, 0, 1, 1>(out); run<4, 0, 1, 1>(out); run<6, 0, 1, 1>(out);
  return 0;
}
