// Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (needed by a hi/lo split of fp32 operands, whose low parts are tiny)?
// A = all (2^-20) (subnormal in fp16: below 2^-14), B = all 1.0 -> every C element should be 16 * 2^-20 = 1.52587890625e-05; 0 = flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float aval) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)1.0f; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  out[threadIdx.x] = c[0];
  // and the conversion itself: does (float -> half) keep subnormals?
  out[64 + threadIdx.x] = (float)a[0];
}
int main() {
  float* d; hipMalloc(&d, 128 * sizeof(float));
  const float vals[3] = {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24: smallest subnormal */, 1.0f / 1024};
  for (float v : vals) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v);
    float h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("a = %.10e: half(a) = %.10e, mfma sum of 16 = %.10e (expected %.10e)\n", v, h[64], h[0], 16.0 * v);
  }
  return 0;
}
