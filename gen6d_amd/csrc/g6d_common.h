// Shared device/host helpers for libgen6d_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gen6d_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define G6D_WAVE 64

// Reduced-precision matrix-core mode (G6dConv.math_mode / g6d_corr2d_patch's math_mode; opt-in speed mode, fp32 stays the
// default and the parity path): operands rounded to bf16 (1) or fp16 (2) when the fragments leave LDS, fp32 accumulation.
// Two consecutive 8-channel fragment slices (lane-half h holds channels 4h..4h+3 of each) form the 8 operand slots of one
// v_mfma_f32_32x32x16_{bf16,f16}: the K permutation is the same for A and B, so the sum is over the same 16 channels.
template <int MM>
__device__ __forceinline__ f32x16 g6d_mfma_lowp(f32x4 alo, f32x4 ahi, f32x4 blo, f32x4 bhi, f32x16 acc) {
  if constexpr (MM == 1) {
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = (__bf16)alo[i]; a[4 + i] = (__bf16)ahi[i]; b[i] = (__bf16)blo[i]; b[4 + i] = (__bf16)bhi[i]; }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  } else {
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = (_Float16)alo[i]; a[4 + i] = (_Float16)ahi[i]; b[i] = (_Float16)blo[i]; b[4 + i] = (_Float16)bhi[i]; }
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
}

int g6d_check_launch(const char* what);   // returns G6D_OK or G6D_ELAUNCH, records the error string
void g6d_set_error(const char* msg);
void g6d_allow_lds(const void* func, int bytes);   // hipFuncAttributeMaxDynamicSharedMemorySize, once per (kernel, device)

static inline bool g6d_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Launch-policy knobs (common.hip): every dispatch decision that used to hide behind an environment variable is a named knob with the
// product default; tools/ and tests set them through the C ABI (g6d_set_knob / g6d_reset_knobs, include/gen6d_hip.h) to force a kernel
// variant or to sweep a model constant.  The library itself reads no environment variable.
enum G6dKnob {
  G6D_KNOB_CONV_PATCH, G6D_KNOB_TILE_POLICY, G6D_KNOB_SPLIT_TARGET, G6D_KNOB_PATCH_PIPE, G6D_KNOB_CORR_SLOTS, G6D_KNOB_SEL_ROWQ,
  G6D_KNOB_CONV1_MFMA, G6D_KNOB_W43_SPLIT_MAX, G6D_KNOB_W43_SPLIT_GAIN, G6D_KNOB_W43_CHUNK_US, G6D_KNOB_WINO_DEBUG, G6D_KNOB_CONV_WINO43,
  G6D_KNOB_WINO_WIDE, G6D_KNOB_WINO_SPLIT_MAX, G6D_KNOB_WINO_SPLIT_GAIN, G6D_KNOB_WINO_SPLIT_FIX, G6D_KNOB_WINO_SPLIT_PER, G6D_KNOB_WINO16_2W,
  G6D_KNOB_CONV_WINO, G6D_KNOB_CONV_WINO16, G6D_KNOB_WINO_MIN_WORK, G6D_KNOB_W43_MAP, G6D_KNOB_CONV_PM, G6D_KNOB_GEMV_MFMA, G6D_KNOB_C16_ABLATE, G6D_KNOB_CONV16_HALO, G6D_KNOB_CONV_NARROW, G6D_KNOB_COUNT
};
double g6d_knob(int id);

// Split launches (a tile's reduction spread over gridDim.z blocks) finish inside the kernel: every block writes its partial
// tile to the workspace, takes a ticket from the tile's counter, and the block that arrives last adds all partials (in
// split order, so the sum does not depend on which block that was) and runs the layer's epilogue.  No second kernel, and the
// partials are read back by the same lane mapping that wrote them (16-byte lane-linear rows).
// Visibility across CUs / XCDs (their L2s are not coherent with each other): the partials are stored write-through (sc1),
// every wave drains its stores, one lane takes the ticket with an agent-scope atomic, and the last block's lane 0 issues ONE
// agent-scope acquire (drops the CU's stale L1 lines) before the block reads the slabs with plain loads — the hand-off of
// MI355X_MICROARCH.md "inter-workgroup visibility"; no release fence (an L2 write-back per block) and no fence per thread.
// The counters are the first G6D_WS_COUNTERS ints of the caller's workspace: zero before the first launch, left zero by
// every launch (include/gen6d_hip.h, "Workspace").  `flag` is one free LDS word of the kernel's single LDS array.
#define G6D_WS_COUNTER_BYTES G6D_WORKSPACE_COUNTER_BYTES
#define G6D_WS_COUNTERS (G6D_WS_COUNTER_BYTES / 4)
__device__ __forceinline__ void g6d_store_wt(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ bool g6d_split_arrive(int* counter, int splits, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have been acknowledged
  __syncthreads();                                       // ... and every other wave's; the kernel's LDS is free from here on
  if (threadIdx.x == 0) {
    const int last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == splits - 1;
    if (last) {
      __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}

// InstanceNorm finalisation by the last block of a statistics-producing launch (G6dConv.fin_*): every block that has
// issued its (sum, sumsq) atomics drains them and takes a ticket; the block that draws the last one reads the completed
// table with agent-scope loads and writes the affine of the following InstanceNorm.  `nblocks` = blocks that run the
// epilogue (tiles; of a split launch only the tile finishers); `flag` = one free word of the kernel's LDS array.
struct G6dFin { float* scale; float* shift; int* counter; const double* stats; double inv_count, eps; int n; };
__device__ __forceinline__ void g6d_finalize_stats(const G6dFin& f, int nblocks, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's atomics have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const int last = __hip_atomic_fetch_add(f.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1;
    if (last) {
      __hip_atomic_store(f.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm: zero at rest, like the split counters
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    *flag = last;
  }
  __syncthreads();
  if (*flag == 0) return;
  for (int i = threadIdx.x; i < f.n; i += blockDim.x) {
    const double s1 = __hip_atomic_load(f.stats + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double s2 = __hip_atomic_load(f.stats + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double mean = s1 * f.inv_count;
    double var = s2 * f.inv_count - mean * mean;
    if (var < 0) var = 0;
    const double rs = 1.0 / sqrt(var + f.eps);
    f.scale[i] = (float)rs;
    f.shift[i] = (float)(-mean * rs);
  }
}
__device__ __forceinline__ G6dFin g6d_fin_of(const G6dConv& d) {
  return G6dFin{d.fin_scale, d.fin_shift, reinterpret_cast<int*>(d.fin_counter), d.stats, 1.0 / d.fin_count, d.fin_eps, d.fin_groups * d.Cout};
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return v > 0.f ? v : 0.1f * v;
  return v;
}
