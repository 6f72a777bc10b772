// Error bookkeeping and ABI version for libgen6d_hip.
#include "g6d_common.h"
#include <string.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <mutex>
#include <set>
#include <utility>

static thread_local char g_err[256] = "";

void g6d_set_error(const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

int g6d_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return G6D_OK;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return G6D_ELAUNCH;
}

// Raise a kernel's dynamic-LDS limit once per (kernel, device): the attribute belongs to the device's copy of the code object,
// so a process that drives several GPUs has to set it on each of them.
void g6d_allow_lds(const void* func, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  if (done.insert({func, dev}).second) (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// ---- launch-policy knobs (g6d_common.h): name, product default
namespace {
struct KnobDef { const char* name; double def; };
const KnobDef kKnobs[G6D_KNOB_COUNT] = {
    {"conv_patch", 1},        // conv family: eligible layers on conv_patch_kernel (0: conv_igemm only)
    {"tile_policy", 1},       // conv_igemm: 64x64 / 128x64 tile choice by block count (0: fixed 128-row tiles)
    {"split_target", 512},    // conv_igemm: blocks a split launch aims for
    {"patch_pipe", 1},        // conv_patch: pipelined variant (0: the round-2 loop)
    {"corr_slots", 512},      // corr_patch: resident blocks assumed by the split model (two 58 KB blocks per CU)
    {"sel_rowq", -1},         // selector_levels: query rows in registers: -1 batches only, 0 never, 1 always
    {"conv1_mfma", 1},        // first trunk layer on the matrix cores (0: vector-pipe kernel)
    {"w43_split_max", 32}, {"w43_split_gain", 0.85}, {"w43_chunk_us", 2.8},      // F(4x4,3x3) split model
    {"wino_debug", 0},        // 1: print the chosen split of every Winograd launch to stderr
    {"conv_wino43", 1},       // conv family: layers that carry weight_wino43 on the F(4x4,3x3) kernel
    {"wino_wide", 1},         // F(2x2,3x3) trunk: 128-channel blocks where the model says so (0 never, 2 whenever eligible)
    {"wino_split_max", 32}, {"wino_split_gain", 0.85}, {"wino_split_fix", 2.0}, {"wino_split_per", 0.6},    // F(2x2,3x3) split model
    {"wino16_2w", 1},         // 16-bit trunk: un-split launches on the two-waves-per-SIMD kernel (0: one-wave kernel only)
    {"conv_wino", 1},         // conv family: eligible layers on the F(2x2,3x3) kernel
    {"conv_wino16", 1},       // ... and on its 16-bit variant in the reduced-precision mode
    {"wino_min_work", -1},    // Winograd profitability rule: -1 = built-in thresholds, 0 = off, > 0 = minimum M*K*Cout
    {"w43_map", 2},           // F(4x4,3x3): block id -> (pixel tile, channel slice): 2 slices fastest (a tile's slices on neighbouring XCDs: the
                              // best THROUGHPUT with three batches in flight, +1.2 % over 1), 1 a tile's slices on one XCD (the best serialised
                              // time), 0 grid order — profiles/r05_w43_experiments.md
    {"conv_pm", 1},           // conv_igemm: position-major tiles for small 2-D maps with many images (padding taps skipped): 1 layers with an
                              // InstanceNorm prologue (where it measured faster), 2 every eligible layer, 0 row order
    {"gemv_mfma", 1},         // linear layers on the matrix cores: 1 for 17..32 right-hand sides (measured rule), 2 from 2 on, 0 never
    {"c16_ablate", 0},        // conv16_direct, timing experiments only (WRONG results): bit 0 = no activation DMA after the first steps, bit 1 = no filter requests after them
    {"conv16_halo", 1},       // conv16_direct with fragment-major filters, 2-D layers: the halo-patch kernel (0: the per-tap kernel conv16r)
    {"conv_narrow", 1},       // conv_igemm: 3x3 layers with <= 4 output channels on the vector-ALU dot-product kernel (0: a matrix-core tile)
};
// Process-global table, filled when the library is loaded (static initialisation, before any entry point can run); reads and writes
// are relaxed atomics, so a g6d_set_knob racing with launches on other threads is a benign race on one value (a launch sees the
// old or the new policy).  Tools and tests set knobs BEFORE the launches they want to steer.
std::atomic<double> g_knob[G6D_KNOB_COUNT];
void knob_defaults() { for (int i = 0; i < G6D_KNOB_COUNT; ++i) g_knob[i].store(kKnobs[i].def, std::memory_order_relaxed); }
struct KnobInit { KnobInit() { knob_defaults(); } } g_knob_init;
}  // namespace
double g6d_knob(int id) { return g_knob[id].load(std::memory_order_relaxed); }
extern "C" int g6d_set_knob(const char* name, double value) {
  for (int i = 0; name && i < G6D_KNOB_COUNT; ++i)
    if (!strcmp(name, kKnobs[i].name)) { g_knob[i].store(value, std::memory_order_relaxed); return G6D_OK; }
  g6d_set_error("set_knob: unknown knob"); return G6D_EINVAL;
}
extern "C" double g6d_get_knob(const char* name) {
  for (int i = 0; name && i < G6D_KNOB_COUNT; ++i)
    if (!strcmp(name, kKnobs[i].name)) return g_knob[i].load(std::memory_order_relaxed);
  return -1e300;
}
extern "C" void g6d_reset_knobs(void) { knob_defaults(); }

extern "C" int g6d_abi_version(void) { return 12; }
extern "C" const char* g6d_last_error(void) { return g_err; }
extern "C" int g6d_sizeof_conv_desc(void) { return (int)sizeof(G6dConv); }

// Profiling aid: an empty kernel that brackets a region of interest in a rocprofv3 kernel trace
// (tools/rocpd_stats.py keeps only the dispatches between the first and the last g6d_marker_kernel).
__global__ void g6d_marker_kernel(int id) { (void)id; }

extern "C" int g6d_marker(int id, g6d_stream_t stream) {
  hipLaunchKernelGGL(g6d_marker_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), id);
  return g6d_check_launch("g6d_marker");
}

// Stream-ordered zero fill of device memory (the per-query InstanceNorm statistics arena, up to its high-water mark).  An own kernel, not
// hipMemsetAsync: captured into a hipGraph, the runtime's memset NODE did not clear a 2-4 MB extent (batches of 16: the replayed rows
// were garbage while the eager launches and the smaller graphs of the tests were right — profiles/r05_notes.md), a kernel node does.
__global__ void g6d_zero_kernel(uint4* p, size_t n16, unsigned char* tail, int ntail) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = uint4{0u, 0u, 0u, 0u};
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
extern "C" int g6d_zero_bytes(void* ptr, size_t bytes, g6d_stream_t stream) {
  if ((!ptr && bytes) || (reinterpret_cast<uintptr_t>(ptr) & 15)) { g6d_set_error("zero_bytes: null or unaligned pointer (16 bytes)"); return G6D_EINVAL; }
  if (bytes == 0) return G6D_OK;
  const size_t n16 = bytes / 16;
  const size_t blocks = (n16 + 255) / 256;
  hipLaunchKernelGGL(g6d_zero_kernel, dim3((unsigned)(blocks < 4096 ? (blocks ? blocks : 1) : 4096)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<uint4*>(ptr), n16, reinterpret_cast<unsigned char*>(ptr) + n16 * 16, (int)(bytes - n16 * 16));
  return g6d_check_launch("g6d_zero_bytes");
}
