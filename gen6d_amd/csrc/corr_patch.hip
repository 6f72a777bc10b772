// Detector correlation with input-patch reuse in LDS (network/detector.py:222-224, the 15x15 level).
//
// The generic implicit-GEMM kernel re-loads a [128 x 32] activation tile from L2 for every tap, which for a skinny
// GEMM (N = rfn = 32 output channels) needs 5 global loads per 16 MFMAs per wave: measured, that load path — not the
// matrix pipe — bounds it (66-71 of ~120 attainable TFLOP/s).  Consecutive taps along kx read the same
// input rows shifted by one pixel, so here a block keeps an input PATCH of 8 rows x (32 + kw - 1) columns x 32
// channels in LDS per (channel chunk, ky) and walks the kw taps over it: 1 global load per 16 MFMAs.
//
//   block  = 512 threads = 8 waves; output tile = 8 rows x 32 columns (256 positions) x 32 output channels;
//            wave w owns output row w of the tile: lane i -> column i, one 32x32 accumulator tile
//   unit   = (channel chunk of 32, ky): patch stage [8][32+kw-1][36 floats] (single buffer); per kx one weight tile
//            [32 co][36] double-buffered; 16 MFMAs (v_mfma_f32_32x32x2_f32) per wave per kx
//   split  = units are split across gridDim.z; partial tiles go to a workspace, the last block of a tile adds them (g6d_common.h).
#include "g6d_common.h"
#include <stdlib.h>
#include <type_traits>

#define LDS_K 36
#define TH 8
#define TW 32

namespace {

__device__ __forceinline__ f32x4 ldg4(const float* __restrict__ base, int elem_off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + ((unsigned)elem_off << 2));
}

// One map of a launch.  A launch may cover up to CORR_MAX_SEG maps (the scales of the detector's image pyramid are correlated with
// the same reference filters): their tiles form ONE flat list, so the launch fills the chip with fewer splits than four separate
// launches.  Offsets in floats from the launch's common base pointers.
#define CORR_MAX_SEG 4
struct CorrSeg { int tile0, H, W, tiles_x, in_off, out_off, ld_in, ld_out, N, tiles_img; };   // N images of H x W, tiles_img tiles each
struct CorrArgs { int nseg; CorrSeg seg[CORR_MAX_SEG]; };

template <int MM>     // 0 = fp32 MFMA; 1 / 2 = bf16 / fp16 operands (two 8-channel groups per v_mfma_f32_32x32x16_*)
__global__ void __launch_bounds__(512) corr_patch_kernel(const float* __restrict__ in_base, const float* __restrict__ wgt,
                                                         float* __restrict__ out_base, const CorrArgs sa, int Cin,
                                                         int Cout, int kh, int kw, int ph, int pw,
                                                         int units_per_split, int total_units, int splits,
                                                         float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int sidx = 0;
#pragma unroll
  for (int k = 1; k < CORR_MAX_SEG; ++k) sidx = (k < sa.nseg && (int)blockIdx.x >= sa.seg[k].tile0) ? k : sidx;
  const CorrSeg& sg = sa.seg[sidx];
  const int H = sg.H, W = sg.W, tiles_x = sg.tiles_x, ld_in = sg.ld_in, ld_out = sg.ld_out;
  const int tile_s = blockIdx.x - sg.tile0, img = tile_s / sg.tiles_img;       // image of the segment (a batch of queries at one scale)
  const float* __restrict__ in = in_base + sg.in_off + (size_t)img * H * W * ld_in;
  float* __restrict__ out = out_base + sg.out_off + (size_t)img * H * W * ld_out;
  const int tile = tile_s - img * sg.tiles_img;
  const int PW = TW + kw - 1;                       // patch width in positions
  const int patch_floats = TH * PW * LDS_K;
  // one patch buffer: the next unit's patch waits in registers during the kw taps and is written between two barriers
  // after the last tap (58 KB per block at kw = 15 instead of 111 KB, so that a second block — possibly of another
  // stream's kernel — fits on the CU)
  float* patch = lds;
  float* bt0 = lds + patch_floats;                  // weight tiles [32][LDS_K] x 2
  float* bt1 = bt0 + 32 * LDS_K;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
  const int u_begin = blockIdx.z * units_per_split, u_end = min(total_units, u_begin + units_per_split);
  const int T = kh * kw;

  // patch loader: 16-byte segment `seg` of position p (p = tid/8 + 64*j), j < NPL
  const int seg = tid & 7;
  const int npos = TH * PW;
  const int NPL = (npos + 63) / 64;                 // <= 8 for kw <= 31
  f32x4 rp[8];
  bool vp[8];
  f32x4 rbw; bool vbw = false;                      // weight tile load (threads 0..255: row = tid/8)
  const int brow = tid >> 3;

  auto unit_ck = [&](int u, int& chunk, int& ky) { chunk = u / kh; ky = u - chunk * kh; };

  auto load_patch = [&](int u) {
    int chunk, ky; unit_ck(u, chunk, ky);
    const bool uv = u < u_end;
    const int c = chunk * 32 + 4 * seg;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < NPL) {
        const int p = (tid >> 3) + 64 * j;
        const int pr = p / PW, pc = p - pr * PW;
        const int iy = ty0 + pr + ky - ph, ix = tx0 + pc - pw;
        const bool v = uv & (p < npos) & (c < Cin) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
        vp[j] = v;
        rp[j] = ldg4(in, v ? (iy * W + ix) * ld_in + c : 0);
      }
    }
  };
  auto store_patch = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < NPL) {
        const int p = (tid >> 3) + 64 * j;
        if (p < npos) *reinterpret_cast<f32x4*>(dst + p * LDS_K + 4 * seg) = vp[j] ? rp[j] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto load_b = [&](int u, int kx) {
    int chunk, ky; unit_ck(u, chunk, ky);
    const int c = chunk * 32 + 4 * seg;
    vbw = (tid < 256) & (u < u_end) & (brow < Cout) & (c < Cin);
    rbw = ldg4(wgt, vbw ? (brow * T + ky * kw + kx) * Cin + c : 0);
  };
  auto store_b = [&](float* dst) {
    if (tid < 256) *reinterpret_cast<f32x4*>(dst + brow * LDS_K + 4 * seg) = vbw ? rbw : f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // two accumulators, alternating per MFMA and added at the end: a wave owns ONE 32x32 output tile, and instructions issued
  // between two MFMAs on the same accumulator (fragment requests, barriers) stretch the dependent pair
  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }

  if (u_begin < u_end) {
    load_patch(u_begin); load_b(u_begin, 0);
    store_patch(patch); store_b(bt0);
    __syncthreads();
    int bcur = 0;
    // Software-pipelined tap loop (same scheme as conv_patch.hip): a tap = 4 groups (8-channel slices) of 4 MFMAs; the
    // fragments of group kc+1 are requested in front of the MFMAs of group kc, and the last group of every tap is
    // deferred across the tap's barrier with its operands in registers (set 1), so that the first fragments of the next
    // weight tile arrive in the shadow of those MFMAs instead of in front of an idle matrix pipe.
    f32x4 fa[2], fb[2];
    fa[0] = fa[1] = fb[0] = fb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mm = [&](int set) {
      if constexpr (MM == 0) {
#pragma unroll
        for (int s2 = 0; s2 < 4; s2 += 2) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][s2], fb[set][s2], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][s2 + 1], fb[set][s2 + 1], acc2, 0, 0, 0);
        }
      } else if (set == 1) {
        // reduced precision: one K = 16 MFMA per two 8-channel groups.  It runs whenever set 1 is due and consumes whatever
        // the two sets hold at that point — (group 0 of this tap, deferred group 3 of the previous tap) at the top of a tap,
        // (group 2, group 1) in its middle: every slot multiplies matching A and B entries and every group is consumed exactly
        // once, so the sum is the same.
        acc = g6d_mfma_lowp<MM>(fa[0], fa[1], fb[0], fb[1], acc);
      }
    };
    for (int u = u_begin; u < u_end; ++u) {
      const float* P = patch;
      load_patch(u + 1);                            // masked beyond u_end; lands during the kw steps below
      for (int kx = 0; kx < kw; ++kx) {
        const float* B = bcur ? bt1 : bt0;
        float* Bn = bcur ? bt0 : bt1;
        const bool last = kx == kw - 1;
        const float* arow = P + (wave * PW + li + kx) * LDS_K + 4 * lh;
        const float* brow_p = B + li * LDS_K + 4 * lh;
        auto rd = [&](int set, int kc) {
          fa[set] = *reinterpret_cast<const f32x4*>(arow + kc * 8);
          fb[set] = *reinterpret_cast<const f32x4*>(brow_p + kc * 8);
        };
        rd(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(1);                                      // deferred group 3 of the previous tap (zeros at the very start)
        load_b(last ? u + 1 : u, last ? 0 : kx + 1);
        __builtin_amdgcn_sched_barrier(0);
        rd(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(0);
        __builtin_amdgcn_sched_barrier(0);
        rd(0, 2);
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
        __builtin_amdgcn_sched_barrier(0);
        rd(1, 3);                                   // stays in set 1 across the barrier
        __builtin_amdgcn_sched_barrier(0);
        mm(0);
        __builtin_amdgcn_sched_barrier(0);
        if (last && u + 1 < u_end) {
          __syncthreads();                          // every wave holds its last fragments of this unit's patch in registers
          store_patch(patch);
        }
        store_b(Bn);
        __syncthreads();
        bcur ^= 1;
      }
    }
    if constexpr (MM != 0) fa[0] = fb[0] = f32x4{0.f, 0.f, 0.f, 0.f};      // group 2 of the last tap is already in the sum
    mm(1);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];

  // split launches: the partial tile goes to the workspace lane-linearly and the block that arrives last adds them up
  // (g6d_common.h)
  if (splits > 1) {
    constexpr int TILE = TH * TW * 32;                       // 512 threads x 16 floats
    float* part = ws + G6D_WS_COUNTERS + (size_t)blockIdx.x * TILE + tid * 4;
    const size_t zstride = (size_t)gridDim.x * TILE;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      g6d_store_wt(part + blockIdx.z * zstride + q * 2048, f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
    if (!g6d_split_arrive(reinterpret_cast<int*>(ws) + blockIdx.x, splits, reinterpret_cast<int*>(lds))) return;
    f32x4 sum[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sum[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < splits; z0 += 4) {      // 4 splits in flight: stays inside the 128 registers of two blocks per CU
      f32x4 v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[u][q] = *reinterpret_cast<const f32x4*>(part + (size_t)min(z0 + u, splits - 1) * zstride + q * 2048);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (z0 + u < splits) {
#pragma unroll
          for (int q = 0; q < 4; ++q) sum[q] += v[u][q];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[4 * q] = sum[q][0]; acc[4 * q + 1] = sum[q][1]; acc[4 * q + 2] = sum[q][2]; acc[4 * q + 3] = sum[q][3]; }
  }
  // epilogue: acc rows = output columns tx0 + (r&3) + 8*(r>>2) + 4*lh of image row ty0 + wave; acc column = co = li
  const int oy = ty0 + wave;
  if (oy < H && li < Cout) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ox = tx0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ox < W) out[((size_t)oy * W + ox) * ld_out + li] = acc[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same correlation with 16-bit matrix-core operands (math_mode 1 / 2) as its own kernel.  corr_patch_kernel<1|2> keeps the fp32
// structure — one barrier and one weight-tile hand-over per TAP — and that is ~1050 cycles per tap for 64 cycles of 16-bit MFMA.
// Here a unit (32 channels, ky) stages ALL kw weight tiles at once: the filters come host-rounded and laid out per unit
// ([unit][kx][32 co][40 halfs]: 32 channels + 8 halfs of padding, 80-byte rows are bank-conflict free for the fragment reads) and
// go straight to LDS with direct-to-LDS loads (double buffered); the patch is rounded to 16 bits when it is written to LDS
// ([8 rows][32 + kw - 1 columns][40 halfs], single buffer, the next unit's pieces wait in registers); a wave then runs its
// 2 kw MFMAs (v_mfma_f32_32x32x16_*) of the unit with one ds_read_b128 per operand and no barrier in between.
typedef _Float16 c16h4 __attribute__((ext_vector_type(4)));
typedef __bf16 c16b4 __attribute__((ext_vector_type(4)));
#define C16_ROW 80                                   // bytes of a (position | tap, co) row: 32 channels x 2 bytes + 16 bytes of padding

template <int MM>
__global__ void __launch_bounds__(512) corr16_patch_kernel(const float* __restrict__ in_base, const char* __restrict__ w16,
                                                           float* __restrict__ out_base, const CorrArgs sa, int Cin, int Cout, int kh,
                                                           int kw, int ph, int pw, int units_per_split, int total_units, int splits,
                                                           float* __restrict__ ws, int wbuf_bytes) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using hv4 = typename std::conditional<MM == 1, c16b4, c16h4>::type;
  using hv8 = typename std::conditional<MM == 1, bf16x8, f16x8>::type;
  int sidx = 0;
#pragma unroll
  for (int k = 1; k < CORR_MAX_SEG; ++k) sidx = (k < sa.nseg && (int)blockIdx.x >= sa.seg[k].tile0) ? k : sidx;
  const CorrSeg& sg = sa.seg[sidx];
  const int H = sg.H, W = sg.W, tiles_x = sg.tiles_x, ld_in = sg.ld_in, ld_out = sg.ld_out;
  const int tile_s = blockIdx.x - sg.tile0, img = tile_s / sg.tiles_img;
  const float* __restrict__ in = in_base + sg.in_off + (size_t)img * H * W * ld_in;
  float* __restrict__ out = out_base + sg.out_off + (size_t)img * H * W * ld_out;
  const int tile = tile_s - img * sg.tiles_img;
  const int PW = TW + kw - 1;
  const int npos = TH * PW;
  char* const ldsb = reinterpret_cast<char*>(lds);
  const int patch_bytes = (npos * C16_ROW + 1023) & ~1023;
  const int unit_bytes = kw * 32 * C16_ROW;          // one unit of the filter tensor; wbuf_bytes = that rounded up to whole 1 KB pieces

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
  const int u_begin = blockIdx.z * units_per_split, u_end = min(total_units, u_begin + units_per_split);

  // patch loader: 16-byte segment `seg` (4 channels) of position p = tid/8 + 64 j
  const int seg = tid & 7;
  const int NPL = (npos + 63) / 64;                 // <= 6 for kw <= 15
  f32x4 rp[6];
  bool vp[6];
  auto load_patch = [&](int u) {
    const int chunk = u / kh, ky = u - chunk * kh;
    const bool uv = u < u_end;
    const int c = chunk * 32 + 4 * seg;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (j < NPL) {
        const int p = (tid >> 3) + 64 * j;
        const int pr = p / PW, pc = p - pr * PW;
        const int iy = ty0 + pr + ky - ph, ix = tx0 + pc - pw;
        const bool v = uv & (p < npos) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
        vp[j] = v;
        rp[j] = ldg4(in, v ? (iy * W + ix) * ld_in + c : 0);
      }
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (j < NPL) {
        const int p = (tid >> 3) + 64 * j;
        const f32x4 v = vp[j] ? rp[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (p < npos) *reinterpret_cast<hv4*>(__builtin_assume_aligned(ldsb + p * C16_ROW + 8 * seg, 8)) = __builtin_convertvector(v, hv4);
      }
    }
  };
  // filter tiles of a unit: wbuf_bytes / 1024 pieces of 1 KB, wave w moves pieces w, w + 8, ...
  const unsigned lds_addr0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  const unsigned lane16 = lane * 16;
  auto load_w = [&](int u, int buf) {
    const char* src = w16 + (size_t)u * unit_bytes;
    const int npieces = wbuf_bytes >> 10;
    for (int i = wave; i < npieces; i += 8) {
      const char* g = src + (size_t)i * 1024;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr0 + (unsigned)(patch_bytes + buf * wbuf_bytes + i * 1024));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(lane16), "s"(g), "s"(dst) : "memory");
    }
  };

  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }

  if (u_begin < u_end) {
    load_w(u_begin, 0);
    load_patch(u_begin);
    store_patch();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int u = u_begin; u < u_end; ++u) {
      if (u + 1 < u_end) load_w(u + 1, cur ^ 1);     // (the copies first: older than the loads the compiler counts)
      load_patch(u + 1);                            // masked beyond u_end
      const char* A = ldsb + (wave * PW + li) * C16_ROW + 16 * lh;
      const char* B = ldsb + patch_bytes + cur * wbuf_bytes + li * C16_ROW + 16 * lh;
      // fragments one tap ahead of their MFMAs (a runtime tap count: the compiler would otherwise wait for every read in front of its use)
      auto rd = [&](int kx, hv8& a0, hv8& b0, hv8& a1, hv8& b1) {
        a0 = *reinterpret_cast<const hv8*>(__builtin_assume_aligned(A + kx * C16_ROW, 16));
        b0 = *reinterpret_cast<const hv8*>(__builtin_assume_aligned(B + kx * 32 * C16_ROW, 16));
        a1 = *reinterpret_cast<const hv8*>(__builtin_assume_aligned(A + kx * C16_ROW + 32, 16));
        b1 = *reinterpret_cast<const hv8*>(__builtin_assume_aligned(B + kx * 32 * C16_ROW + 32, 16));
      };
      hv8 a0, b0, a1, b1;
      rd(0, a0, b0, a1, b1);
      for (int kx = 0; kx < kw; ++kx) {
        hv8 n0, m0, n1, m1;
        rd(min(kx + 1, kw - 1), n0, m0, n1, m1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MM == 1) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc2, 0, 0, 0);
        } else {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc2, 0, 0, 0);
        }
        a0 = n0; b0 = m0; a1 = n1; b1 = m1;
      }
      __syncthreads();                              // every wave has read its fragments of this unit's patch
      if (u + 1 < u_end) store_patch();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next unit's filter tiles have landed
      __syncthreads();
      cur ^= 1;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];

  if (splits > 1) {
    constexpr int TILE = TH * TW * 32;
    float* part = ws + G6D_WS_COUNTERS + (size_t)blockIdx.x * TILE + tid * 4;
    const size_t zstride = (size_t)gridDim.x * TILE;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      g6d_store_wt(part + blockIdx.z * zstride + q * 2048, f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
    if (!g6d_split_arrive(reinterpret_cast<int*>(ws) + blockIdx.x, splits, reinterpret_cast<int*>(lds))) return;
    f32x4 sum[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sum[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < splits; ++z) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sum[q] += *reinterpret_cast<const f32x4*>(part + (size_t)z * zstride + q * 2048);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[4 * q] = sum[q][0]; acc[4 * q + 1] = sum[q][1]; acc[4 * q + 2] = sum[q][2]; acc[4 * q + 3] = sum[q][3]; }
  }
  const int oy = ty0 + wave;
  if (oy < H && li < Cout) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ox = tx0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ox < W) out[((size_t)oy * W + ox) * ld_out + li] = acc[r];
    }
  }
}

}  // namespace

namespace {

int corr_run(const float* in_base, float* out_base, CorrArgs& sa, int Cin, const float* wgt, int Cout, int kh, int kw, float* workspace,
             size_t workspace_bytes, int math_mode, hipStream_t stream, const void* w16 = nullptr) {
  int tiles = 0;
  for (int k = 0; k < sa.nseg; ++k) {
    CorrSeg& g = sa.seg[k];
    g.tiles_x = (g.W + TW - 1) / TW;
    g.tile0 = tiles;
    g.tiles_img = g.tiles_x * ((g.H + TH - 1) / TH);
    tiles += g.N * g.tiles_img;
  }
  const int total_units = ((Cin + 31) / 32) * kh;
  // split the (chunk, ky) units so that the grid fills whole rounds of the chip (256 CUs x 2 resident blocks):
  // pick the split count with the best last-round utilisation, preferring fewer splits on ties
  int splits = 1;
  {
    const size_t per = (size_t)tiles * TH * TW * 32 * sizeof(float);       // tile-padded partials (>= H*W*Cout)
    const size_t room = workspace && workspace_bytes > G6D_WS_COUNTER_BYTES ? workspace_bytes - G6D_WS_COUNTER_BYTES : 0;
    int max_s = total_units / 2;
    if (max_s > 64) max_s = 64;
    if ((size_t)max_s > room / per) max_s = (int)(room / per);
    if (max_s < 1 || tiles > G6D_WS_COUNTERS) max_s = 1;
    const int slots2 = (int)g6d_knob(G6D_KNOB_CORR_SLOTS);   // two 58 KB blocks per CU
    const int slots = w16 ? 256 : slots2;                  // (the 16-bit kernel's block holds 100+ KB of LDS: one per CU)
    double best = -1.0;
    for (int sp = 1; sp <= max_s; ++sp) {
      const int ups_ = (total_units + sp - 1) / sp;
      const int real = (total_units + ups_ - 1) / ups_;
      const long long blocks = (long long)tiles * real;
      const double util = (double)blocks / ((double)slots * ((blocks + slots - 1) / slots));
      if (util > best + 0.02) { best = util; splits = real; }
    }
  }
  const int ups = (total_units + splits - 1) / splits;
  splits = (total_units + ups - 1) / ups;
  if (w16) {
    const int wbuf = (kw * 32 * C16_ROW + 1023) & ~1023;
    const size_t lds16 = (size_t)((TH * (TW + kw - 1) * C16_ROW + 1023) & ~1023) + 2 * (size_t)wbuf;
    auto go16 = [&](auto V) {
      constexpr int MM = decltype(V)::value;
      g6d_allow_lds(reinterpret_cast<const void*>(&corr16_patch_kernel<MM>), 160 * 1024);
      hipLaunchKernelGGL(corr16_patch_kernel<MM>, dim3(tiles, 1, splits), dim3(512), lds16, stream, in_base, reinterpret_cast<const char*>(w16),
                         out_base, sa, Cin, Cout, kh, kw, kh / 2, kw / 2, ups, total_units, splits, workspace, wbuf);
    };
    if (math_mode == 1) go16(std::integral_constant<int, 1>{}); else go16(std::integral_constant<int, 2>{});
    return g6d_check_launch("corr2d_patch16");
  }
  const size_t lds_bytes = (size_t)(TH * (TW + kw - 1) * LDS_K + 2 * 32 * LDS_K) * sizeof(float);
  auto go = [&](auto V) {
    constexpr int MM = decltype(V)::value;
    g6d_allow_lds(reinterpret_cast<const void*>(&corr_patch_kernel<MM>), 160 * 1024);      // the patch size depends on kw
    hipLaunchKernelGGL(corr_patch_kernel<MM>, dim3(tiles, 1, splits), dim3(512), lds_bytes, stream, in_base, wgt, out_base, sa, Cin,
                       Cout, kh, kw, kh / 2, kw / 2, ups, total_units, splits, workspace);
  };
  if (math_mode == 1) go(std::integral_constant<int, 1>{});
  else if (math_mode == 2) go(std::integral_constant<int, 2>{});
  else go(std::integral_constant<int, 0>{});
  return g6d_check_launch("corr2d_patch");
}

}  // namespace

// Stride-1 2-D cross-correlation without bias for Cout <= 32 (the detector's reference-as-filter correlation).
//   in  [H][W][ld_in] channels-last, wgt [Cout][kh*kw][Cin], out [H*W][ld_out]; zero padding (ph, pw) with
//   H_out = H, W_out = W (i.e. 2*ph = kh-1, 2*pw = kw-1).  workspace: split scratch (include/gen6d_hip.h, "Workspace").
extern "C" int g6d_corr2d_patch(const float* in, int H, int W, int Cin, int ld_in, const float* wgt, int Cout, int kh,
                                int kw, float* out, int ld_out, float* workspace, size_t workspace_bytes, int math_mode,
                                g6d_stream_t stream_) {
  if (!in || !wgt || !out || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || (ld_in & 3) || ld_in < Cin || Cout <= 0 ||
      Cout > 32 || ld_out < Cout || !(kh & 1) || !(kw & 1) || kw > 31 || !g6d_aligned16(in) || !g6d_aligned16(wgt) ||
      (long long)H * W * ld_in >= (1ll << 30) || (long long)Cout * kh * kw * Cin >= (1ll << 30)) {
    g6d_set_error("corr2d_patch: bad args (Cout <= 32, odd kernel <= 31, Cin % 4 == 0)"); return G6D_EINVAL;
  }
  if (math_mode < 0 || math_mode > 2) { g6d_set_error("corr2d_patch: math_mode must be 0, 1 or 2"); return G6D_EINVAL; }
  CorrArgs sa = {};
  sa.nseg = 1;
  sa.seg[0] = CorrSeg{0, H, W, 0, 0, 0, ld_in, ld_out, 1, 0};
  return corr_run(in, out, sa, Cin, wgt, Cout, kh, kw, workspace, workspace_bytes, math_mode, reinterpret_cast<hipStream_t>(stream_));
}

// The same correlation for up to 4 maps in ONE launch (G6dCorrSeg, include/gen6d_hip.h): the scales of the detector's image
// pyramid against the same reference filters (network/detector.py:236-241 around 222-224).
extern "C" int g6d_corr2d_patch_multi(const G6dCorrSeg* segs, int nseg, int Cin, const float* wgt, int Cout, int kh, int kw,
                                      float* workspace, size_t workspace_bytes, int math_mode, g6d_stream_t stream_) {
  if (!segs || nseg < 1 || nseg > CORR_MAX_SEG || !wgt || Cin <= 0 || (Cin & 3) || Cout <= 0 || Cout > 32 || !(kh & 1) || !(kw & 1) ||
      kw > 31 || !g6d_aligned16(wgt) || (long long)Cout * kh * kw * Cin >= (1ll << 30) || math_mode < 0 || math_mode > 2) {
    g6d_set_error("corr2d_patch_multi: bad args (1..4 maps, Cout <= 32, odd kernel <= 31, Cin % 4 == 0)"); return G6D_EINVAL;
  }
  const float* in0 = segs[0].in; float* out0 = segs[0].out;
  for (int k = 0; k < nseg; ++k) {
    const G6dCorrSeg& g = segs[k];
    if (!g.in || !g.out || g.H <= 0 || g.W <= 0 || g.N <= 0 || (g.ld_in & 3) || g.ld_in < Cin || g.ld_out < Cout || !g6d_aligned16(g.in)) {
      g6d_set_error("corr2d_patch_multi: bad map"); return G6D_EINVAL;
    }
    if (g.in < in0) in0 = g.in;
    if (g.out < out0) out0 = g.out;
  }
  CorrArgs sa = {};
  sa.nseg = nseg;
  for (int k = 0; k < nseg; ++k) {
    const G6dCorrSeg& g = segs[k];
    const long long io = g.in - in0, oo = g.out - out0;
    if (io + (long long)g.N * g.H * g.W * g.ld_in >= (1ll << 30) || oo + (long long)g.N * g.H * g.W * g.ld_out >= (1ll << 31)) {
      g6d_set_error("corr2d_patch_multi: maps must lie within 2^30 floats of each other (allocate them from one buffer)"); return G6D_EINVAL;
    }
    sa.seg[k] = CorrSeg{0, g.H, g.W, 0, (int)io, (int)oo, g.ld_in, g.ld_out, g.N, 0};
  }
  return corr_run(in0, out0, sa, Cin, wgt, Cout, kh, kw, workspace, workspace_bytes, math_mode, reinterpret_cast<hipStream_t>(stream_));
}

// g6d_corr2d_patch_multi with 16-bit matrix-core operands on its own kernel (corr16_patch_kernel): w16 = the filters rounded to the
// operand type of math_mode (1 = bf16, 2 = fp16) on the host, per unit: [(Cin/32) * kh units][kw][32 co][40 x 16 bit] (32 channels of
// chunk u / kh at tap row u % kh, 8 x 16 bit of zero padding per row; rows co >= Cout zero), followed by >= 1 KB of padding.
extern "C" int g6d_corr2d_patch16_multi(const G6dCorrSeg* segs, int nseg, int Cin, const void* w16, int Cout, int kh, int kw,
                                        float* workspace, size_t workspace_bytes, int math_mode, g6d_stream_t stream_) {
  if (!segs || nseg < 1 || nseg > CORR_MAX_SEG || !w16 || Cin <= 0 || (Cin & 31) || Cout <= 0 || Cout > 32 || !(kh & 1) || !(kw & 1) ||
      kw > 15 || !g6d_aligned16(w16) || (math_mode != 1 && math_mode != 2)) {
    g6d_set_error("corr2d_patch16_multi: bad args (1..4 maps, Cin % 32 == 0, Cout <= 32, odd kernel <= 15, math_mode 1 or 2)"); return G6D_EINVAL;
  }
  const float* in0 = segs[0].in; float* out0 = segs[0].out;
  for (int k = 0; k < nseg; ++k) {
    const G6dCorrSeg& g = segs[k];
    if (!g.in || !g.out || g.H <= 0 || g.W <= 0 || g.N <= 0 || (g.ld_in & 3) || g.ld_in < Cin || g.ld_out < Cout || !g6d_aligned16(g.in)) {
      g6d_set_error("corr2d_patch16_multi: bad map"); return G6D_EINVAL;
    }
    if (g.in < in0) in0 = g.in;
    if (g.out < out0) out0 = g.out;
  }
  CorrArgs sa = {};
  sa.nseg = nseg;
  for (int k = 0; k < nseg; ++k) {
    const G6dCorrSeg& g = segs[k];
    const long long io = g.in - in0, oo = g.out - out0;
    if (io + (long long)g.N * g.H * g.W * g.ld_in >= (1ll << 30) || oo + (long long)g.N * g.H * g.W * g.ld_out >= (1ll << 31)) {
      g6d_set_error("corr2d_patch16_multi: maps must lie within 2^30 floats of each other (allocate them from one buffer)"); return G6D_EINVAL;
    }
    sa.seg[k] = CorrSeg{0, g.H, g.W, 0, (int)io, (int)oo, g.ld_in, g.ld_out, g.N, 0};
  }
  return corr_run(in0, out0, sa, Cin, nullptr, Cout, kh, kw, workspace, workspace_bytes, math_mode, reinterpret_cast<hipStream_t>(stream_), w16);
}
