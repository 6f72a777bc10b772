// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
//   out[m][co] = act(bias[co] + sum_{tap,ci} X(m,tap,ci) * W[co][tap][ci]),   m = (n,od,oh,ow) flattened
//
// GEMM view: M = N*Do*Ho*Wo output positions, N = Cout, K = taps*Cin.  Activations are channels-last, so the K axis
// of both operands is contiguous in memory (128-byte runs of 32 channels): a 256-thread workgroup stages a
// [BM x 32] activation tile and a [BN x 32] weight tile per K step through LDS with 16-byte loads, LDS rows padded to
// 36 floats so that the ds_read_b128 fragment reads (lane i -> row i) are bank-conflict free.
//
// Fragment trick: v_mfma_f32_32x32x2_f32 wants A[i][k], B[k][j] with k = lane>>5.  The K order inside an 8-wide
// chunk is free as long as A and B agree, so lane-half h reads the 4 consecutive k = 4h..4h+3 with ONE ds_read_b128
// per operand and feeds them to 4 consecutive MFMAs (MFMA s consumes k = s and k = 4+s).
//
// Pipeline: double-buffered LDS, register-staged global prefetch of K step t+1 issued before the MFMAs of step t,
// one barrier per K step.  The operand loader applies the fused prologue (elementwise multiplier, InstanceNorm
// affine, ReLU; zero padding stays exactly zero); the epilogue adds bias/activation and accumulates the
// InstanceNorm statistics of the output in fp64.  Small grids are split along K: partial tiles go to a workspace and the
// block of a tile that arrives last adds them and runs the epilogue (g6d_common.h).
#include "g6d_common.h"
#include <type_traits>
#include <stdlib.h>

#define LDS_K 36
#define BK 32
#ifndef G6D_ABLATE
#define G6D_ABLATE 0   // profiling builds only (tools/ablate.sh): 1 = no prefetch / LDS stores, 2 = MFMA only, 3 = no barrier,
                       // 4 = global loads but no LDS stores (loads get optimised away), 5 = LDS stores but no global loads,
                       // 6 = global loads consumed by a dummy add, no LDS stores, 7 = full kernel without the per-step barrier
#endif

// wino_conv.hip: eligible 3x3 / 3x3x3 stride-1 layers with pre-transformed filters (G6dConv.weight_wino) on the Winograd kernel
bool g6d_wino_eligible(const G6dConv& d);
int g6d_wino_launch(const G6dConv& d, hipStream_t stream);
// wino43_conv.hip: the F(4x4,3x3) kernel for layers that carry G6dConv.weight_wino43
bool g6d_wino43_eligible(const G6dConv& d);
int g6d_wino43_launch(const G6dConv& d, hipStream_t stream);
// conv_patch.hip: 3x3 / 3x3x3 stride-1 layers with Cout <= 64: spatial output tile, input patch reused by all taps
bool g6d_conv_patch_eligible(const G6dConv& d);
int g6d_conv_patch_launch(const G6dConv& d, int M, hipStream_t stream);

namespace {

// 16-byte load at base + (unsigned 32-bit element offset): scalar base + 32-bit vector offset addressing
__device__ __forceinline__ f32x4 ldg(const float* __restrict__ base, int elem_off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + ((unsigned)elem_off << 2));
}

// MODE: 0 = plain operand, 1 = affine(+ReLU) with one table, 2 = affine(+ReLU) with a table per image group (n / in_affine_per_n),
//       3 = elementwise multiplier + affine with one table (selector product), 4 = multiplier + a table per image group (the
//       selector product of a BATCH of queries: multiplier and table of query n / k, input image n % k — G6dConv.in_image_mod).
// MM: 0 = fp32 MFMA (default); 1 / 2 = bf16 / fp16 operands, fp32 accumulation (G6dConv.math_mode, g6d_common.h).
template <int BM, int BN, int WGM, int WGN, int MODE, int MM>
__global__ void __launch_bounds__(256, 2) conv_igemm_kernel(const G6dConv p, const int M, const int T,
                                                         const int nChunks, const int itersPerSplit,
                                                         const int totalIters, const int splits, const unsigned in_bytes,
                                                         const unsigned w_bytes, const unsigned mul_bytes, const int pm_hw,
                                                         const int pm_tiles, const int pm_ny_count) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int RA = BM / 32, RB = BN / 32;
  constexpr int STAGE = (BM + BN) * LDS_K;
  constexpr bool AFF = MODE != 0, PER_N = MODE == 2 || MODE == 4, MUL = MODE == 3 || MODE == 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int lrow = tid >> 3, lseg = tid & 7;

  // Row order of the GEMM.  Default: row = output position in memory order, tile = BM consecutive positions.  POSITION-MAJOR
  // (pm_hw = Ho*Wo > 0: small 2-D maps, stride 1, many images — the selector's 4x4 / 8x8 stacks over 2560 hypothesis images): a tile is
  // ONE output position (oy, ox) of BM consecutive images, so the taps that fall into the zero padding are the same for every row
  // of the tile and the K loop simply skips them — a 3x3 "same" conv on a 4x4 map has 100 live (position, tap) pairs of 144 (1.44x
  // fewer K steps), on an 8x8 map 484 of 576.  Block id (1-D grid) -> (image tile, channel tile, position): hardware hands consecutive
  // ids to the 8 XCDs in turn, so within a group of 8 image tiles the id runs (tile-in-group fastest, then channel tile, then position):
  // XCD k gets ALL channel tiles and positions of image tile 8g + k back to back — its 64 resident blocks are 16 positions x 4 channel
  // tiles of ONE image tile, whose 2 MB of input rows (each read by up to 9 positions x 4 channel tiles) stay in that XCD's 4 MB L2.
  // (First version: channel tiles in gridDim.y, i.e. four image tiles x 16 positions resident per XCD = 8 MB of input: the family's
  // HBM-side traffic went from 171 to 262 MB per launch, profiles/r05_pmc_conv_traffic.json.)
  int pm_pos = 0, pm_tile = blockIdx.x, pm_ny = blockIdx.y;
  if (pm_hw > 0) {
    const int per = 8 * pm_ny_count * pm_hw, g = blockIdx.x / per, r = blockIdx.x - g * per;
    const int m8 = min(8, pm_tiles - 8 * g);
    const int r2 = r / m8;
    pm_tile = 8 * g + (r - r2 * m8);
    pm_pos = r2 / pm_ny_count; pm_ny = r2 - pm_pos * pm_ny_count;
  }
  const int m0 = pm_tile * BM, n0 = pm_ny * BN;                 // (position-major: first IMAGE of the tile)
  auto row_m = [&](int r) { return pm_hw > 0 ? (m0 + r) * pm_hw + pm_pos : m0 + r; };      // tile row -> output position index m
  int ky0 = 0, ky1 = p.kh - 1, kx0 = 0, kx1 = p.kw - 1;        // live taps of the tile (position-major: the padding taps are cut)
  if (pm_hw > 0) {
    const int oy = pm_pos / p.Wo, ox = pm_pos - oy * p.Wo;
    ky0 = max(0, p.ph - oy); ky1 = min(p.kh - 1, p.Hi - 1 + p.ph - oy);
    kx0 = max(0, p.pw - ox); kx1 = min(p.kw - 1, p.Wi - 1 + p.pw - ox);
  }
  const int it_begin = blockIdx.z * itersPerSplit;
  const int it_end = pm_hw > 0 ? p.kd * (ky1 - ky0 + 1) * (kx1 - kx0 + 1) * nChunks : min(totalIters, it_begin + itersPerSplit);

  const int Cin = p.Cin, khw = p.kh * p.kw;
  const float* __restrict__ gsc = p.in_scale;
  const float* __restrict__ gsh = p.in_shift;
  const int relu = p.in_relu;

  // ---- per-thread rows of the activation tile: decode the output position once.  On this matrix pipe every vector-ALU
  //      instruction beside an fp32 MFMA is paid in full (tools/ubench/mfma_fill.hip), so the per-K-step work of the loader is
  //      reduced to a few bit operations: a row keeps three bitmasks — which kz / ky / kx taps fall inside the input for it —
  //      and ONE constant byte offset; the tap and channel-chunk offsets are uniform and go into the scalar offset of a
  //      bounds-checked buffer load; an invalid (padding / masked) row asks for an offset beyond the tensor and the hardware
  //      returns zeros.  The buffer base sits `pad` elements in front of the tensor so that the row offsets of the first
  //      output positions (virtual, in the zero padding) are not negative.
  const int pad_a = ((p.pd * p.Hi + p.ph) * p.Wi + p.pw) * p.ld_in;       // elements between the buffer base and p.in
  const int pad_m = (p.ph * p.Wi + p.pw) * Cin;
  unsigned amz[RA], amy[RA], amx[RA], avoff[RA], mvoff[MUL ? RA : 1];
  int nbase[PER_N ? RA : 1];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = row_m(lrow + 32 * j);
    amz[j] = amy[j] = amx[j] = 0u; avoff[j] = 0u;
    if constexpr (MUL) mvoff[j] = 0u;
    if constexpr (PER_N) nbase[j] = 0;
    if (m < M) {
      const int ow = m % p.Wo, t1 = m / p.Wo;
      const int oh = t1 % p.Ho, t2 = t1 / p.Ho;
      const int od = t2 % p.Do, n = t2 / p.Do;
      const int az0 = od * p.sd - p.pd, ay0 = oh * p.sh - p.ph, ax0 = ow * p.sw - p.pw;
      for (int k = 0; k < p.kd; ++k) amz[j] |= (unsigned)((unsigned)(az0 + k) < (unsigned)p.Di) << k;
      for (int k = 0; k < p.kh; ++k) amy[j] |= (unsigned)((unsigned)(ay0 + k) < (unsigned)p.Hi) << k;
      for (int k = 0; k < p.kw; ++k) amx[j] |= (unsigned)((unsigned)(ax0 + k) < (unsigned)p.Wi) << k;
      const int n_in = p.in_image_mod > 0 ? n % p.in_image_mod : n;          // query batches share the input images
      avoff[j] = (unsigned)((((n_in * p.Di + az0) * p.Hi + ay0) * p.Wi + ax0) * p.ld_in + pad_a + 4 * lseg) << 2;
      if constexpr (MUL) {
        const int mg = p.mul_group_images > 0 ? n / p.mul_group_images : 0;    // the image group's own multiplier map
        mvoff[j] = (unsigned)(((mg * p.Hi + ay0) * p.Wi + ax0) * Cin + pad_m + 4 * lseg) << 2;
      }
      if constexpr (PER_N) nbase[j] = (n / p.in_affine_per_n) * Cin;
    }
  }
  // ---- per-thread rows of the weight tile: constant byte offset, beyond the tensor for rows >= Cout
  unsigned bvoff[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int co = n0 + lrow + 32 * j;
    bvoff[j] = co < p.Cout ? (unsigned)(co * T * Cin + 4 * lseg) << 2 : 0x80000000u;
  }
  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) - pad_a, 0, in_bytes, 0x00020000);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.weight), 0, w_bytes, 0x00020000);
  const auto rs_mul = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MUL ? p.mul - pad_m : p.in), 0, mul_bytes, 0x00020000);
  // channels of the last (partial) chunk that exist for this thread: all of them unless Cin % 32 != 0
  const unsigned thr_last = 4 * lseg < Cin - 32 * (nChunks - 1) ? 0xffffffffu : 0u;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Two register sets: the global loads of K step t+2 are issued during step t and written to LDS during step t+1,
  // which gives every load ~1.6 K steps (>3000 cycles) to land while needing only two LDS stages.
  f32x4 ra[2][RA], rm[2][MUL ? RA : 1], rb[2][RB], rsc[2][PER_N ? RA : 1], rsh[2][PER_N ? RA : 1];
  unsigned va[2][AFF ? RA : 1];            // row validity masks (all ones / zero) of the staged tile, for the zeroing after the affine

  // K position of the tile being LOADED = (channel chunk cc, tap = (kz,ky,kx)); taps vary FASTEST so that consecutive
  // K steps read the same channel chunk at positions shifted by one tap: the shifted window is still in L1/L2, whereas a
  // chunk-fastest order brings it back only after nChunks tiles per resident block (~4 MB per XCD: L2 thrash -> MALL).
  int lt = it_begin;                       // index of the tile being loaded
  int cc = pm_hw > 0 ? 0 : it_begin / T;
  const int tap0 = it_begin - cc * T;
  int kz = pm_hw > 0 ? 0 : tap0 / khw, ky = pm_hw > 0 ? ky0 : (tap0 - kz * khw) / p.kw, kx = pm_hw > 0 ? kx0 : tap0 - kz * khw - ky * p.kw;
  auto advance = [&]() {
    lt += 1;
    kx += 1; const bool w1 = kx > kx1; kx = w1 ? kx0 : kx;
    ky += w1; const bool w2 = ky > ky1; ky = w2 ? ky0 : ky;
    kz += w2; const bool w3 = kz == p.kd; kz = w3 ? 0 : kz;
    cc += w3;
  };

  // All loads are unconditional so that they pipeline (a branch around a load makes hipcc wait for each one separately).
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto bload = [&](const auto& rs, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0)));
  };
  int toff = 0, moff = 0, cch = 0, woff = 0;      // uniform per-K-step byte offsets (scalar registers)
  unsigned cm = 0;                                // per-thread mask of the step: 0 beyond the K range / for missing channels
  bool cv = false;
  auto begin_step = [&](auto S) {
    constexpr int s = decltype(S)::value;
    const bool live = lt < it_end;
    cch = cc * BK + 4 * lseg;
    cv = (cch < Cin) & live;
    cm = live ? (cc == nChunks - 1 ? thr_last : 0xffffffffu) : 0u;
    toff = live ? (((kz * p.Hi + ky) * p.Wi + kx) * p.ld_in + cc * BK) << 2 : 0;
    moff = live ? ((ky * p.Wi + kx) * Cin + cc * BK) << 2 : 0;
    woff = live ? (((kz * p.kh + ky) * p.kw + kx) * Cin + cc * BK) << 2 : 0;
    if constexpr (AFF && !PER_N) { rsc[s][0] = ldg(gsc, cv ? cch : 0); rsh[s][0] = ldg(gsh, cv ? cch : 0); }
  };
  auto load_a = [&](auto S, int j) {
    constexpr int s = decltype(S)::value;
    // all ones if tap (kz, ky, kx) of this row lies inside the input and the step / channel is live
    const unsigned m = (unsigned)__builtin_amdgcn_sbfe(amz[j], kz, 1) & (unsigned)__builtin_amdgcn_sbfe(amy[j], ky, 1) &
                       (unsigned)__builtin_amdgcn_sbfe(amx[j], kx, 1) & cm;
    if constexpr (AFF) va[s][j] = m;
    ra[s][j] = bload(rs_in, (avoff[j] & m) | (~m & 0x80000000u), toff);
    if constexpr (MUL) rm[s][j] = bload(rs_mul, (mvoff[j] & m) | (~m & 0x80000000u), moff);
    if constexpr (PER_N) { rsc[s][j] = ldg(gsc, cv ? nbase[j] + cch : 0); rsh[s][j] = ldg(gsh, cv ? nbase[j] + cch : 0); }
  };
  auto load_b = [&](auto S, int j) {
    constexpr int s = decltype(S)::value;
    rb[s][j] = bload(rs_w, bvoff[j], woff);      // channels beyond Cin meet zero activations; rows beyond Cout are out of range
  };
  auto store_a = [&](auto S, float* As, int j) {
    constexpr int s = decltype(S)::value;
    f32x4 v = ra[s][j];
    if constexpr (MUL) v *= rm[s][j];
    if constexpr (AFF) {
      v = v * rsc[s][PER_N ? j : 0] + rsh[s][PER_N ? j : 0];
      if (relu) {         // ONE v_max per value (fmaxf() costs a second, canonicalising v_max)
#pragma unroll
        for (int e = 0; e < 4; ++e) asm("v_max_f32 %0, 0, %1" : "=v"(v[e]) : "v"(v[e]));
      }
      u32x4 b = __builtin_bit_cast(u32x4, v);       // zero padding stays exactly zero: it follows the norm in the reference
      b &= va[s][j];
      v = __builtin_bit_cast(f32x4, b);
    }
    *reinterpret_cast<f32x4*>(As + (lrow + 32 * j) * LDS_K + 4 * lseg) = v;
  };
  auto store_b = [&](auto S, float* Bs, int j) {
    constexpr int s = decltype(S)::value;
    *reinterpret_cast<f32x4*>(Bs + (lrow + 32 * j) * LDS_K + 4 * lseg) = rb[s][j];
  };

  f32x4 dummy = {0.f, 0.f, 0.f, 0.f};   // G6D_ABLATE == 6 only
  f32x4 fa[2][MT], fb[2][NT];
  auto read_frags = [&](const float* As, const float* Bs, int kc) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
      fa[kc & 1][i] = *reinterpret_cast<const f32x4*>(As + (wm * WM + i * 32 + li) * LDS_K + kc * 8 + 4 * lh);
#pragma unroll
    for (int j = 0; j < NT; ++j)
      fb[kc & 1][j] = *reinterpret_cast<const f32x4*>(Bs + (wn * WN + j * 32 + li) * LDS_K + kc * 8 + 4 * lh);
  };
  // MFMA number q of a K step, in (chunk kc, sub-step s, tile i, tile j) order.  A wave tile of ONE 32x32 accumulator (the
  // 64x64 and 128x32 block tiles of the small layers) alternates between two accumulators that are added in front of the
  // epilogue: instructions issued between two MFMAs on the same accumulator stretch the dependent pair.
  constexpr bool ONE_ACC = MT * NT == 1;
  f32x16 acc_odd;
  if constexpr (ONE_ACC) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_odd[r] = 0.f;
  }
  auto mfma_q = [&](int q) {
    const int kc = q / (4 * MT * NT), r = q % (4 * MT * NT);
    const int sidx = r / (MT * NT), i = (r % (MT * NT)) / NT, j = r % NT;
    if constexpr (ONE_ACC) {
      if (sidx & 1) { acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kc & 1][0][sidx], fb[kc & 1][0][sidx], acc_odd, 0, 0, 0); return; }
    }
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kc & 1][i][sidx], fb[kc & 1][j][sidx], acc[i][j], 0, 0, 0);
  };

  // One K step = NP "pieces" of 2 MFMAs, each followed by a slice of the non-matrix work (one global load of K step
  // t+2 early in the step, one LDS row store of K step t+1 late in the step, the fragment reads of the next chunk),
  // with a scheduling barrier between pieces: v_mfma_f32_32x32x2_f32 holds the pipe for 64 cycles, i.e. ~14 issue
  // slots per MFMA are free for other instructions, so a single wave can keep the matrix pipe fed.
  constexpr int NMFMA = MT * NT * 16, NP = NMFMA / 2, QPC = NMFMA / 4 / 2;   // QPC = pieces per 8-wide chunk
  constexpr int NROW = RA + RB;
  static_assert(NP >= NROW, "tile too small for the piece schedule");
  // PAR = parity of the tile being computed: loads go to register set PAR (tile t+2), stores come from set PAR^1.
  auto k_step = [&](auto PAR, const float* As, const float* Bs, float* An, float* Bn) {
    constexpr int par = decltype(PAR)::value;
    using SL = std::integral_constant<int, par>;
    using SS = std::integral_constant<int, par ^ 1>;
    if constexpr (MM != 0) {
      // reduced precision: a K step is 2 * MT * NT MFMAs of K = 16 (two 8-channel fragment slices each); the step is bound by
      // the loads, so they are simply issued first, the stores last
#pragma unroll
      for (int j = 0; j < RA; ++j) load_a(SL{}, j);
#pragma unroll
      for (int j = 0; j < RB; ++j) load_b(SL{}, j);
#pragma unroll
      for (int kp = 0; kp < 2; ++kp) {
        f32x4 a0[MT], a1[MT], b0[NT], b1[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          a0[i] = *reinterpret_cast<const f32x4*>(As + (wm * WM + i * 32 + li) * LDS_K + (2 * kp) * 8 + 4 * lh);
          a1[i] = *reinterpret_cast<const f32x4*>(As + (wm * WM + i * 32 + li) * LDS_K + (2 * kp + 1) * 8 + 4 * lh);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          b0[j] = *reinterpret_cast<const f32x4*>(Bs + (wn * WN + j * 32 + li) * LDS_K + (2 * kp) * 8 + 4 * lh);
          b1[j] = *reinterpret_cast<const f32x4*>(Bs + (wn * WN + j * 32 + li) * LDS_K + (2 * kp + 1) * 8 + 4 * lh);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = g6d_mfma_lowp<MM>(a0[i], a1[i], b0[j], b1[j], acc[i][j]);
      }
#pragma unroll
      for (int j = 0; j < RA; ++j) store_a(SS{}, An, j);
#pragma unroll
      for (int j = 0; j < RB; ++j) store_b(SS{}, Bn, j);
      advance(); begin_step(SS{});
      return;
    }
    read_frags(As, Bs, 0);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
      if (pc % QPC == 0 && pc / QPC < 3 && (G6D_ABLATE < 2 || G6D_ABLATE >= 6)) read_frags(As, Bs, pc / QPC + 1);
      mfma_q(2 * pc); mfma_q(2 * pc + 1);
      if (G6D_ABLATE < 1 || G6D_ABLATE >= 6) {
        if (pc < RA) load_a(SL{}, pc); else if (pc < NROW) load_b(SL{}, pc - RA);
        const int sr = pc - (NP - NROW);
        if (sr >= 0) {
          if (G6D_ABLATE == 6) {          // consume the loads without touching LDS
            if (sr < RA) dummy += ra[par ^ 1][sr]; else dummy += rb[par ^ 1][sr - RA];
          } else if (sr < RA) store_a(SS{}, An, sr); else store_b(SS{}, Bn, sr - RA);
        }
        // the K position / offsets of the NEXT step's loads are advanced here, behind the last MFMAs of this step
        // and in front of the barrier, instead of ahead of the first MFMA of the next step
        if (pc == NP - 1) { advance(); begin_step(SS{}); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (it_begin < it_end) {
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    begin_step(S0{});                                   // tile 0 -> set 0 -> LDS stage 0
#pragma unroll
    for (int j = 0; j < RA; ++j) load_a(S0{}, j);
#pragma unroll
    for (int j = 0; j < RB; ++j) load_b(S0{}, j);
    advance(); begin_step(S1{});                        // tile 1 -> set 1 (stored during K step 0)
#pragma unroll
    for (int j = 0; j < RA; ++j) load_a(S1{}, j);
#pragma unroll
    for (int j = 0; j < RB; ++j) load_b(S1{}, j);
#pragma unroll
    for (int j = 0; j < RA; ++j) store_a(S0{}, lds, j);
#pragma unroll
    for (int j = 0; j < RB; ++j) store_b(S0{}, lds + BM * LDS_K, j);
    advance(); begin_step(S0{});                        // tile 2 -> set 0, loaded during K step 0
    __syncthreads();
    float* L0 = lds; float* L1 = lds + STAGE;
    for (int it = it_begin; it < it_end; it += 2) {
      k_step(S0{}, L0, L0 + BM * LDS_K, L1, L1 + BM * LDS_K);
      if (G6D_ABLATE < 3 || G6D_ABLATE == 6) __syncthreads();   // 7: no barrier
      if (it + 1 < it_end) {
        k_step(S1{}, L1, L1 + BM * LDS_K, L0, L0 + BM * LDS_K);
        if (G6D_ABLATE < 3 || G6D_ABLATE == 6) __syncthreads();   // 7: no barrier
      }
    }
    if (G6D_ABLATE == 6) acc[0][0][0] += dummy[0] + dummy[1] + dummy[2] + dummy[3];
  }

  // ---------------------------------------------------------------- epilogue
  if constexpr (ONE_ACC) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += acc_odd[r];
  }
  const int Cout = p.Cout;
  if (splits > 1) {
    // partial tile -> workspace as [split][tile][(i, j, r/4)][thread] 16-byte pieces; the block that arrives last adds them in
    // split order (ZU splits in flight: the partials of other XCDs come from HBM / Infinity Cache, ~1.5 us away) and goes on
    // to the epilogue below
    constexpr int TILE = BM * BN;                              // = MT * NT * 16 floats x 256 threads
    constexpr int PIECES = MT * NT * 4;
    constexpr int ZU = 16 / PIECES;                          // 16 pieces in flight: no more registers than the K loop needs
    const int ntiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x;
    float* part = p.workspace + G6D_WS_COUNTERS + (size_t)tile * TILE + tid * 4;
    const size_t zstride = (size_t)ntiles * TILE;
    float* mine = part + blockIdx.z * zstride;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          g6d_store_wt(mine + ((i * NT + j) * 4 + q) * 1024,
                       f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]});
    if (!g6d_split_arrive(reinterpret_cast<int*>(p.workspace) + tile, splits, reinterpret_cast<int*>(lds))) return;
    __syncthreads();                                           // the flag word is read; lds is reused below
    f32x4 sum[PIECES];
#pragma unroll
    for (int k = 0; k < PIECES; ++k) sum[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < splits; z0 += ZU) {
      f32x4 v[ZU][PIECES];
#pragma unroll
      for (int u = 0; u < ZU; ++u) {
        const float* src = part + (size_t)min(z0 + u, splits - 1) * zstride;
#pragma unroll
        for (int k = 0; k < PIECES; ++k) v[u][k] = *reinterpret_cast<const f32x4*>(src + k * 1024);
      }
#pragma unroll
      for (int u = 0; u < ZU; ++u)
        if (z0 + u < splits) {
#pragma unroll
          for (int k = 0; k < PIECES; ++k) sum[k] += v[u][k];
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t = sum[(i * NT + j) * 4 + q];
          acc[i][j][4 * q] = t[0]; acc[i][j][4 * q + 1] = t[1]; acc[i][j][4 * q + 2] = t[2]; acc[i][j][4 * q + 3] = t[3];
        }
  }

  const bool do_stats = p.stats != nullptr;
  const int rpg = p.stat_rows_per_group;
  const int mlast = pm_hw > 0 ? min(row_m(BM - 1), M - pm_hw + pm_pos) : min(m0 + BM, M) - 1;      // last valid row of the tile
  const int g0 = rpg > 0 ? row_m(0) / rpg : 0;
  const bool one_group = rpg <= 0 || (mlast / rpg) == g0;
  float* sred = lds;   // [BN][2], reused after the K loop (all waves passed the final barrier)
  if (do_stats && one_group) {
    for (int i = tid; i < BN * 2; i += 256) sred[i] = 0.f;
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * WN + j * 32 + li;
    const bool cval = col < Cout;
    const float bv = (p.bias && cval) ? p.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_m(wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
        float v = apply_act(acc[i][j][r] + bv, p.out_act);
        if (row < M && cval) {
          p.out[(size_t)row * p.ld_out + col] = v;
          if (do_stats) {
            if (one_group) { s1 += v; s2 += v * v; }
            else {
              double* st = p.stats + ((size_t)(row / rpg) * Cout + col) * 2;
              atomicAdd(st, (double)v); atomicAdd(st + 1, (double)v * v);
            }
          }
        }
      }
    }
    if (do_stats && one_group) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lh == 0) {
        atomicAdd(&sred[(wn * WN + j * 32 + li) * 2], s1);
        atomicAdd(&sred[(wn * WN + j * 32 + li) * 2 + 1], s2);
      }
    }
  }
  if (do_stats && one_group) {
    __syncthreads();
    if (tid < BN && n0 + tid < Cout) {
      double* st = p.stats + ((size_t)g0 * Cout + n0 + tid) * 2;
      atomicAdd(st, (double)sred[tid * 2]);
      atomicAdd(st + 1, (double)sred[tid * 2 + 1]);
    }
  }
  if (do_stats && p.fin_scale) {
    __syncthreads();                                     // sred (= lds) is no longer read
    g6d_finalize_stats(g6d_fin_of(p), gridDim.x * gridDim.y, reinterpret_cast<int*>(lds));
  }
}

// Position-major row order (see the kernel): 2-D stride-1 "same" layers on maps of at most 8x8 with at least 4 tiles of images, un-split.
// Knob conv_pm: 1 = where it measured faster — the layers with an InstanceNorm prologue (64-channel N tiles; the selector's 4x4 stacks:
// 480 -> 465 us and 970 -> 945 us per batch of 8).  A position tile reads every input row ONCE (the row-order tile re-reads its 16 KB
// of rows for each of the 9 taps out of L1), so the K loop gets 1.44x shorter while its operand stream moves from L1 to L2: the plain
// 128x128-tile layer (4x4x128 -> 256) lost 8 % and keeps the row order; 2 = every eligible layer (tests).
bool igemm_position_major(const G6dConv& d, int bm, int splits) {
  const int pm = (int)g6d_knob(G6D_KNOB_CONV_PM);
  return pm != 0 && (pm == 2 || d.in_scale != nullptr) && splits == 1 && d.Di == 1 && d.Do == 1 && d.kd == 1 && d.sh == 1 && d.sw == 1 && d.Ho == d.Hi &&
         d.Wo == d.Wi && d.Ho * d.Wo <= 64 && d.Ho * d.Wo > 1 && (d.kh > 1 || d.kw > 1) && d.kh == 2 * d.ph + 1 && d.kw == 2 * d.pw + 1 &&
         d.N >= 4 * bm;
}

template <int BM, int BN, int WGM, int WGN, int MODE, int MM>
int launch_mm(const G6dConv& d, int M, int T, int nChunks, int splits, hipStream_t stream) {
  const int total = T * nChunks;
  const int ips = (total + splits - 1) / splits;
  splits = (total + ips - 1) / ips;
  const bool pm = igemm_position_major(d, BM, splits);
  const int pm_hw = pm ? d.Ho * d.Wo : 0, pm_tiles = pm ? (d.N + BM - 1) / BM : 0;
  const int ny = (d.Cout + BN - 1) / BN;
  dim3 grid(pm ? pm_hw * pm_tiles * ny : (M + BM - 1) / BM, pm ? 1 : ny, splits);
  const size_t lds_bytes = 2 * (size_t)(BM + BN) * LDS_K * sizeof(float);
  g6d_allow_lds(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WGM, WGN, MODE, MM>), (int)lds_bytes);
  // extents of the buffer-load descriptors (the activation and multiplier descriptors start `pad` elements in front of the tensor)
  const long long pad_a = ((long long)(d.pd * d.Hi + d.ph) * d.Wi + d.pw) * d.ld_in, pad_m = (long long)(d.ph * d.Wi + d.pw) * d.Cin;
  const long long n_in = d.in_image_mod > 0 ? d.in_image_mod : d.N;
  const long long n_mul = d.mul_group_images > 0 ? (d.N + d.mul_group_images - 1) / d.mul_group_images : 1;
  const unsigned in_bytes = (unsigned)((n_in * d.Di * d.Hi * d.Wi * d.ld_in + pad_a) * 4);
  const unsigned w_bytes = (unsigned)((long long)d.Cout * T * d.Cin * 4);
  const unsigned mul_bytes = (unsigned)((n_mul * d.Hi * d.Wi * d.Cin + pad_m) * 4);
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, MODE, MM>), grid, dim3(256), lds_bytes, stream, d, M, T, nChunks,
                     ips, total, splits, in_bytes, w_bytes, mul_bytes, pm_hw, pm_tiles, ny);
  return g6d_check_launch("conv_igemm");
}

template <int BM, int BN, int WGM, int WGN, int MODE>
int launch_mode(const G6dConv& d, int M, int T, int nChunks, int splits, hipStream_t stream) {
  if (d.math_mode == 1) return launch_mm<BM, BN, WGM, WGN, MODE, 1>(d, M, T, nChunks, splits, stream);
  if (d.math_mode == 2) return launch_mm<BM, BN, WGM, WGN, MODE, 2>(d, M, T, nChunks, splits, stream);
  return launch_mm<BM, BN, WGM, WGN, MODE, 0>(d, M, T, nChunks, splits, stream);
}

template <int BM, int BN, int WGM, int WGN>
int launch_cfg(const G6dConv& d, int M, int T, int nChunks, int splits, hipStream_t stream) {
  if (d.mul) return d.in_affine_per_n ? launch_mode<BM, BN, WGM, WGN, 4>(d, M, T, nChunks, splits, stream)
                                      : launch_mode<BM, BN, WGM, WGN, 3>(d, M, T, nChunks, splits, stream);
  if (!d.in_scale) return launch_mode<BM, BN, WGM, WGN, 0>(d, M, T, nChunks, splits, stream);
  if (d.in_affine_per_n) return launch_mode<BM, BN, WGM, WGN, 2>(d, M, T, nChunks, splits, stream);
  return launch_mode<BM, BN, WGM, WGN, 1>(d, M, T, nChunks, splits, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_narrow_kernel: 3x3 layers with at most FOUR output channels (the detector heads' merged last conv 192 -> 4, reference
// network/detector.py:164-184: 116 us per batch of 8 at 4.6 TFLOP/s on a 128 x 32 matrix-core tile, 7/8 of whose columns are padding —
// VERDICT r05 weak #6).  A layer like that is a dot product per pixel, bound by reading its input: here a wave walks a run of 8
// consecutive pixels of one row, lane l holds input channels 4l .. 4l+3 of all 36 filter rows in registers, keeps the 3 x 3 window of
// 16-byte input pieces in registers and loads only the window's new column per pixel (3 coalesced row pieces instead of 9), and the four
// sums are reduced over the lanes with butterfly shuffles.  fp32 on the vector ALUs in every math mode.
constexpr int NARROW_RUN = 8;
__global__ void __launch_bounds__(256) conv_narrow_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int NH, int H, int W, int Cin, int ld_in, int Cout, int ld_out,
                                                         int act, int runs_per_row, int nruns) {
  const int lane = threadIdx.x & 63;
  const int c4 = lane * 4;
  const bool on = c4 < Cin;
  f32x4 wr[4][9];
#pragma unroll
  for (int co = 0; co < 4; ++co)
#pragma unroll
    for (int t = 0; t < 9; ++t)
      wr[co][t] = (on && co < Cout) ? *reinterpret_cast<const f32x4*>(w + ((long)co * 9 + t) * Cin + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nwaves = (gridDim.x * blockDim.x) >> 6;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int run = wave; run < nruns; run += nwaves) {
    const int g = run / runs_per_row, x0 = (run - g * runs_per_row) * NARROW_RUN;      // g = row of the tall image [N*H]
    const int y = g % H;
    const float* rowp[3];
    bool rv[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      rv[ky] = yy >= 0 && yy < H;
      rowp[ky] = in + ((long)(g + ky - 1) * W) * ld_in + c4;
    }
    auto column = [&](int x, f32x4 (&col)[3]) {
      const bool xv = x >= 0 && x < W;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) col[ky] = (on && xv && rv[ky]) ? *reinterpret_cast<const f32x4*>(rowp[ky] + (long)x * ld_in) : zero;
    };
    f32x4 win[3][3];                                             // [kx][ky]
    column(x0 - 1, win[0]);
    column(x0, win[1]);
#pragma unroll
    for (int i = 0; i < NARROW_RUN; ++i) {
      const int x = x0 + i;
      column(x + 1, win[2]);
      float acc[4];
#pragma unroll
      for (int co = 0; co < 4; ++co) {
        f32x4 a = zero;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) a += win[kx][ky] * wr[co][3 * ky + kx];
        acc[co] = (a[0] + a[1]) + (a[2] + a[3]);
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int co = 0; co < 4; ++co) acc[co] += __shfl_xor(acc[co], off, 64);
      if (lane == 0 && x < W) {
#pragma unroll
        for (int co = 0; co < 4; ++co)
          if (co < Cout) {
            float v = acc[co] + (bias ? bias[co] : 0.f);
            if (act == 1) v = fmaxf(v, 0.f);
            else if (act == 2) v = v > 0.f ? v : 0.1f * v;
            out[((long)g * W + x) * ld_out + co] = v;
          }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) { win[0][ky] = win[1][ky]; win[1][ky] = win[2][ky]; }
    }
  }
}

bool conv_narrow_eligible(const G6dConv& d) {
  return g6d_knob(G6D_KNOB_CONV_NARROW) != 0 && d.Cout <= 4 && d.Cin <= 256 && d.kd == 1 && d.kh == 3 && d.kw == 3 && d.sd == 1 && d.sh == 1 && d.sw == 1 &&
         d.pd == 0 && d.ph == 1 && d.pw == 1 && d.Di == 1 && d.Do == 1 && d.Ho == d.Hi && d.Wo == d.Wi && !d.mul && !d.in_scale && !d.stats &&
         !d.in_image_mod && d.out_act >= 0 && d.out_act <= 2;
}

int conv_narrow_launch(const G6dConv& d, hipStream_t stream) {
  const int rpr = (d.Wi + NARROW_RUN - 1) / NARROW_RUN;
  const long nruns = (long)d.N * d.Hi * rpr;
  if (nruns >= (1l << 31)) { g6d_set_error("conv (narrow): too many pixel runs"); return G6D_EINVAL; }
  const int blocks = (int)((nruns + 3) / 4 < 2048 ? (nruns + 3) / 4 : 2048);
  hipLaunchKernelGGL(conv_narrow_kernel, dim3(blocks), dim3(256), 0, stream, d.in, d.weight, d.bias, d.out, d.N * d.Hi, d.Hi, d.Wi, d.Cin, d.ld_in, d.Cout,
                     d.ld_out, d.out_act, rpr, (int)nruns);
  return g6d_check_launch("conv_narrow");
}

}  // namespace

// Which kernel family g6d_conv_igemm will run this descriptor on: 0 generic implicit GEMM, 1 LDS-patch kernel, 2 Winograd kernel,
// 3 F(4x4,3x3) kernel, 4 the narrow-output kernel on the vector ALUs
// (no launch; bench.py uses it to book the executed FLOPs of a launch in the right roofline family).
extern "C" int g6d_conv_plan(const G6dConv* desc) {
  if (!desc) return G6D_EINVAL;
  if (conv_narrow_eligible(*desc)) return 4;
  if (g6d_wino43_eligible(*desc)) return 3;
  if (g6d_wino_eligible(*desc)) return 2;
  const bool use_patch = g6d_knob(G6D_KNOB_CONV_PATCH) != 0;
  return (use_patch && g6d_conv_patch_eligible(*desc)) ? 1 : 0;
}

extern "C" int g6d_conv_igemm(const G6dConv* desc, g6d_stream_t stream_) {
  if (!desc) return G6D_EINVAL;
  const G6dConv& d = *desc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!d.in || !d.weight || !d.out) { g6d_set_error("conv: null pointer"); return G6D_EINVAL; }
  if (d.math_mode < 0 || d.math_mode > 2) { g6d_set_error("conv: math_mode must be 0 (fp32), 1 (bf16) or 2 (fp16)"); return G6D_EINVAL; }
  if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.Do <= 0 || d.Ho <= 0 || d.Wo <= 0 || d.kd <= 0 || d.kh <= 0 || d.kw <= 0) {
    g6d_set_error("conv: bad shape"); return G6D_EINVAL;
  }
  if ((d.Cin & 3) || (d.ld_in & 3) || d.ld_in < d.Cin || d.ld_out < d.Cout) { g6d_set_error("conv: Cin/ld_in must be multiples of 4"); return G6D_EINVAL; }
  if (!g6d_aligned16(d.in) || !g6d_aligned16(d.weight) || (d.mul && !g6d_aligned16(d.mul)) ||
      (d.in_scale && (!g6d_aligned16(d.in_scale) || !d.in_shift || !g6d_aligned16(d.in_shift)))) {
    g6d_set_error("conv: operand pointers must be 16-byte aligned"); return G6D_EINVAL;
  }
  if (d.mul && !d.in_scale) { g6d_set_error("conv: mul requires in_scale/in_shift"); return G6D_EINVAL; }
  if (d.in_affine_per_n < 0 || d.in_image_mod < 0 || d.mul_group_images < 0 || (d.mul_group_images > 0 && !d.mul) ||
      d.in_image_mod > d.N) {
    g6d_set_error("conv: in_affine_per_n / in_image_mod / mul_group_images must be >= 0 (mul_group_images needs mul; in_image_mod <= N)");
    return G6D_EINVAL;
  }
  if ((long long)(d.mul_group_images > 0 ? (d.N + d.mul_group_images - 1) / d.mul_group_images : 1) * d.Hi * d.Wi * d.Cin >= (1ll << 29)) {
    g6d_set_error("conv: multiplier tensor exceeds 2^31 bytes"); return G6D_EINVAL;
  }
  if (d.fin_scale && (!d.stats || !d.fin_shift || !d.fin_counter || d.fin_count <= 0 || d.fin_groups <= 0)) {
    g6d_set_error("conv: fin_scale needs stats, fin_shift, fin_counter, fin_count > 0 and fin_groups"); return G6D_EINVAL;
  }
  if ((long long)(d.in_image_mod > 0 ? d.in_image_mod : d.N) * d.Di * d.Hi * d.Wi * d.ld_in + ((long long)(d.pd * d.Hi + d.ph) * d.Wi + d.pw) * d.ld_in >= (1ll << 29) ||
      (long long)d.Cout * d.kd * d.kh * d.kw * d.Cin >= (1ll << 29) || d.kd > 32 || d.kh > 32 || d.kw > 32) {
    g6d_set_error("conv: tensor exceeds 2^31 bytes (buffer-load offsets) or kernel extent > 32"); return G6D_EINVAL;
  }
  const long long Mll = (long long)d.N * d.Do * d.Ho * d.Wo;
  if (Mll > (1ll << 30)) { g6d_set_error("conv: M too large"); return G6D_EINVAL; }
  const int M = (int)Mll;
  if (conv_narrow_eligible(d)) return conv_narrow_launch(d, stream);
  if (g6d_wino43_eligible(d)) return g6d_wino43_launch(d, stream);
  if (g6d_wino_eligible(d)) return g6d_wino_launch(d, stream);
  const bool use_patch = g6d_knob(G6D_KNOB_CONV_PATCH) != 0;
  if (use_patch && g6d_conv_patch_eligible(d)) return g6d_conv_patch_launch(d, M, stream);
  const int T = d.kd * d.kh * d.kw;
  const int nChunks = (d.Cin + BK - 1) / BK;
  const int total = T * nChunks;

  // tile configuration
  int bm = 128, bn = d.Cout <= 32 ? 32 : (d.Cout <= 64 ? 64 : 128);
  if (bn == 128 && (d.mul || (d.in_scale && d.in_affine_per_n))) bn = 64;   // 128x128 with two register sets + per-row tables would spill
  if (M <= 64 && d.Cout > 32) { bm = 64; bn = 64; }
  // Grids that do not fill the 256 CUs with 128-row tiles (measured per layer, profiles/r01_conv_microbench.md):
  //  * if 64x64 tiles give >= 256 blocks on their own, take them and skip split-K and its reduce launch altogether;
  //  * mid-size M (>= 2048 rows): 64x64 tiles need 4x fewer splits, i.e. 4x less partial-sum traffic;
  //  * small M (<= 1024 rows) with wide N: 128x64 tiles halve the split count at the same block count.
  const bool tile_policy = g6d_knob(G6D_KNOB_TILE_POLICY) != 0;
  bool no_split = false;
  if (tile_policy && d.Cout > 32 && bm == 128 && d.split_k <= 0) {
    const long long blocks128 = (long long)((M + 127) / 128) * ((d.Cout + bn - 1) / bn);
    const long long blocks64 = (long long)((M + 63) / 64) * ((d.Cout + 63) / 64);
    if (blocks128 < 256) {
      if (blocks64 >= 256) { bm = 64; bn = 64; no_split = true; }
      else if (M >= 2048) { bm = 64; bn = 64; }
      else if (M <= 1024 && bn == 128) bn = 64;
    }
  }
  const long long blocks = (long long)((M + bm - 1) / bm) * ((d.Cout + bn - 1) / bn);

  int splits = d.split_k;
  if (splits <= 0) {
    splits = 1;
    const int split_target = (int)g6d_knob(G6D_KNOB_SPLIT_TARGET);
    if (blocks < (split_target < 256 ? split_target : 256) && total >= 8 && !no_split) {
      splits = (int)((split_target + blocks - 1) / blocks);
      if (splits > total / 4) splits = total / 4;
      if (splits > 64) splits = 64;
      if (splits < 1) splits = 1;
    }
  }
  if (splits > total) splits = total;
  if (splits > 1) {     // workspace = tile counters + [split][tile] partial tiles (g6d_common.h)
    const size_t per = (size_t)blocks * bm * bn * sizeof(float);
    const size_t room = d.workspace && d.workspace_bytes > G6D_WS_COUNTER_BYTES ? d.workspace_bytes - G6D_WS_COUNTER_BYTES : 0;
    if (room < (size_t)splits * per || blocks > G6D_WS_COUNTERS) {
      if (d.split_k > 1) { g6d_set_error("conv: workspace too small for forced split_k"); return G6D_ENOSPC; }
      splits = blocks > G6D_WS_COUNTERS ? 1 : (int)(room / per);
      if (splits < 2) splits = 1;
    }
  }
  if (bm == 64) return launch_cfg<64, 64, 2, 2>(d, M, T, nChunks, splits, stream);
  if (bn == 32) return launch_cfg<128, 32, 4, 1>(d, M, T, nChunks, splits, stream);
  if (bn == 64) return launch_cfg<128, 64, 2, 2>(d, M, T, nChunks, splits, stream);
  return launch_cfg<128, 128, 2, 2>(d, M, T, nChunks, splits, stream);
}
