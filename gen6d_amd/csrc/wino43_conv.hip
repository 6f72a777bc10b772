// 3x3 stride-1 pad-1 convolution as Winograd F(4x4,3x3) on the fp32 matrix cores of gfx950: 36 multiplications per 16 outputs —
// 4x fewer than the direct form, 1.78x fewer than the F(2x2,3x3) kernel of wino_conv.hip.  Used where the parity budget has
// room for the larger transform constants: the detector's image-pyramid trunk (reference
// network/pretrain_models.py:9-31, network/detector.py:188-197), its 15x15 reference-as-filter correlation as 5x5 blocks of 3x3
// (network/detector.py:222-224) and the refiner's 32^3 volume layers (network/refiner.py:88-143).
//
//   Y(4x4) = A^T [ sum_ci U_ci (.) V_ci ] A,   U = G g G^T (host, fp64, once per checkpoint),   V = B^T d B (6x6 input tile)
//
// Interpolation points (0, +-3/4, +-3/2, inf) instead of the textbook (0, +-1, +-2, inf): same operation count (every step is an
// FMA with a constant either way), 3.6x smaller fp32 error (measured: 1.3e-6 of the output range at Cin = 512 against 4.6e-6;
// F(2x2,3x3): 2.5e-7).  With a = 3/4, b = 3/2 the 1-D transforms are
//   B^T d : r0 = a^2b^2 d0 - (a^2+b^2) d2 + d4          A^T m : y0 = m0 + (m1+m2) + (m3+m4)
//           r1,r2 = (d4 - b^2 d2) +- a (d3 - b^2 d1)            y1 = a (m1-m2) + b (m3-m4)
//           r3,r4 = (d4 - a^2 d2) +- b (d3 - a^2 d1)            y2 = a^2 (m1+m2) + b^2 (m3+m4)
//           r5 = a^2b^2 d1 - (a^2+b^2) d3 + d5                  y3 = a^3 (m1-m2) + b^3 (m3-m4) + m5
// and G = diag(1/N_i) [1 p_i p_i^2] (row inf = [0 0 1]), N_i = prod_{k != i} (p_i - p_k)  (backbone.winograd43_filters).
//
// Mapping to v_mfma_f32_16x16x4_f32 (M = tiles, N = co, K = ci), designed around two facts measured on this chip:
// vector-ALU instructions never overlap with fp32 MFMAs, and 36 accumulator tiles of 32 tiles x 32 channels (576 registers) do not
// fit one wave.
//   block   = 256 threads = 4 waves = 8 "quarters" (8x8 output pixels = 2x2 Winograd tiles, the same flat quarter list as
//             wino_conv.hip) x 16*NT output channels (NT = 4: 64 channels; NT = 2: 32, the correlation's 32 references).
//   wave    = (pair pr, half hh): pair pr owns 16 tiles (4 quarters); the two waves of a pair split the 36 transform positions
//             by ROW: wave hh holds rows a = 3hh..3hh+2, all 6 columns b, for all 16*NT channels: 18 x NT accumulators of 4
//             registers (288 at NT = 4).  A 16-row A operand against a 64-column B operand halves the input-transform work per
//             MFMA compared with 32 x 32 tiles: a lane transforms ONE tile (2 channels) for 18 x NT x 2 MFMAs.
//             The row split costs nothing in the transform: r0..r2 need 6 operations, r3..r5 need 6 (72 packed FMAs per wave
//             and 8-channel chunk against 144 MFMAs of 32 cycles).
//   K loop  = chunks of 8 input channels, two phases each.  The filter tile of a chunk (36 positions x 64 channels x 8 = 72 KB)
//             does not fit LDS twice, so it is staged in halves: X = columns b 0..2, Y = columns b 3..5 (36 KB each, direct to
//             LDS); phase X runs the 9 positions (3 rows x 3 columns) of slot 0 while slot 1 receives Y, phase Y runs slot 1
//             while slot 0 receives X of the next chunk: one barrier per phase.  The RAW 10x10 patch of every quarter is
//             staged once per chunk (single buffer: all raw reads of a chunk happen in phase X, the next chunk's pieces are
//             written in phase Y) and transformed in registers at the start of phase X.
//   K trick = lane group kg = lane >> 4 reads channels 2kg, 2kg+1 with ONE ds_read_b64 per operand; MFMA s (0, 1) consumes channel
//             2kg + s of both operands (the same permutation of K on both sides).
//   epilogue= output transform of the wave's three rows (A^T is applied per row, then the three rows are combined into a partial
//             4x4 output), the two waves of a pair exchange half of their partial outputs through LDS (wave hh finishes output
//             rows 2hh, 2hh+1), then bias / ReLU / stores / 2x2 max-pool / statistics per lane.  Chunk split with the in-kernel
//             hand-off of g6d_common.h as in the other kernels.
#include "g6d_common.h"
#include <stdlib.h>
#include <stdio.h>
#include <algorithm>
#include <type_traits>

#ifndef W43_ABLATE
#define W43_ABLATE 0     // profiling builds only (results wrong by construction): 1 no barriers, 2 no next-chunk traffic, 3 no transform,
                         // 4 no fragment reads, 5 all of them, 7 no epilogue, 8 requests issued with EXEC = 0 (their scalar code stays), 10 filter pieces always from
                         // the first chunk (L1 / L2 hits)
#endif
#define W43ABL(n) (W43_ABLATE == (n) || W43_ABLATE == 5)
// -DW43_TIMING (profiling builds only, tools/w43_timing.sh): every wave stamps the shader clock at the phase boundaries of its chunk loop and
// adds the intervals to g_w43_timing[wave][slot] (slots: 0 prologue, 1 raw reads + input transform, 2 phase X, 3 barrier after X,
// 4 phase Y, 5 barrier after Y, 6 epilogue, 7 chunks counted); g6d_w43_timing_read copies and clears the table.
#ifdef W43_TIMING
__device__ unsigned long long g_w43_timing[4][8];
#define W43_T(slot) do { const long long t_ = clock64(); tacc[slot] += t_ - tlast; tlast = t_; } while (0)
#else
#define W43_T(slot) do { } while (0)
#endif

namespace {

#define W43_QPIX 101                         // position stride between quarters (10 x 10 used); odd: conflict-free b64 reads
#define W43_NQ 8                             // quarters per block
#define W43_RAWF (28 * 64 * 4)               // floats of the raw patch region: 8 quarters x 101 positions x 8 channels, rounded up to 28 KB
#define W43_MAX_SEG 4

constexpr float WA = 0.75f, WB = 1.5f;
constexpr float WA2 = WA * WA, WB2 = WB * WB, WA3 = WA * WA * WA, WB3 = WB * WB * WB;
constexpr float WC0 = WA2 * WB2, WC2 = WA2 + WB2;

struct W43Seg { int qstart, N, H, W, QH, QW, in_off, full_off, pool_off, ld_in, ld_full, ld_pool; };

struct W43Args {
  const float* in; const float* U; const float* bias; float* out_full; float* out_pool;
  int Cin, Cout, relu;
  int nseg, qtotal; W43Seg seg[W43_MAX_SEG];
  unsigned in_bytes;                             // extent of the input tensor(s) from `in` (< 2^31): bound of the buffer loads
  int splits, chunks_per_split; float* ws;
  int D;                                         // depth slices per image (1 for 2-D layers); KD = 3 pads in depth
  int H0, W0, ld0;                               // geometry of segment 0 (KD = 3 depth step)
  const float* in_scale; const float* in_shift; int in_relu;
  int aff_div;                                   // MODE 2: image i uses affine table i / aff_div
  double* stats; int stats_div;                  // [groups][Cout][2]; group of image i = i / stats_div (0: one group)
  G6dFin fin;
  int gx, gy, map_mode;                          // pixel-tile blocks, channel-slice blocks; block id -> (bx, by) mapping (w43_block_of)
};

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, float k, f2 c) { return __builtin_elementwise_fma(a, f2{k, k}, c); }

// ds_read_b64 that the compiler's load/store optimizer does not fuse into ds_read2(_st64)_b64: those run at half the LDS rate and with
// 16-lane service groups over 32 banks, for which the images below are 2- and 4-way conflicted (volatile keeps the counters tracked)
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ f2 lds_rd64(const lds_float* p) { return *reinterpret_cast<const volatile __attribute__((address_space(3))) f2*>(p); }
__device__ __forceinline__ f32x4 lds_rd128(const lds_float* p) { return *reinterpret_cast<const volatile __attribute__((address_space(3))) f32x4*>(p); }

// Accumulators: 18 x 4 tiles of 4 registers = 288 > the 256 AGPRs hipcc gives the builtin's accumulators; the 8 tiles that do not
// fit would be copied VGPR <-> AGPR around every MFMA (v_accvgpr_write / read: 128 vector-ALU instructions per chunk).  Those tiles
// (row 2, columns 4 and 5 at NT = 4) therefore use the VGPR form of the instruction, hand-issued: an accumulator is reused 4 MFMAs
// (128 cycles) later at the earliest, beyond every software wait state of the instruction; the epilogue waits 32 cycles first.
template <bool VG>
__device__ __forceinline__ void mfma16(float a, float b, f32x4& c) {
  if constexpr (VG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Block id -> (pixel-tile block bx, channel-slice block by).  map_mode 0: the launch grid itself (x = pixel tiles fastest, y = slices): the
// 256 resident blocks work on ONE channel slice, and every slice re-reads the layer's input from HBM (pyramid inputs are 0.6-1.2 GB at
// batches of 8-16: far beyond the 256 MB Infinity Cache).  map_mode 1 (1-D grid): hardware hands consecutive workgroup ids to the 8 XCDs
// in turn, so within a group of 8 pixel tiles the id runs (tile-in-group fastest, then slice): XCD k gets tile 8g + k with ALL its gy
// slices back to back — they run side by side on that XCD's CUs and share the tile's raw patches through its L2 (input read from HBM
// once instead of gy times; the filter slices of a layer are then streamed by every XCD from the Infinity Cache).  map_mode 2: slices
// fastest, plain (the slices of a tile land on neighbouring XCDs: shared through the Infinity Cache only).
__device__ __forceinline__ void w43_block_of(const W43Args& p, int& bx, int& by) {
  if (p.map_mode == 0) { bx = blockIdx.x; by = blockIdx.y; return; }
  const int L = blockIdx.x;
  if (p.map_mode == 2) { bx = L / p.gy; by = L - bx * p.gy; return; }
  const int per = 8 * p.gy, g = L / per, Lp = L - g * per;
  const int m = min(8, p.gx - 8 * g);
  by = Lp / m; bx = 8 * g + (Lp - by * m);
}

struct QGeo { int n, oy0, ox0; bool valid; int H, W, ld_in, ld_full, ld_pool, in_off, full_off, pool_off; };
__device__ __forceinline__ QGeo quarter_of(const W43Args& p, int bx, int q) {
  const int Q = bx * W43_NQ + q;
  int sidx = 0;
#pragma unroll
  for (int k = 1; k < W43_MAX_SEG; ++k) sidx = (k < p.nseg && Q >= p.seg[k].qstart) ? k : sidx;
  const W43Seg& sg = p.seg[sidx];
  QGeo g;
  g.valid = Q < p.qtotal;
  const int Ql = g.valid ? Q - sg.qstart : 0;
  const int per = sg.QH * sg.QW;
  g.n = Ql / per;
  const int r = Ql - g.n * per;
  const int qy = r / sg.QW, qx = r - qy * sg.QW;
  g.oy0 = 8 * qy; g.ox0 = 8 * qx;
  g.H = sg.H; g.W = sg.W; g.ld_in = sg.ld_in; g.ld_full = sg.ld_full; g.ld_pool = sg.ld_pool;
  g.in_off = sg.in_off; g.full_off = sg.full_off; g.pool_off = sg.pool_off;
  return g;
}

__device__ __forceinline__ void bt_full(const f2 (&d)[6], f2 (&r)[6]) {
  const f2 Ea = fma2(d[2], -WB2, d[4]), Oa = fma2(d[1], -WB2, d[3]);
  const f2 Eb = fma2(d[2], -WA2, d[4]), Ob = fma2(d[1], -WA2, d[3]);
  r[0] = fma2(d[0], WC0, fma2(d[2], -WC2, d[4]));
  r[1] = fma2(Oa, WA, Ea);
  r[2] = fma2(Oa, -WA, Ea);
  r[3] = fma2(Ob, WB, Eb);
  r[4] = fma2(Ob, -WB, Eb);
  r[5] = fma2(d[1], WC0, fma2(d[3], -WC2, d[5]));
}
__device__ __forceinline__ f2 mul2(f2 a, float k) { return a * f2{k, k}; }
// 1-D output transform of one row of six transform-domain values, two channel tiles at a time (packed)
__device__ __forceinline__ void at_row2(const f2 (&m)[6], f2 (&y)[4]) {
  const f2 s1 = fma2(m[2], 1.f, m[1]), d1 = fma2(m[2], -1.f, m[1]), s2 = fma2(m[4], 1.f, m[3]), d2 = fma2(m[4], -1.f, m[3]);
  y[0] = fma2(s2, 1.f, fma2(s1, 1.f, m[0]));
  y[1] = fma2(d2, WB, mul2(d1, WA));
  y[2] = fma2(s2, WB2, mul2(s1, WA2));
  y[3] = fma2(d2, WB3, fma2(d1, WA3, m[5]));
}
// 1-D output transform of one row of six transform-domain values
__device__ __forceinline__ void at_row(const float (&m)[6], float (&y)[4]) {
  const float s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
  y[0] = m[0] + s1 + s2;
  y[1] = fmaf(WB, d2, WA * d1);
  y[2] = fmaf(WB2, s2, WA2 * s1);
  y[3] = fmaf(WB3, d2, fmaf(WA3, d1, m[5]));
}

// MODE  operand prologue applied when the raw patch is written to LDS: 0 none, 1 InstanceNorm affine (+ReLU) with one table,
//       2 one table per image group (image / aff_div; LDS holds the tables of the block's eight quarters); zero padding stays zero
// KD    1: 2-D layer; 3: 3x3x3 layer, depth taps folded into the reduction (chunk = (kd, 8 channels) reads slice d + kd - 1);
//       25: 15x15 "same" correlation as 5x5 blocks of 3x3 (chunk = (8 channels, block), block shifts innermost); 9: 9x9 as 3x3 blocks
//       (the detector's 7x7 level, zero-extended by one tap on every side)
// NT    output channels of a block in 16s (4 or 2)
template <int MODE, int KD, int NT>
__global__ void __launch_bounds__(256, 1) wino43_kernel(const W43Args p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef W43_TIMING
  const long long t_start = clock64();
#endif
  constexpr int THREADS = 256, NQ = W43_NQ;
  constexpr int KB = KD == 25 ? 5 : (KD == 9 ? 3 : 0);     // correlation: the 3 KB x 3 KB "same" filter cut into KB x KB blocks of 3x3 (KD = KB^2)
  constexpr bool CORR = KB > 0;
  constexpr int RAWF = W43_RAWF;                            // 28 wave-instructions of 64 16-byte slots (the image itself ends at slot 1616)
  constexpr int HALF = 18 * 16 * NT * 8;                    // floats of one filter half-slot: [18 positions][16 NT channels][8]
  constexpr int FLT0 = 2 * RAWF;                            // two raw stages, then two filter slots, then the affine tables
  constexpr int AFF0 = FLT0 + 2 * HALF;
  constexpr int NPR = 7;                                    // 16-byte raw slots per thread and chunk
  constexpr int PCS = 9 * NT;                               // 1 KB direct-to-LDS pieces per half-slot (32 channels x 32 B each)
  constexpr int NPC = (PCS + 3) / 4;                        // ... per wave (NT = 2: 4.5 -> 5, the surplus pieces repeat piece idx % PCS)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx, by;
  w43_block_of(p, bx, by);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int pr = wave >> 1, hh = wave & 1;
  const int lt = lane & 15, kg = lane >> 4;
  const int n0 = by * (16 * NT);
  const int nc8 = p.Cin >> 3;
  const int c_first = blockIdx.z * p.chunks_per_split;
  const int c_last = min(KD * nc8, c_first + p.chunks_per_split) - 1;

  // ---- filter half-tiles: U43 is [chunk][half][co block][18 positions][NT/2][kg][lt][4] — a block's half is ONE contiguous run of
  // 18 NT KB in exactly the order of its LDS image.  A lane's 16 bytes at [position][np][kg][lt] are its B operands for the channel
  // tiles 2np and 2np+1 (two channels each): ONE ds_read_b128 per two tiles, conflict-free in all four 16-lane service groups, and at
  // the full LDS rate from one wave per SIMD (8-byte reads reach a fifth of it there: MI355X_MICROARCH.md, LDS).
  // Two slots: phase X computes from slot 0 while slot 1 receives Y of the chunk, phase Y from slot 1 while slot 0 receives X of the
  // next chunk.  The pieces (1 KB per wave instruction) travel global -> registers -> LDS: measured on this kernel
  // (profiles/r04_w43_ablate_*.md), a direct-to-LDS piece costs its wave ~60 cycles of issue beside the MFMAs — 25 pieces per wave and
  // chunk were 20 % of the kernel, plus 10 % for their scalar address / M0 code — against ~6 + 13 for a load and a ds_write_b128; the
  // registers are those of the input transform, idle between two transforms.  (Also measured and dropped: fragments straight from
  // L2 into registers without LDS, 17 % slower; a three-slot ring with two phases of lead; requests staggered over the waves.)
  const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U) + (size_t)by * HALF, 0, 0x7ffffff0, 0x00020000);
  const unsigned lane16 = lane * 16;                       // lane offset; the piece's offset (< 2^31: launch check) is the instruction's scalar offset
  const unsigned half_bytes = p.gy * (HALF * 4);
  f32x4 fp[NPC];
  auto fidx = [&](int k) { int idx = wave * NPC + k; if (PCS % 4 != 0) idx = idx % PCS; return idx; };
  auto load_f = [&](int chunk, int half, int k) {
    if (W43_ABLATE == 12) { asm volatile("" : "+v"(fp[k])); return; }
    if (W43ABL(2)) return;
    fp[k] = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, lane16, (2 * chunk + half) * half_bytes + fidx(k) * 1024, 0)));
  };
  auto store_f = [&](int slot, int k) {
    if (W43_ABLATE == 11) { asm volatile("" :: "v"(fp[k])); return; }
    if (W43ABL(2)) return;
    *reinterpret_cast<f32x4*>(__builtin_assume_aligned(lds + FLT0 + slot * HALF + fidx(k) * 256 + lane * 4, 16)) = fp[k];
  };

  // the first chunk's filter half is requested BEFORE the piece geometry (segment search, quarter table, ~60 LDS reads and the
  // affine tables): its ~2 us of latency under load run beside that work instead of behind it (profiles/r04_w43_phase_timing.md:
  // a block spent 12-15 k cycles in front of its first chunk)
#pragma unroll
  for (int k = 0; k < NPC; ++k) load_f(c_first, 0, k);
  __builtin_amdgcn_sched_barrier(0);

  // geometry of the block's eight quarters, computed ONCE (segment search + two runtime divisions each) by eight lanes and kept in LDS:
  // every thread needs it for its seven raw pieces and again in the epilogue — done per thread, the ~30 divisions were 3 us of a
  // block's 6 us prologue (profiles/r04_w43_phase_timing.md)
  __shared__ __attribute__((aligned(128))) int qtab[W43_NQ * 16];
  if (tid < W43_NQ) {
    const QGeo g = quarter_of(p, bx, tid);
    int* t = qtab + tid * 16;
    t[0] = g.valid; t[1] = g.n; t[2] = g.oy0; t[3] = g.ox0; t[4] = g.H; t[5] = g.W; t[6] = g.ld_in; t[7] = g.ld_full;
    t[8] = g.ld_pool; t[9] = g.in_off; t[10] = g.full_off; t[11] = g.pool_off;
  }
  __syncthreads();
  auto qgeo = [&](int q) {
    const int* t = qtab + q * 16;
    QGeo g;
    g.valid = t[0] != 0; g.n = t[1]; g.oy0 = t[2]; g.ox0 = t[3]; g.H = t[4]; g.W = t[5]; g.ld_in = t[6]; g.ld_full = t[7];
    g.ld_pool = t[8]; g.in_off = t[9]; g.full_off = t[10]; g.pool_off = t[11];
    return g;
  };

  // ---- raw patch loader.  LDS image: position (q, py, px) at ((q*101 + py*10 + px) * 8) floats, its two 4-channel halves swapped
  // when (py >> 2) is odd (every ds_read_b64 of the fragment loop is then bank-conflict free: exhaustive check over both 32-lane
  // service groups).  The image is filled in 16-byte slots s = (w*7 + j)*64 + lane — lane-linear per wave instruction, so that
  // MODE 0 can move the pieces global -> LDS directly (buffer_load ... lds: the LDS address is M0 + 16 lane, the GLOBAL address is
  // per lane: each lane fetches the piece that belongs into its slot; pieces outside the image, the padding position 100 of a
  // quarter and the slots behind the image ask for an offset beyond the tensor and the hardware writes zeros).
  int poff[NPR], aoff[MODE != 0 ? NPR : 1];
  bool pval[NPR];
  unsigned dbits = 0;                                          // KD = 3: bit 2j / 2j+1 = piece j has a slice below / above
  unsigned smask[CORR ? NPR : 1];                              // correlation: bits 0-4 / 8-12 = row / column of the piece inside the image under block shift b
  int rstep[CORR ? NPR : 1];
#pragma unroll
  for (int j = 0; j < NPR; ++j) {
    const int sl = (wave * NPR + j) * 64 + lane;
    const int pl = sl >> 1, q = pl / W43_QPIX, pp = pl - q * W43_QPIX;
    const int py = pp / 10, px = pp - py * 10;
    const int half = (sl & 1) ^ ((py >> 2) & 1);
    const bool live = (q < NQ) & (pp < 100);
    const QGeo g = qgeo(live ? q : 0);
    const int n = g.n;
    const int iy = g.oy0 + py - 1, ix = g.ox0 + px - 1;
    pval[j] = live & g.valid & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
    poff[j] = pval[j] ? g.in_off + ((n * g.H + iy) * g.W + ix) * g.ld_in + 4 * half : 0;
    if constexpr (MODE != 0) aoff[j] = (MODE >= 2 ? (live ? q : 0) * p.Cin : 0) + 4 * half;
    if constexpr (KD == 3) { const int dd = n % p.D; dbits |= (unsigned)(dd > 0) << (2 * j) | (unsigned)(dd < p.D - 1) << (2 * j + 1); }
    if constexpr (CORR) {
      unsigned m = 0;
#pragma unroll
      for (int b = 0; b < KB; ++b)
        m |= (unsigned)((unsigned)(iy + 3 * b - 3 * (KB / 2)) < (unsigned)g.H) << b | (unsigned)((unsigned)(ix + 3 * b - 3 * (KB / 2)) < (unsigned)g.W) << (8 + b);
      smask[j] = (live & g.valid) ? m : 0u;
      rstep[j] = 3 * g.W * g.ld_in;
      poff[j] = g.in_off + ((n * g.H + iy) * g.W + ix) * g.ld_in + 4 * half;      // the UNSHIFTED position (used under smask only)
    }
  }
  unsigned pboff[KD == 1 ? NPR : 1];                          // byte offsets of the pieces for the buffer loads (beyond the tensor: zero)
  if constexpr (KD == 1) {
#pragma unroll
    for (int j = 0; j < NPR; ++j) pboff[j] = pval[j] ? (unsigned)poff[j] << 2 : 0x80000000u;
  }
  const int slice = p.H0 * p.W0 * p.ld0;                      // KD = 3: one depth step
  const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  f32x4 rp[NPR];
  bool rv[MODE != 0 ? NPR : 1];
  // piece j of chunk `chunk` -> rp[j] (the LDS image is written by store_piece, where the MODE != 0 prologue runs)
  auto load_piece = [&](int j, int chunk) {
    if (W43_ABLATE == 12) { asm volatile("" : "+v"(rp[j])); return; }      // (an opaque definition: the stores stay)
    if (W43ABL(2)) return;
    const int kd = CORR ? chunk % KD : (KD != 1 ? chunk / nc8 : 0), cc = CORR ? chunk / KD : (KD != 1 ? chunk - kd * nc8 : chunk);
    unsigned voff; int soff = 0;
    if constexpr (KD == 1) { voff = pboff[j]; soff = cc * 32; if constexpr (MODE != 0) rv[j] = pval[j]; }
    else {
      bool v = pval[j];
      int off = poff[j] + cc * 8;
      if constexpr (KD == 3) { v &= kd == 1 || ((dbits >> (2 * j + (kd >> 1))) & 1u) != 0; off += (kd - 1) * slice; }
      if constexpr (CORR) {
        const int bi = kd / KB, bj = kd - KB * bi;
        v = ((smask[j] >> bi) & (smask[j] >> (8 + bj)) & 1u) != 0;
        off += (bi - KB / 2) * rstep[j] + (bj - KB / 2) * 3 * p.ld0;
      }
      if constexpr (MODE != 0) rv[j] = v;
      voff = v ? (unsigned)off << 2 : 0x80000000u;
    }
    // bounds-checked buffer load: pieces outside the image ask for an offset beyond the tensor and get zeros from the hardware
    rp[j] = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, voff, soff, 0)));
  };
  auto store_piece = [&](int j, int chunk, int stage) {        // affine (+ReLU) with exact zeros outside the image for MODE != 0, -> LDS
    if (W43_ABLATE == 11) { asm volatile("" :: "v"(rp[j])); return; }         // (an opaque use: the loads stay)
    if (W43ABL(2)) return;
    f32x4 v = rp[j];
    if constexpr (MODE != 0) {
      const int cc = KD != 1 ? chunk % nc8 : chunk;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(lds + AFF0 + aoff[j] + cc * 8);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(lds + AFF0 + (MODE >= 2 ? NQ : 1) * p.Cin + aoff[j] + cc * 8);
      v = v * sc + sh;
      if (p.in_relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      if (!rv[j]) v = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    *reinterpret_cast<f32x4*>(__builtin_assume_aligned(lds + stage * RAWF + ((wave * NPR + j) * 64 + lane) * 4, 16)) = v;
  };
  if constexpr (MODE != 0) {                  // InstanceNorm affine tables -> LDS: [G][Cin] scales, then [G][Cin] shifts
    constexpr int G = MODE >= 2 ? NQ : 1;
    for (int i = tid; i < G * p.Cin; i += THREADS) {
      int g = 0;
      if constexpr (MODE >= 2) g = p.aff_div > 0 ? (qgeo(i / p.Cin).n / p.D) / p.aff_div : 0;
      const int c = MODE >= 2 ? i % p.Cin : i;
      lds[AFF0 + i] = p.in_scale[g * p.Cin + c];
      lds[AFF0 + G * p.Cin + i] = p.in_shift[g * p.Cin + c];
    }
    __syncthreads();
  }
  // ---- fragment bases.  A: tile lt of the pair = quarter 4 pr + (lt >> 2), tile (ty, tx) of its 2x2; raw rows 4 ty + i
  const int ty = (lt >> 1) & 1, tx = lt & 1;
  const int apos = ((4 * pr + (lt >> 2)) * W43_QPIX + (4 * ty) * 10 + 4 * tx) * 8;
  const int abase_lo = apos + 2 * (kg ^ (2 * ty));            // rows i = 0..3: ((4 ty + i) >> 2) & 1 == ty
  const int abase_hi = apos + 2 * (kg ^ (2 * (ty ^ 1)));      // rows i = 4, 5
  int arow[5];                                                // float offsets of raw rows hh .. hh+4 of the lane's tile
#pragma unroll
  for (int k = 0; k < 5; ++k) arow[k] = (hh + k < 4 ? abase_lo : abase_hi) + (hh + k) * 80;
  // B: 16 bytes at [position 9 hh + pp][np][kg][lt] of the slot = lane-linear within the 1 KB row of (position, np)
  const int bbase = FLT0 + (9 * hh) * (NT / 2) * 256 + lane * 4;

  f32x4 acc[3][6][NT];

  const lds_float* L = (const lds_float*)lds;
#ifdef W43_TIMING
  const long long t_geo = clock64() - t_start;      // launch to first request: piece geometry (quarter_of: runtime divisions), tables
#endif
#pragma unroll
  for (int j = 0; j < NPR; ++j) load_piece(j, c_first);
  __builtin_amdgcn_sched_barrier(0);
  // accumulator initialisation (288 vector-ALU instructions) in the shadow of the first chunk's requests
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[a][b][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < NPC; ++k) store_f(0, k);
#pragma unroll
  for (int j = 0; j < NPR; ++j) store_piece(j, c_first, 0);
  __syncthreads();

  f2 V[3][6];
  // One phase: the 9 positions (3 rows x columns B0..B0+2) of filter slot `slot`, 2 NT MFMAs each (s = 0: channels 2kg, s = 1: 2kg + 1;
  // consecutive MFMAs go to different accumulators).  One wave per SIMD: only the wave's own instruction order hides latency, so
  // everything else is pinned into the MFMA gaps — the NT/2 fragment reads of position pp + 1 behind the first MFMAs of position pp,
  // memory request j of the phase (`req(j)`: direct-to-LDS pieces) behind MFMA 2j + (w & 1) of the phase's second halves.
  f32x4 bq[2][NT / 2];
  auto phase_begin = [&](int slot) {              // fragments of the phase's first position
    const lds_float* S = L + bbase + slot * HALF;
#pragma unroll
    for (int n = 0; n < NT / 2; ++n) bq[0][n] = W43ABL(4) ? f32x4{1.f, 1.f, 1.f, 1.f} : lds_rd128(S + n * 256);
    if (W43ABL(4)) {
#pragma unroll
      for (int n = 0; n < NT / 2; ++n) bq[1][n] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
  };
  auto phase = [&](auto B0c, int slot, auto&& req) {
    constexpr int B0 = decltype(B0c)::value;
    const lds_float* S = L + bbase + slot * HALF;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pp = 0; pp < 9; ++pp) {
      const int ai = pp / 3, b = B0 + pp % 3;
#pragma unroll
      for (int m = 0; m < 2 * NT; ++m) {
        const int sidx = m / NT, n = m % NT;
        const float bv = bq[pp & 1][n >> 1][2 * (n & 1) + sidx];
        if (NT == 4 && ai == 2 && b >= 4) mfma16<true>(V[ai][b][sidx], bv, acc[ai][b][n]);
        else mfma16<false>(V[ai][b][sidx], bv, acc[ai][b][n]);
        if (m < NT / 2) { if (pp + 1 < 9 && !W43ABL(4)) bq[(pp + 1) & 1][m] = lds_rd128(S + ((pp + 1) * (NT / 2) + m) * 256); }
        else if (m >= NT) {
          const int k = pp * NT + (m - NT);          // request slot of the phase (9 NT of them)
          req(k);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I3 = std::integral_constant<int, 3>;

#ifdef W43_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = t_start;
#endif
  W43_T(0);
#ifdef W43_TIMING_GEO
  tacc[0] = t_geo;                                  // (-DW43_TIMING_GEO: slot 0 = the part of the prologue in front of the first request)
#endif
  for (int cc = c_first; cc <= c_last; ++cc) {
    const int cn = min(cc + 1, c_last);           // the last chunk re-requests itself: no branches
    const int st = (cc - c_first) & 1;            // raw stage of this chunk
    // ---- phase X: raw tile -> registers, input transform, 9 positions of slot 0 while slot 1 receives Y of this chunk.
    // The wave's rows a = 3hh..3hh+2 of B^T d need raw rows hh..hh+4 only (r0..r2: d0..d4, r3..r5: d1..d5)
    {
      f2 e[5][6];
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int k = 0; k < 5; ++k) e[k][j] = W43ABL(4) ? f2{1.f, 1.f} : lds_rd64(L + st * RAWF + arow[k] + j * 8);
      phase_begin(0);                              // ... and the first filter fragments: they arrive behind the transform
      if (W43ABL(3)) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 6; ++b) V[a][b] = e[a][b];
      } else {
        f2 T[3][6];
        if (hh == 0) {
#pragma unroll
          for (int j = 0; j < 6; ++j) {            // columns: rows 0..2 of B^T d
            const f2 E = fma2(e[2][j], -WB2, e[4][j]), O = fma2(e[1][j], -WB2, e[3][j]);
            T[0][j] = fma2(e[0][j], WC0, fma2(e[2][j], -WC2, e[4][j]));
            T[1][j] = fma2(O, WA, E);
            T[2][j] = fma2(O, -WA, E);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 6; ++j) {            // rows 3..5 (e[k] = d[k + 1])
            const f2 E = fma2(e[1][j], -WA2, e[3][j]), O = fma2(e[0][j], -WA2, e[2][j]);
            T[0][j] = fma2(O, WB, E);
            T[1][j] = fma2(O, -WB, E);
            T[2][j] = fma2(e[0][j], WC0, fma2(e[2][j], -WC2, e[4][j]));
          }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) bt_full(T[a], V[a]);
      }
    }
    W43_T(1);
    __builtin_amdgcn_sched_barrier(0);
    // requests of phase X: the NPC pieces of Y of this chunk and the raw pieces of chunk c+1 are loaded behind the first positions;
    // the filter pieces go to slot 1 behind the last positions (the raw pieces wait for phase Y: the other raw stage is free, but
    // the LDS stores are spread over both phases)
    constexpr int NS = 9 * NT;                       // request slots of a phase
    phase(I0{}, 0, [&](int k) {
      if (k < NPC) load_f(cc, 1, k);
      else if (k - NPC < NPR) load_piece(k - NPC, cn);
      else if (k >= NS - NPC) store_f(1, k - (NS - NPC));
    });
    W43_T(2);
    if (!W43ABL(1)) __syncthreads();
    W43_T(3);
    // ---- phase Y: 9 positions of slot 1 while slot 0 receives X of chunk c+1 and the other raw stage its image
    phase_begin(1);
    __builtin_amdgcn_sched_barrier(0);
    phase(I3{}, 1, [&](int k) {
      if (k < NPC) load_f(cn, 0, k);
      else if (k - NPC < NPR) store_piece(k - NPC, cn, st ^ 1);
      else if (k >= NS - NPC) store_f(0, k - (NS - NPC));
    });
    W43_T(4);
    if (!W43ABL(1)) __syncthreads();
    W43_T(5);
#ifdef W43_TIMING
    tacc[7] += 1;
#endif
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the hand-issued MFMAs of the last positions have left the pipe

  if (W43_ABLATE == 7) {
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b)
#pragma unroll
        for (int n = 0; n < NT; ++n) t += acc[a][b][n][0] + acc[a][b][n][1] + acc[a][b][n][2] + acc[a][b][n][3];
    if (t == 12345.f && p.out_full) p.out_full[tid] = t;
    return;
  }
  // ---------------------------------------------------------------- epilogue
  // element e = (nt, r): accumulator register r of channel tile nt = Winograd tile r of quarter 4 pr + kg, channel n0 + 16 nt + lt.
  // Partial outputs of the wave's three transform rows, P[x][y] = sum_a A^T[x][3hh+a] (row a transformed along b).
  // Wave hh keeps output rows 2hh, 2hh+1 and hands rows 2(1-hh), 2(1-hh)+1 to its partner, NT/2 channel tiles per round (64 floats per
  // lane and round, lane-linear 16-byte rows: 16 KB per wave; the K loop's LDS is free: every wave has passed its last barrier).
  // All of it on PAIRS of channel tiles (n = 2 rd, 2 rd + 1 in the two halves of a 64-bit register pair): v_pk_fma_f32 does two lanes'
  // worth of the transform per issue slot, and vector-ALU time is what the epilogue is made of (~2100 scalar instructions before).
  f2 Y[NT / 2][4][2][4];                             // [rd][r][output row 2hh + xl][output column], .x = tile 2 rd, .y = tile 2 rd + 1
  float* xbuf = lds;
  constexpr float ONE = 1.f;
#pragma unroll
  for (int rd = 0; rd < NT / 2; ++rd) {
    f2 keep[4][2][4], send[4][2][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f2 R[3][4];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        f2 m[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) m[b] = f2{acc[a][b][2 * rd][r], acc[a][b][2 * rd + 1][r]};
        at_row2(m, R[a]);
      }
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        if (hh == 0) {                               // points 0, +a, -a
          const f2 s = fma2(R[2][y], ONE, R[1][y]), dd = fma2(R[2][y], -ONE, R[1][y]);
          keep[r][0][y] = fma2(s, ONE, R[0][y]); keep[r][1][y] = mul2(dd, WA);
          send[r][0][y] = mul2(s, WA2); send[r][1][y] = mul2(dd, WA3);
        } else {                                     // points +b, -b, inf
          const f2 s = fma2(R[1][y], ONE, R[0][y]), dd = fma2(R[1][y], -ONE, R[0][y]);
          send[r][0][y] = s; send[r][1][y] = mul2(dd, WB);
          keep[r][0][y] = mul2(s, WB2); keep[r][1][y] = fma2(dd, WB3, R[2][y]);
        }
      }
    }
    if (rd > 0) __syncthreads();                     // the previous round's rows have been read
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int xl = 0; xl < 2; ++xl)
#pragma unroll
        for (int yp = 0; yp < 2; ++yp)
          *reinterpret_cast<f32x4*>(xbuf + wave * 4096 + (((r * 2 + xl) * 2 + yp) * 64 + lane) * 4) =
              f32x4{send[r][xl][2 * yp].x, send[r][xl][2 * yp].y, send[r][xl][2 * yp + 1].x, send[r][xl][2 * yp + 1].y};
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int xl = 0; xl < 2; ++xl)
#pragma unroll
        for (int yp = 0; yp < 2; ++yp) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(xbuf + (wave ^ 1) * 4096 + (((r * 2 + xl) * 2 + yp) * 64 + lane) * 4);
          Y[rd][r][xl][2 * yp] = fma2(f2{o[0], o[1]}, ONE, keep[r][xl][2 * yp]);
          Y[rd][r][xl][2 * yp + 1] = fma2(f2{o[2], o[3]}, ONE, keep[r][xl][2 * yp + 1]);
        }
  }
  if (p.splits > 1) {
    // partial OUTPUT tiles (the output transform is linear) -> workspace as [split][tile][k][thread] 16-byte pieces; the block of a tile
    // that arrives last adds them in split order and carries on with bias / ReLU / pool / statistics
    constexpr int TILE = THREADS * NT * 32;
    const int ntiles = p.gx * p.gy, tile = by * p.gx + bx;
    float* part = p.ws + G6D_WS_COUNTERS + (size_t)tile * TILE + tid * 4;
    const size_t zstride = (size_t)ntiles * TILE;
#pragma unroll
    for (int rd = 0; rd < NT / 2; ++rd)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int xl = 0; xl < 2; ++xl)
#pragma unroll
          for (int yp = 0; yp < 2; ++yp)
            g6d_store_wt(part + blockIdx.z * zstride + (((rd * 4 + r) * 2 + xl) * 2 + yp) * (THREADS * 4),
                         f32x4{Y[rd][r][xl][2 * yp].x, Y[rd][r][xl][2 * yp].y, Y[rd][r][xl][2 * yp + 1].x, Y[rd][r][xl][2 * yp + 1].y});
    if (!g6d_split_arrive(reinterpret_cast<int*>(p.ws) + tile, p.splits, reinterpret_cast<int*>(lds))) return;
#pragma unroll
    for (int rd = 0; rd < NT / 2; ++rd)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int xl = 0; xl < 2; ++xl)
#pragma unroll
          for (int y = 0; y < 4; ++y) Y[rd][r][xl][y] = f2{0.f, 0.f};
    for (int z = 0; z < p.splits; ++z) {
#pragma unroll
      for (int rd = 0; rd < NT / 2; ++rd)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int xl = 0; xl < 2; ++xl)
#pragma unroll
            for (int yp = 0; yp < 2; ++yp) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(part + (size_t)z * zstride + (((rd * 4 + r) * 2 + xl) * 2 + yp) * (THREADS * 4));
              Y[rd][r][xl][2 * yp] = fma2(f2{v[0], v[1]}, ONE, Y[rd][r][xl][2 * yp]);
              Y[rd][r][xl][2 * yp + 1] = fma2(f2{v[2], v[3]}, ONE, Y[rd][r][xl][2 * yp + 1]);
            }
    }
  }
  const QGeo g = qgeo(4 * pr + kg);                   // the lane's quarter: accumulator register r = tile r of it
  const bool do_relu = p.relu != 0, do_stats = p.stats != nullptr;
  const int Hp = g.H >> 1, Wp = g.W >> 1;
  // Stores.  The address of an output element is hoisted out of the element loops: one 64-bit base per tile row of the lane's quarter
  // (r, output row), then 32-bit column / channel-tile offsets — computed per element (the compiler does not hoist it through the
  // unrolled loops) the address arithmetic was ~1300 of the epilogue's instructions, 370 of them quarter-rate integer multiplies.
  float st1[NT], st2[NT];
  f2 bv[NT / 2];
#pragma unroll
  for (int n = 0; n < NT; ++n) { st1[n] = 0.f; st2[n] = 0.f; }
#pragma unroll
  for (int rd = 0; rd < NT / 2; ++rd)
    bv[rd] = p.bias ? f2{p.bias[n0 + 32 * rd + lt], p.bias[n0 + 32 * rd + 16 + lt]} : f2{0.f, 0.f};
  const int ldf = g.ld_full, ldp = g.ld_pool;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int oy = g.oy0 + 4 * (r >> 1) + 2 * hh, ox = g.ox0 + 4 * (r & 1);
    const int ncol = g.W - ox;                                          // columns c < ncol of the tile row are inside the map
    const bool row0 = g.valid && oy < g.H, row1 = g.valid && oy + 1 < g.H;
    float* f0 = p.out_full + (size_t)g.full_off + ((size_t)(g.n * g.H + oy) * g.W + ox) * (size_t)ldf + (n0 + lt);
    float* f1 = f0 + (size_t)g.W * (size_t)ldf;
    const int py = oy >> 1, px0 = ox >> 1;
    float* q0 = p.out_pool + (size_t)g.pool_off + ((size_t)(g.n * Hp + py) * Wp + px0) * (size_t)ldp + (n0 + lt);
    const bool prow = g.valid && py < Hp;
#pragma unroll
    for (int rd = 0; rd < NT / 2; ++rd) {
      f2 yv[2][4];
#pragma unroll
      for (int xl = 0; xl < 2; ++xl)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          yv[xl][c] = fma2(Y[rd][r][xl][c], ONE, bv[rd]);
          if (do_relu) yv[xl][c] = f2{fmaxf(yv[xl][c].x, 0.f), fmaxf(yv[xl][c].y, 0.f)};
        }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int n = 2 * rd + u;
        float y[2][4];
#pragma unroll
        for (int xl = 0; xl < 2; ++xl)
#pragma unroll
          for (int c = 0; c < 4; ++c) y[xl][c] = u ? yv[xl][c].y : yv[xl][c].x;
        if (p.out_full) {
#pragma unroll
          for (int xl = 0; xl < 2; ++xl)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if ((xl ? row1 : row0) && c < ncol) {
                (xl ? f1 : f0)[c * ldf + 16 * n] = y[xl][c];
                if (do_stats) { st1[n] += y[xl][c]; st2[n] += y[xl][c] * y[xl][c]; }
              }
        }
        if (p.out_pool && prow) {
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
            if (px0 + c2 < Wp)
              q0[c2 * ldp + 16 * n] = fmaxf(fmaxf(y[0][2 * c2], y[0][2 * c2 + 1]), fmaxf(y[1][2 * c2], y[1][2 * c2 + 1]));
        }
      }
    }
  }
  if (do_stats) {
    // [quarter][half][channel][2] float partials in LDS, then one fp64 atomic per channel and run of quarters with the same group
    float* sred = lds;
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      sred[(((4 * pr + kg) * 2 + hh) * 16 * NT + 16 * n + lt) * 2] = st1[n];
      sred[(((4 * pr + kg) * 2 + hh) * 16 * NT + 16 * n + lt) * 2 + 1] = st2[n];
    }
    __syncthreads();
    if (tid < 16 * NT) {
      double a1 = 0.0, a2 = 0.0;
      int cur = -1;
      for (int q = 0; q < NQ; ++q) {
        const QGeo qg = qgeo(q);
        if (!qg.valid) break;
        const int gi = p.stats_div > 0 ? (qg.n / p.D) / p.stats_div : 0;
        if (gi != cur && cur >= 0) {
          double* st = p.stats + ((size_t)cur * p.Cout + n0 + tid) * 2;
          atomicAdd(st, a1); atomicAdd(st + 1, a2); a1 = a2 = 0.0;
        }
        cur = gi;
        a1 += (double)sred[((q * 2 + 0) * 16 * NT + tid) * 2] + (double)sred[((q * 2 + 1) * 16 * NT + tid) * 2];
        a2 += (double)sred[((q * 2 + 0) * 16 * NT + tid) * 2 + 1] + (double)sred[((q * 2 + 1) * 16 * NT + tid) * 2 + 1];
      }
      if (cur >= 0) {
        double* st = p.stats + ((size_t)cur * p.Cout + n0 + tid) * 2;
        atomicAdd(st, a1); atomicAdd(st + 1, a2);
      }
    }
    if (p.fin.scale) {
      __syncthreads();
      g6d_finalize_stats(p.fin, p.gx * p.gy, reinterpret_cast<int*>(lds));
    }
  }
#ifdef W43_TIMING
  W43_T(6);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&g_w43_timing[wave][i], (unsigned long long)tacc[i]);
  }
#endif
}

// ---- launch: split over the chunks, instantiation
template <int MODE, int KD, int NT>
int w43_launch_t(W43Args& a, long long blocks, hipStream_t stream) {
  const size_t lds_bytes = ((size_t)2 * W43_RAWF + 2 * (18 * 16 * NT * 8) + (MODE == 0 ? 0 : (MODE >= 2 ? 2 * W43_NQ : 2) * a.Cin)) * sizeof(float);
  const size_t need = std::max(lds_bytes, (size_t)4 * 4096 * sizeof(float));      // the epilogue exchange: 16 KB per wave
  g6d_allow_lds(reinterpret_cast<const void*>(&wino43_kernel<MODE, KD, NT>), 160 * 1024 - 512);      // (512 B of static LDS: the quarter table)
  a.gx = (int)blocks; a.gy = a.Cout / (16 * NT);
  a.map_mode = (int)g6d_knob(G6D_KNOB_W43_MAP);
  if (a.gy == 1 || (long long)a.gx * a.gy >= (1ll << 31)) a.map_mode = 0;
  const dim3 grid = a.map_mode == 0 ? dim3((unsigned)blocks, a.gy, a.splits) : dim3((unsigned)(blocks * a.gy), 1, a.splits);
  hipLaunchKernelGGL((wino43_kernel<MODE, KD, NT>), grid, dim3(256), need, stream, a);
  return g6d_check_launch("wino43_conv3x3");
}

// The hard size limits of a launch, shared by w43_run (error) and g6d_wino43_eligible (fall back to another kernel): 32-bit grid,
// the 2^31-byte reach of the buffer loads on input and filter bank, the affine tables beside the stages in LDS.  nullptr = fits.
const char* w43_limits(long long quarters, long long in_extent_floats, int kd, int Cin, int Cout, int mode) {
  if ((quarters + W43_NQ - 1) / W43_NQ > 0x3fffffffll) return "wino43: grid too large";
  if (in_extent_floats * 4 >= (1ll << 31)) return "wino43: input tensor exceeds 2^31 bytes";
  if ((long long)kd * (Cin / 8) * 36 * Cout * 32 >= (1ll << 31)) return "wino43: filter bank exceeds 2^31 bytes";
  const int nt = (Cout & 63) ? 2 : 4;
  if (mode != 0 && (size_t)(mode >= 2 ? 2 * W43_NQ : 2) * Cin * 4 + ((size_t)2 * W43_RAWF + 2 * 18 * 16 * nt * 8) * 4 > 160 * 1024 - 512)
    return "wino43: affine tables do not fit LDS";
  return nullptr;
}

int w43_run(W43Args& a, int mode, int kd, float* workspace, size_t workspace_bytes, hipStream_t stream) {
  long long quarters = 0, in_extent = 0;
  double out_elems = 0.0;
  for (int k = 0; k < a.nseg; ++k) {
    W43Seg& g = a.seg[k];
    in_extent = std::max(in_extent, (long long)g.in_off + (long long)g.N * g.H * g.W * g.ld_in);
    g.QH = (g.H + 7) / 8; g.QW = (g.W + 7) / 8;
    g.qstart = (int)quarters;
    quarters += (long long)g.N * g.QH * g.QW;
    out_elems += (double)g.N * g.H * g.W;
  }
  a.H0 = a.seg[0].H; a.W0 = a.seg[0].W; a.ld0 = a.seg[0].ld_in;
  const long long blocks = (quarters + W43_NQ - 1) / W43_NQ;
  const int nt = (a.Cout & 63) ? 2 : 4;
  if (const char* why = w43_limits(quarters, in_extent, kd, a.Cin, a.Cout, mode)) { g6d_set_error(why); return G6D_EINVAL; }
  a.qtotal = (int)quarters;
  a.in_bytes = (unsigned)(in_extent * 4);
  // Split of the (kd, chunk) list over gridDim.z: one block per CU is resident and runs a serial loop of ~2.6 us per chunk (NT = 4;
  // ~1.5 at NT = 2); pick the split count with the smallest modelled time, as wino_conv.hip does (constants overridable)
  const int nchunks = kd * (a.Cin / 8);
  int splits = 1;
  const long long grid2 = blocks * (a.Cout / (16 * nt));
  const int split_max = (int)g6d_knob(G6D_KNOB_W43_SPLIT_MAX);
  const double m_gain = g6d_knob(G6D_KNOB_W43_SPLIT_GAIN);
  const double m_chunk4 = g6d_knob(G6D_KNOB_W43_CHUNK_US);
  const size_t room = workspace && workspace_bytes > G6D_WS_COUNTER_BYTES ? workspace_bytes - G6D_WS_COUNTER_BYTES : 0;
  const double tile_bytes = (double)grid2 * 256 * nt * 32 * sizeof(float);
  if (room > 0 && grid2 <= G6D_WS_COUNTERS && split_max > 1 && nchunks >= 4) {
    const double out_bytes = out_elems * a.Cout * sizeof(float);
    double best = 1e30;
    for (int sp = 1; sp <= split_max && sp <= nchunks / 2; ++sp) {
      if ((double)sp * tile_bytes > (double)room) break;
      const int cps_ = (nchunks + sp - 1) / sp, real = (nchunks + cps_ - 1) / cps_;
      if (real != sp) continue;
      const double rounds = (double)((grid2 * sp + 255) / 256);
      double t = rounds * (cps_ * (nt == 4 ? m_chunk4 : 0.55 * m_chunk4) + 6.0);
      if (sp > 1) t += 2.0 + 0.6 * sp + (2 * sp + 1) * out_bytes / 2.5e6;
      if (t < best * m_gain) { best = t; splits = sp; }
    }
  }
  const int cps = (nchunks + splits - 1) / splits;
  splits = (nchunks + cps - 1) / cps;
  a.splits = splits; a.chunks_per_split = cps; a.ws = workspace;
  const bool debug = g6d_knob(G6D_KNOB_WINO_DEBUG) == 1;
  if (debug) fprintf(stderr, "wino43 %d seg, %dx%dx%d->%d kd=%d mode=%d: grid %lld x %d splits of %d chunks\n", a.nseg, a.seg[0].H, a.seg[0].W, a.Cin,
                     a.Cout, kd, mode, grid2, splits, cps);
  if (kd == 25) return nt == 4 ? w43_launch_t<0, 25, 4>(a, blocks, stream) : w43_launch_t<0, 25, 2>(a, blocks, stream);
  if (kd == 9) return nt == 4 ? w43_launch_t<0, 9, 4>(a, blocks, stream) : w43_launch_t<0, 9, 2>(a, blocks, stream);
  if (nt != 4) { g6d_set_error("wino43: Cout % 64 == 0 expected"); return G6D_EINVAL; }
  if (kd == 3) return mode == 2 ? w43_launch_t<2, 3, 4>(a, blocks, stream) : mode == 1 ? w43_launch_t<1, 3, 4>(a, blocks, stream) : w43_launch_t<0, 3, 4>(a, blocks, stream);
  if (mode == 2) return w43_launch_t<2, 1, 4>(a, blocks, stream);
  if (mode == 1) return w43_launch_t<1, 1, 4>(a, blocks, stream);
  return w43_launch_t<0, 1, 4>(a, blocks, stream);
}

}  // namespace

#ifdef W43_TIMING
extern "C" int g6d_w43_timing_read(unsigned long long* out) {      // [4 waves][8 slots]; clears the table
  unsigned long long zero[32] = {};
  if (hipDeviceSynchronize() != hipSuccess) return G6D_ELAUNCH;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w43_timing), sizeof(zero)) != hipSuccess) return G6D_ELAUNCH;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_w43_timing), zero, sizeof(zero)) == hipSuccess ? G6D_OK : G6D_ELAUNCH;
}
#endif

// One trunk layer over up to 4 map sizes in ONE launch, as g6d_wino_conv3x3_multi, on the F(4x4,3x3) kernel.  U43 = the filters
// transformed on the host (backbone.winograd43_filters): [Cin/8][2][Cout/64][18][2][4][16][4] (include/gen6d_hip.h).
// Replaces features[4..27] of vgg11_bn on the detector's image pyramid (network/pretrain_models.py:17-25, network/detector.py:236-241).
extern "C" int g6d_wino43_conv3x3_multi(const G6dWinoSeg* segs, int nseg, int Cin, const float* U43, const float* bias, int Cout, int relu,
                                        float* workspace, size_t workspace_bytes, g6d_stream_t stream) {
  if (!segs || nseg < 1 || nseg > W43_MAX_SEG || !U43 || Cin <= 0 || (Cin & 7) || Cout <= 0 || (Cout & 63) || !g6d_aligned16(U43)) {
    g6d_set_error("wino43_conv3x3_multi: bad args (1..4 segments, Cin % 8 == 0, Cout % 64 == 0)"); return G6D_EINVAL;
  }
  W43Args a = {};
  const bool want_full = segs[0].out_full != nullptr, want_pool = segs[0].out_pool != nullptr;
  if (!want_full && !want_pool) { g6d_set_error("wino43_conv3x3_multi: no output"); return G6D_EINVAL; }
  const float* in0 = segs[0].in; float* f0 = segs[0].out_full; float* p0 = segs[0].out_pool;
  for (int k = 0; k < nseg; ++k) {
    const G6dWinoSeg& g = segs[k];
    if (!g.in || (g.out_full != nullptr) != want_full || (g.out_pool != nullptr) != want_pool || g.N <= 0 || g.H <= 0 || g.W <= 0 ||
        (g.ld_in & 3) || g.ld_in < Cin || (want_full && g.ld_full < Cout) || (want_pool && (g.ld_pool < Cout || g.H < 2 || g.W < 2)) ||
        !g6d_aligned16(g.in)) {
      g6d_set_error("wino43_conv3x3_multi: bad segment (all segments give the same kinds of output)"); return G6D_EINVAL;
    }
    if (g.in < in0) in0 = g.in;
    if (want_full && g.out_full < f0) f0 = g.out_full;
    if (want_pool && g.out_pool < p0) p0 = g.out_pool;
  }
  a.in = in0; a.U = U43; a.bias = bias; a.out_full = f0; a.out_pool = p0;
  a.Cin = Cin; a.Cout = Cout; a.relu = relu; a.D = 1; a.nseg = nseg;
  for (int k = 0; k < nseg; ++k) {
    const G6dWinoSeg& g = segs[k];
    const long long io = g.in - in0, fo = want_full ? g.out_full - f0 : 0, po = want_pool ? g.out_pool - p0 : 0;
    if (io + (long long)g.N * g.H * g.W * g.ld_in >= (1ll << 29) || fo + (long long)g.N * g.H * g.W * g.ld_full >= (1ll << 31) ||
        po + (long long)g.N * g.H * g.W * g.ld_pool >= (1ll << 31)) {
      g6d_set_error("wino43_conv3x3_multi: segments must lie within 2^29 floats of each other (allocate them from one buffer)"); return G6D_EINVAL;
    }
    a.seg[k] = W43Seg{0, g.N, g.H, g.W, 0, 0, (int)io, (int)fo, (int)po, g.ld_in, g.ld_full, g.ld_pool};
  }
  return w43_run(a, 0, 1, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

// The detector's 15x15 reference-as-filter correlation (network/detector.py:222-224) as 5x5 blocks of 3x3 sub-filters accumulated in
// the F(4x4,3x3) transform domain: 225 taps cost 25 * 36 / 16 = 56.25 multiplications per output.  kblocks = 3: a 9x9 "same" correlation
// as 3x3 blocks — the 7x7 level with its filters zero-extended by one tap on every side: 9 * 36 / 16 = 20.25 multiplications instead of 49.  Maps as g6d_corr2d_wino_multi;
// U43 = the 25 sub-filter banks transformed like g6d_wino43_conv3x3_multi's, CHUNK-major: [Cin/8 * 25][2][Cout/CB][18][CB/32][4][16][4], row
// c * 25 + b = 8-channel chunk c of block b = 5 bi + bj holding w[:, 3bi..3bi+2, 3bj..3bj+2, :]; Cout % 32 == 0.
extern "C" int g6d_corr2d_wino43_multi(const G6dCorrSeg* segs, int nseg, int Cin, const float* U43, int Cout, int kblocks, float* workspace,
                                       size_t workspace_bytes, g6d_stream_t stream) {
  if (!segs || nseg < 1 || nseg > W43_MAX_SEG || !U43 || (kblocks != 5 && kblocks != 3) || Cin <= 0 || (Cin & 7) || Cout <= 0 || (Cout & 31) || !g6d_aligned16(U43)) {
    g6d_set_error("corr2d_wino43_multi: bad args (1..4 map sizes, 15x15 = 5 blocks or 9x9 = 3 blocks, Cin % 8 == 0, Cout % 32 == 0)"); return G6D_EINVAL;
  }
  W43Args a = {};
  const float* in0 = segs[0].in; float* f0 = segs[0].out;
  for (int k = 0; k < nseg; ++k) {
    const G6dCorrSeg& g = segs[k];
    if (!g.in || !g.out || g.N <= 0 || g.H <= 0 || g.W <= 0 || (g.ld_in & 3) || g.ld_in < Cin || g.ld_in != segs[0].ld_in || g.ld_out < Cout ||
        !g6d_aligned16(g.in)) {
      g6d_set_error("corr2d_wino43_multi: bad map (all maps share ld_in)"); return G6D_EINVAL;
    }
    if (g.in < in0) in0 = g.in;
    if (g.out < f0) f0 = g.out;
  }
  a.in = in0; a.U = U43; a.bias = nullptr; a.out_full = f0; a.out_pool = nullptr;
  a.Cin = Cin; a.Cout = Cout; a.relu = 0; a.D = 1; a.nseg = nseg;
  for (int k = 0; k < nseg; ++k) {
    const G6dCorrSeg& g = segs[k];
    const long long io = g.in - in0, fo = g.out - f0;
    if (io + (long long)g.N * g.H * g.W * g.ld_in >= (1ll << 29) || fo + (long long)g.N * g.H * g.W * g.ld_out >= (1ll << 31)) {
      g6d_set_error("corr2d_wino43_multi: maps must lie within 2^29 floats of each other (allocate them from one buffer)"); return G6D_EINVAL;
    }
    a.seg[k] = W43Seg{0, g.N, g.H, g.W, 0, 0, (int)io, (int)fo, 0, g.ld_in, g.ld_out, 0};
  }
  return w43_run(a, 0, kblocks * kblocks, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

// ---- the conv family on the F(4x4,3x3) kernel (called by g6d_conv_igemm when G6dConv.weight_wino43 is set): kernel (1,3,3) on 2-D maps
// or (3,3,3), stride 1, "same" padding, Cin % 8 == 0, Cout % 64 == 0, fp32, no multiplier prologue, no LeakyReLU
bool g6d_wino43_eligible(const G6dConv& d) {
  const bool on = g6d_knob(G6D_KNOB_CONV_WINO43) != 0;
  if (!on || !d.weight_wino43 || d.math_mode != 0 || d.mul || d.in_image_mod > 0 || d.mul_group_images > 0) return false;
  const bool k2 = d.kd == 1 && d.Di == 1 && d.pd == 0, k3 = d.kd == 3 && d.pd == 1;
  if (!(k2 || k3) || d.kh != 3 || d.kw != 3 || d.ph != 1 || d.pw != 1 || d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  if ((d.Cin & 7) || (d.Cout & 63) || d.Hi < 8 || d.Wi < 8 || d.out_act > 1 || d.split_k > 1) return false;
  if (d.stats && d.stat_rows_per_group > 0 && d.stat_rows_per_group % (d.Do * d.Ho * d.Wo)) return false;
  if (d.in_scale && (d.Cin > 256 || (d.Cin & 3))) return false;      // affine prologue: tested up to 256 input channels
  if (!g6d_aligned16(d.weight_wino43)) return false;
  // the launch's own hard limits (w43_run would refuse): such a layer falls through to the F(2x2,3x3) / implicit-GEMM kernels and
  // g6d_conv_plan reports that family
  const long long quarters = (long long)d.N * d.Di * ((d.Hi + 7) / 8) * ((d.Wi + 7) / 8);
  const int mode = !d.in_scale ? 0 : (d.in_affine_per_n ? 2 : 1);
  if (w43_limits(quarters, (long long)d.N * d.Di * d.Hi * d.Wi * d.ld_in, d.kd, d.Cin, d.Cout, mode)) return false;
  if ((long long)d.N * d.Do * d.Ho * d.Wo * d.ld_out >= (1ll << 31)) return false;      // 32-bit output offsets
  return (long long)d.N * d.Di * d.Hi * d.Wi * d.ld_in < (1ll << 29);
}

int g6d_wino43_launch(const G6dConv& d, hipStream_t stream) {
  W43Args a = {};
  a.in = d.in; a.U = d.weight_wino43; a.bias = d.bias; a.out_full = d.out; a.out_pool = nullptr;
  a.D = d.Di; a.Cin = d.Cin; a.Cout = d.Cout; a.relu = d.out_act == 1;
  a.nseg = 1;
  a.seg[0] = W43Seg{0, d.N * d.Di, d.Hi, d.Wi, 0, 0, 0, 0, 0, d.ld_in, d.ld_out, 0};
  a.in_scale = d.in_scale; a.in_shift = d.in_shift; a.in_relu = d.in_relu;
  a.stats = d.stats; a.stats_div = d.stat_rows_per_group > 0 ? d.stat_rows_per_group / (d.Do * d.Ho * d.Wo) : 0;
  a.aff_div = d.in_affine_per_n;
  if (d.fin_scale)
    a.fin = G6dFin{d.fin_scale, d.fin_shift, reinterpret_cast<int*>(d.fin_counter), d.stats, 1.0 / d.fin_count, d.fin_eps, d.fin_groups * d.Cout};
  const int mode = !d.in_scale ? 0 : (d.in_affine_per_n ? 2 : 1);
  return w43_run(a, mode, d.kd, d.workspace, d.workspace_bytes, stream);
}
