// Selector similarity kernels: the query x (reference x rotation) feature product of network/selector.py:183-195 is
// never materialised.  The reference cache [D][HW][C] (D = rfn*an hypotheses) is streamed once per query with
// 16-byte coalesced loads; each wavefront owns one (hypothesis, location) row of C channels and reduces the
// per-location cosine score with shuffles.  The statistics InstanceNorm3d(512) needs over the product
// (selector.py:28,49,63) come from two query-independent sums prepared at load time:
//     mean_c = (1/N) sum_hw q_c(hw) R1_c(hw),  E[x^2]_c = (1/N) sum_hw q_c(hw)^2 R2_c(hw),
//     R1 = sum_d r, R2 = sum_d r^2,  N = D*HW.
#include "g6d_common.h"
#include <stdlib.h>

namespace {

__global__ void ref_sums_kernel(const float* __restrict__ refs, int D, int HWC, double* __restrict__ r1,
                                double* __restrict__ r2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HWC) return;
  double s1 = 0, s2 = 0;
  for (int d = 0; d < D; ++d) { double v = refs[(size_t)d * HWC + i]; s1 += v; s2 += v * v; }
  r1[i] = s1; r2[i] = s2;
}

// block = 64 channels x 4 position lanes; fp64 accumulation, LDS combine
__global__ void __launch_bounds__(256) prod_affine_kernel(const float* __restrict__ que, const double* __restrict__ r1,
                                                          const double* __restrict__ r2, int HW, int C, double inv_n,
                                                          double eps, float* __restrict__ scale,
                                                          float* __restrict__ shift) {
  __shared__ double sm[4][64], se[4][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double m = 0, e = 0;
  if (c < C)
    for (int p = pl; p < HW; p += 4) {
      const double q = que[(size_t)p * C + c];
      m += q * r1[(size_t)p * C + c];
      e += q * q * r2[(size_t)p * C + c];
    }
  sm[pl][cl] = m; se[pl][cl] = e;
  __syncthreads();
  if (pl == 0 && c < C) {
    m = (sm[0][cl] + sm[1][cl] + sm[2][cl] + sm[3][cl]) * inv_n;
    e = (se[0][cl] + se[1][cl] + se[2][cl] + se[3][cl]) * inv_n;
    double var = e - m * m; if (var < 0) var = 0;
    const double rs = 1.0 / sqrt(var + eps);
    scale[c] = (float)rs; shift[c] = (float)(-m * rs);
  }
}

// rows = D*HW; one wave per row, 4 rows in flight per wave.
__global__ void __launch_bounds__(256) scan_kernel(const float* __restrict__ que, const float* __restrict__ refs,
                                                   int rows, int HW, int C, float* __restrict__ score_map) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int row0 = wave * 4; row0 < rows; row0 += nwaves * 4) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = row0 + u;
      if (row < rows) {
        const float* r = refs + (size_t)row * C;
        const float* q = que + (size_t)(row % HW) * C;
        for (int c = lane * 4; c < C; c += 256) {
          f32x4 rv = *reinterpret_cast<const f32x4*>(r + c);
          f32x4 qv = *reinterpret_cast<const f32x4*>(q + c);
          s[u] += rv[0] * qv[0] + rv[1] * qv[1] + rv[2] * qv[2] + rv[3] * qv[3];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float t = wave_sum(s[u]);
      if (lane == 0 && row0 + u < rows) score_map[row0 + u] = t;
    }
  }
}

// vps[d] = sum_hw S * (S / max_hw S); one wave per hypothesis.
__global__ void __launch_bounds__(64) vps_kernel(const float* __restrict__ score_map, int HW, float* __restrict__ vps) {
  const int d = blockIdx.x, lane = threadIdx.x;
  const float* s = score_map + (size_t)d * HW;
  float mx = -INFINITY;
  for (int p = lane; p < HW; p += 64) mx = fmaxf(mx, s[p]);
  mx = wave_max(mx);
  float acc = 0.f;
  for (int p = lane; p < HW; p += 64) { float v = s[p]; acc += v * (v / mx); }
  acc = wave_sum(acc);
  if (lane == 0) vps[d] = acc;
}

// All pyramid levels of a BATCH of queries in ONE launch (network/selector.py:165-175 takes [qn,...] queries): block kinds
//   scan blocks    (level l, hypothesis d, part pt): rows [64 pt, 64 pt + 64) of refs_l[d] — 16 waves x 4 rows, 16-byte non-temporal
//                  loads, 128 KB of reference rows requested per block — against the same rows of ALL qn queries (L2 resident):
//                  the reference cache is streamed once per BATCH, and level 0 (HW = 256) is cut into four parts so that 256 CUs
//                  are evenly loaded (round 2: one 512 KB block per level-0 hypothesis, 320 of them on 256 CUs).  Scores go to
//                  score_maps[l][q][d][row];
//   affine blocks  (query q, level l, 16 channels): InstanceNorm affine of the never-materialised product, 64 position lanes, fp64.
// The largest level comes first.  vps_levels_kernel then reduces every (q, l, d) score row: vps = sum_hw S * (S / max_hw S).
struct SelLevel { const float* que; const float* refs; const double* r1; const double* r2; float* score_map; int HW, parts, blk0, pad; };
struct SelArgs { SelLevel lv[3]; int nlev, qn, D, C, scan_blocks, dc; double inv_dg, eps; float* vps; float* scale; float* shift; };   // dc: hypotheses per ROWQ unit
#define SEL_MAX_HW 1024
#define SEL_MAX_QN 8

// QN = compile-time bound of the batch (1, 2, 4, 8): the score accumulators of a wave are 4 rows x QN registers
// ROWQ (C == 512, batches): a wave owns ONE location row of a level and a chunk of `dc` hypotheses; the row of every query sits in
// registers (2 x 16 bytes per lane and query) and the wave streams the dc reference rows of that location past them, four 2 KB
// rows in flight.  Without it (the layout above) every block re-reads the rows of all QN queries from L2 for each hypothesis — at
// QN = 8 that is 8x the reference bytes through L2 and the launch drops from 0.54 to 0.14 of the HBM rate (measured, round 3).
template <int QN, bool ROWQ>
__global__ void __launch_bounds__(ROWQ ? 512 : 1024) selector_levels_kernel(const SelArgs a) {
  constexpr int WAVES = ROWQ ? 8 : 16, NPL = WAVES * 4;          // ROWQ: 512-thread blocks (256 registers per lane: QN = 8 needs ~140)
  __shared__ double sm[64][16 + 1], se[64][16 + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int b = blockIdx.x;
  if (ROWQ && b < a.scan_blocks) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k) l = (k < a.nlev && b >= a.lv[k].blk0) ? k : l;
    const SelLevel& L = a.lv[l];
    const int HW = L.HW, qn = a.qn, D = a.D;
    const int u = (b - L.blk0) * WAVES + wave;                   // unit = (hypothesis chunk, row): consecutive waves -> consecutive rows
    const int dch = u / HW, row = u - dch * HW;
    const int d0 = dch * a.dc;
    if (d0 >= D) return;
    f32x4 qv[QN][2];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const float* qr = L.que + ((size_t)min(q, qn - 1) * HW + row) * 512 + lane * 4;
      qv[q][0] = *reinterpret_cast<const f32x4*>(qr); qv[q][1] = *reinterpret_cast<const f32x4*>(qr + 256);
    }
    const float* rbase = L.refs + (size_t)row * 512 + lane * 4;
    const size_t dstride = (size_t)HW * 512;
    const int dn = min(a.dc, D - d0);
    for (int dd = 0; dd < dn; dd += 4) {
      f32x4 rv[4][2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {                                // hypotheses beyond the chunk re-read its last one (discarded)
        const float* r = rbase + (size_t)(d0 + min(dd + k, dn - 1)) * dstride;
        rv[k][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(r));
        rv[k][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(r + 256));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // the QN dot products of this reference row, reduced over the wave TOGETHER: every exchange step halves the number of
        // values a lane carries (lane bit 5 keeps the odd / even query of a pair, bit 4 of the next pair, ...), so QN = 8 values cost
        // 4 + 2 + 1 + 3 cross-lane exchanges instead of 8 x 6 (the exchanges go through the LDS crossbar and bounded this kernel)
        float v[QN];
#pragma unroll
        for (int q = 0; q < QN; ++q)
          v[q] = rv[k][0][0] * qv[q][0][0] + rv[k][0][1] * qv[q][0][1] + rv[k][0][2] * qv[q][0][2] + rv[k][0][3] * qv[q][0][3] +
                 rv[k][1][0] * qv[q][1][0] + rv[k][1][1] * qv[q][1][1] + rv[k][1][2] * qv[q][1][2] + rv[k][1][3] * qv[q][1][3];
        int bit = 32;
#pragma unroll
        for (int nv = QN; nv > 1; nv >>= 1, bit >>= 1) {
          const bool hi = (lane & bit) != 0;
#pragma unroll
          for (int i = 0; i < nv / 2; ++i) {
            const float keep = hi ? v[2 * i + 1] : v[2 * i], send = hi ? v[2 * i] : v[2 * i + 1];
            v[i] = keep + __shfl_xor(send, bit, 64);
          }
        }
        float t = v[0];
        for (int o = bit; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        // lane bits 5, 4, 3 (as far as used) spell the query index, least significant first
        int qi = 0;
#pragma unroll
        for (int sft = 0, nv = QN; nv > 1; nv >>= 1, ++sft) qi |= ((lane >> (5 - sft)) & 1) << sft;
        if ((lane & (64 / QN - 1)) == 0 && qi < qn && dd + k < dn) L.score_map[((size_t)qi * D + d0 + dd + k) * HW + row] = t;
      }
    }
    return;
  }
  if (!ROWQ && b < a.scan_blocks) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k) l = (k < a.nlev && b >= a.lv[k].blk0) ? k : l;
    const SelLevel& L = a.lv[l];
    b -= L.blk0;
    const int d = b / L.parts, pt = b - d * L.parts;
    const int HW = L.HW, C = a.C, qn = a.qn;
    const int row0 = pt * 64 + wave * 4;
    if (row0 >= HW) return;
    const float* refs = L.refs + ((size_t)d * HW + row0) * C;
    const size_t qstride = (size_t)HW * C;
    float s[4][QN];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < QN; ++q) s[u][q] = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      f32x4 rv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)      // rows beyond HW re-read the last row (discarded below): unconditional loads pipeline
        rv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(refs + (size_t)min(u, HW - 1 - row0) * C + c));
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        if (q < qn) {
          const float* qr = L.que + q * qstride + (size_t)row0 * C + c;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(qr + (size_t)min(u, HW - 1 - row0) * C);
            s[u][q] += rv[u][0] * qv[0] + rv[u][1] * qv[1] + rv[u][2] * qv[2] + rv[u][3] * qv[3];
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      if (q < qn) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float t = wave_sum(s[u][q]);
          if (lane == 0 && row0 + u < HW) L.score_map[((size_t)q * a.D + d) * HW + row0 + u] = t;
        }
      }
    }
    return;
  }
  b -= a.scan_blocks;
  const int groups = (a.C + 15) / 16;
  const int q = b / (a.nlev * groups), r = b - q * (a.nlev * groups);
  const int l = r / groups, cg = r - l * groups;
  const SelLevel& L = a.lv[l];
  const float* que = L.que + (size_t)q * L.HW * a.C;
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = cg * 16 + cl;
  double m = 0, e = 0;
  if (c < a.C)
    for (int p = pl; p < L.HW; p += NPL) {
      const double qv = que[(size_t)p * a.C + c];
      m += qv * L.r1[(size_t)p * a.C + c];
      e += qv * qv * L.r2[(size_t)p * a.C + c];
    }
  sm[pl][cl] = m; se[pl][cl] = e;
  __syncthreads();
  if (pl == 0 && c < a.C) {
    m = 0; e = 0;
#pragma unroll
    for (int k = 0; k < NPL; ++k) { m += sm[k][cl]; e += se[k][cl]; }
    const double inv_n = a.inv_dg / (double)L.HW;
    m *= inv_n; e *= inv_n;
    double var = e - m * m; if (var < 0) var = 0;
    const double rs = 1.0 / sqrt(var + a.eps);
    a.scale[((size_t)q * a.nlev + l) * a.C + c] = (float)rs; a.shift[((size_t)q * a.nlev + l) * a.C + c] = (float)(-m * rs);
  }
}

// vps[q][l][d] = sum_hw S * (S / max_hw S) (network/selector.py:192-195); one wave per (q, l, d) score row.
__global__ void __launch_bounds__(256) vps_levels_kernel(const SelArgs a) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int per_q = a.nlev * a.D;
  if (w >= a.qn * per_q) return;
  const int q = w / per_q, r = w - q * per_q, l = r / a.D, d = r - l * a.D;
  const SelLevel& L = a.lv[l];
  const float* s = L.score_map + ((size_t)q * a.D + d) * L.HW;
  float mx = -INFINITY;
  for (int p = lane; p < L.HW; p += 64) mx = fmaxf(mx, s[p]);
  mx = wave_max(mx);
  float acc = 0.f;
  for (int p = lane; p < L.HW; p += 64) { const float v = s[p]; acc += v * (v / mx); }
  acc = wave_sum(acc);
  if (lane == 0) a.vps[w] = acc;
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

// network/selector.py:183-195 + 28,49,63 for all pyramid levels of a batch of qn <= 8 queries at once.  Per level l < nlev (<= 3):
// que[l] [qn][HW_l][C], refs[l] [D][HW_l][C], r1[l] / r2[l] [HW_l][C] (g6d_selector_ref_sums), score_maps[l] [qn][D][HW_l] (written;
// the cosine score maps of selector.py:192) -> vps [qn][nlev][D], scale / shift [qn][nlev][C] (the InstanceNorm3d affine of the
// never-materialised product over Dg * HW_l values; Dg = global hypothesis count, = D unless the references are sharded).
extern "C" int g6d_selector_levels(int nlev, int qn, const float* const* que, const float* const* refs, const double* const* r1,
                                   const double* const* r2, const int* HW, int D, int Dg, int C, double eps, float* const* score_maps,
                                   float* vps, float* scale, float* shift, g6d_stream_t stream) {
  if (nlev < 1 || nlev > 3 || qn < 1 || qn > SEL_MAX_QN || !que || !refs || !r1 || !r2 || !HW || !score_maps || !vps || !scale || !shift ||
      D <= 0 || Dg <= 0 || C <= 0 || (C & 3)) {
    g6d_set_error("selector_levels: bad args (1 <= qn <= 8, score_maps required)"); return G6D_EINVAL;
  }
  SelArgs a = {};
  a.nlev = nlev; a.qn = qn; a.D = D; a.C = C; a.inv_dg = 1.0 / (double)Dg; a.eps = eps; a.vps = vps; a.scale = scale; a.shift = shift;
  int blk = 0;
  for (int l = 0; l < nlev; ++l) {
    if (!que[l] || !refs[l] || !r1[l] || !r2[l] || !score_maps[l] || HW[l] <= 0 || HW[l] > SEL_MAX_HW || !g6d_aligned16(que[l]) ||
        !g6d_aligned16(refs[l]) || (long long)D * HW[l] > (1ll << 30)) {
      g6d_set_error("selector_levels: bad level (HW <= 1024, 16-byte aligned operands)"); return G6D_EINVAL;
    }
    const int parts = (HW[l] + 63) / 64;
    a.lv[l] = SelLevel{que[l], refs[l], r1[l], r2[l], score_maps[l], HW[l], parts, blk, 0};
    blk += D * parts;
  }
  // query rows in registers (ROWQ) for batches with C == 512 (knob sel_rowq = 0: never, 1: also for a single query)
  const int rowq_env = (int)g6d_knob(G6D_KNOB_SEL_ROWQ);
  const bool rowq = C == 512 && (rowq_env < 0 ? qn > 1 : rowq_env == 1 || (rowq_env != 0 && qn > 1));
  if (rowq) {
    // hypotheses per unit: all units are equal-sized, so the launch takes ceil(blocks / resident blocks) rounds of `dc` rows — pick the
    // chunk count with the fewest row-rounds (D = 320: 12 chunks of 27 = 504 blocks = ONE round of the 512 resident 8-wave blocks)
    long long rows_all = 0;
    for (int l = 0; l < nlev; ++l) rows_all += HW[l];
    int best_nch = 1; long long best_cost = -1;
    for (int nch = 1; nch <= 64 && nch <= D; ++nch) {
      const int dc = (D + nch - 1) / nch;
      const long long blocks_ = (rows_all * nch + 7) / 8, rounds = (blocks_ + 511) / 512;
      const long long cost = rounds * (dc + 2);                       // + per-unit set-up (query rows into registers)
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_nch = nch; }
    }
    a.dc = (D + best_nch - 1) / best_nch;
    const int nch = (D + a.dc - 1) / a.dc;
    blk = 0;
    for (int l = 0; l < nlev; ++l) { a.lv[l].blk0 = blk; blk += (HW[l] * nch + 7) / 8; }
  }
  a.scan_blocks = blk;
  const int blocks = blk + qn * nlev * ((C + 15) / 16);
#define SEL_LAUNCH(QN_, RQ_) hipLaunchKernelGGL((selector_levels_kernel<QN_, RQ_>), dim3(blocks), dim3(RQ_ ? 512 : 1024), 0, STREAM(stream), a)
  if (rowq) {
    if (qn == 1) SEL_LAUNCH(1, true); else if (qn == 2) SEL_LAUNCH(2, true); else if (qn <= 4) SEL_LAUNCH(4, true); else SEL_LAUNCH(8, true);
  } else {
    if (qn == 1) SEL_LAUNCH(1, false); else if (qn == 2) SEL_LAUNCH(2, false); else if (qn <= 4) SEL_LAUNCH(4, false); else SEL_LAUNCH(8, false);
  }
#undef SEL_LAUNCH
  int rc = g6d_check_launch("selector_levels");
  if (rc != G6D_OK) return rc;
  hipLaunchKernelGGL(vps_levels_kernel, dim3((qn * nlev * D + 3) / 4), dim3(256), 0, STREAM(stream), a);
  return g6d_check_launch("selector_vps_levels");
}

extern "C" int g6d_selector_ref_sums(const float* refs, int D, int HW, int C, double* r1, double* r2, g6d_stream_t stream) {
  if (!refs || !r1 || !r2 || D <= 0 || HW <= 0 || C <= 0) { g6d_set_error("selector_ref_sums: bad args"); return G6D_EINVAL; }
  const int n = HW * C;
  hipLaunchKernelGGL(ref_sums_kernel, dim3((n + 255) / 256), dim3(256), 0, STREAM(stream), refs, D, n, r1, r2);
  return g6d_check_launch("selector_ref_sums");
}

extern "C" int g6d_selector_prod_affine(const float* que, const double* r1, const double* r2, int D, int HW, int C,
                                        double eps, float* scale, float* shift, g6d_stream_t stream) {
  if (!que || !r1 || !r2 || !scale || !shift || D <= 0 || HW <= 0 || C <= 0) { g6d_set_error("selector_prod_affine: bad args"); return G6D_EINVAL; }
  hipLaunchKernelGGL(prod_affine_kernel, dim3((C + 63) / 64), dim3(256), 0, STREAM(stream), que, r1, r2, HW, C,
                     1.0 / ((double)D * HW), eps, scale, shift);
  return g6d_check_launch("selector_prod_affine");
}

extern "C" int g6d_selector_scan(const float* que, const float* refs, int D, int HW, int C, float* score_map, float* vps,
                                 g6d_stream_t stream) {
  if (!que || !refs || !score_map || !vps || D <= 0 || HW <= 0 || C <= 0 || (C & 3) || !g6d_aligned16(que) || !g6d_aligned16(refs)) {
    g6d_set_error("selector_scan: bad args"); return G6D_EINVAL;
  }
  const long long rows = (long long)D * HW;
  if (rows > (1ll << 30)) { g6d_set_error("selector_scan: too many rows"); return G6D_EINVAL; }
  long long blocks = (rows + 15) / 16;            // 4 waves x 4 rows per pass
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scan_kernel, dim3((int)blocks), dim3(256), 0, STREAM(stream), que, refs, (int)rows, HW, C, score_map);
  int rc = g6d_check_launch("selector_scan");
  if (rc != G6D_OK) return rc;
  hipLaunchKernelGGL(vps_kernel, dim3(D), dim3(64), 0, STREAM(stream), score_map, HW, vps);
  return g6d_check_launch("selector_vps");
}
