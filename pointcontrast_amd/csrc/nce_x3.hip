// PointInfoNCE on the bf16 matrix cores at fp32 accuracy (three-term operand split, x3_split.h) -- the matrix-core
// form of loss.hip's nce_fwd_kernel / nce_bwd_kernel (pc/lib/ddp_trainer.py:419-426: torch.mm(q, k^T) / T,
// CrossEntropyLoss over the rows, and its autograd).
//
// The n x n logits are never written to memory.  Every GEMM of the loss has one dimension of 32 (the feature width):
//   S^T[b][a] = sum_d other[b][d] own[a][d]          contraction = the feature width = ONE v_mfma_f32_16x16x32_bf16
//   d_own[a][d] = sum_b W[a][b] other[b][d]          W = (softmax - I) * scale, formed in registers from S^T
// and the D layout of the first (lane (i, kk) holds column a = i, rows b = 4 kk + r) IS the A layout of the second
// (lane (i, kk) supplies row a = i and eight contraction slots) once the second's B operand enumerates the 32 rows of
// a chunk in the order slot(kk, e) = 16 (e >> 2) + 4 kk + (e & 3): the softmax weights go from accumulator to operand
// without leaving the lane.
//
// nce_pack_kernel splits q and k ONCE per call into both operand images (row fragments for the first GEMM, slot-ordered
// column fragments for the second), zero padded to 128 rows: the main kernels issue no split for their inputs, a
// fragment is one 16-byte load, and padded rows need no masks in the backward (their column fragments are zero).
// A workgroup owns 128 rows (32 per wave, resident as B fragments) and walks a share of the other operand in chunks of
// 32 rows staged once per workgroup through a two-deep LDS ring (one barrier per chunk).  gridDim.y workgroups share a
// row tile; the last of them to arrive (common.h: arrive_last) merges their partials in split order: the log-sum-exp
// states, lse and the loss in the forward (two hand-overs: the splits of a tile, then the tiles), the gradient tiles of dq
// and dk in the backward -- no finishing launches, no float atomics.  Forward + backward: 4 launches (pack + forward,
// pack + backward) instead of 8, 88 instead of 168 us at n = 4096, c = 32; what is left is hand-over latency, not matrix
// work (profiles/r04n_loss_block_nce_matrix_cores_and_hardest_host.txt: chunk loops 11.5 / 27 us, finish 14 / 19 us).
// -DPCMI_NCE_DIAG_NO_LOOP / -DPCMI_NCE_DIAG_NO_FINISH compile one of the two out (timing only, wrong results).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "internal.h"
#include "x3_split.h"

namespace pcmi {

namespace {

constexpr int kOwn = 128;    // rows of the own operand per workgroup (4 waves x 2 groups of 16)
constexpr int kChunk = 32;   // rows of the other operand per step
constexpr int kStage = 768;  // 16-byte pieces of a staged chunk: 384 row-fragment + 384 column-fragment pieces
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
constexpr float kFloor = -1e30f;  // "no key seen yet" in the log2-domain running maximum (finite: no inf - inf)

// pieces per operand image: n_pad * 12 (row fragments: [16-row tile][term][lane]; column fragments:
// [32-row chunk][term][16-column tile][lane]); images of one call: q.rows | q.cols | k.rows | k.cols
__host__ __device__ inline int64_t image_pieces(int64_t n_pad) { return n_pad * 12; }

// One thread per (image, 16-byte piece position): eight fp32 values -> the three bf16 pieces of one fragment.
__global__ __launch_bounds__(256) void nce_pack_kernel(const float* __restrict__ q, const float* __restrict__ k, int64_t n,
                                                       int64_t n_pad, int c, u32x4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = n_pad * 4;
  if (idx >= 4 * per) return;
  const int which = (int)(idx / per);
  const int64_t rem = idx - which * per;
  const float* __restrict__ x = which < 2 ? q : k;
  u32x4* __restrict__ img = out + which * image_pieces(n_pad);
  const int lane = (int)(rem & 63), i = lane & 15, kk = lane >> 4;
  v4f x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
  int64_t piece;
  if ((which & 1) == 0) {  // row fragment: row = 16 tile + i, channels 8 kk .. 8 kk + 7
    const int64_t tile = rem >> 6, row = tile * 16 + i;
    if (row < n && 8 * kk < c) {
      x0 = *reinterpret_cast<const v4f*>(x + row * c + 8 * kk);
      x1 = *reinterpret_cast<const v4f*>(x + row * c + 8 * kk + 4);
    }
    piece = tile * 3 * 64 + lane;
    u32x4 h, m, l;
    split3(x0, x1, h, m, l);
    img[piece] = h;
    img[piece + 64] = m;
    img[piece + 128] = l;
  } else {  // column fragment: column d = 16 ct + i, rows 32 chunk + 16 (e >> 2) + 4 kk + (e & 3)
    const int ct = (int)((rem >> 6) & 1);
    const int64_t chunk = rem >> 7;
    const int d = 16 * ct + i;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t b = chunk * kChunk + 16 * (e >> 2) + 4 * kk + (e & 3);
      const float v = (b < n && d < c) ? x[b * c + d] : 0.f;
      if (e < 4) x0[e] = v; else x1[e - 4] = v;
    }
    piece = chunk * 3 * 2 * 64 + ct * 64 + lane;
    u32x4 h, m, l;
    split3(x0, x1, h, m, l);
    img[piece] = h;
    img[piece + 128] = m;
    img[piece + 256] = l;
  }
}

#define PCMI_NCE_MFMA(ACC, A, B) \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0)
// six products, the small ones first (spconv_x3.hip), back to back on one accumulator; A / B = {h, m, l}
#define PCMI_NCE_SIX(ACC, A, B)  \
  PCMI_NCE_MFMA(ACC, A[2], B[0]); \
  PCMI_NCE_MFMA(ACC, A[0], B[2]); \
  PCMI_NCE_MFMA(ACC, A[1], B[1]); \
  PCMI_NCE_MFMA(ACC, A[1], B[0]); \
  PCMI_NCE_MFMA(ACC, A[0], B[1]); \
  PCMI_NCE_MFMA(ACC, A[0], B[0])

// The staged chunk of the other operand: thread t copies pieces t, t + 256, t + 512 of [row fragments | column
// fragments]; the backward of the keys also stages the chunk's 32 lse values.  EVERY load and store of the ring is
// unconditional (the caller clamps the chunk index; the last step re-loads its own chunk into the slot nobody reads
// any more): with a load or its wait under a condition, the compiler's wait-count pass carries "maybe pending" around
// the loop and answers with s_waitcnt vmcnt(0) right behind the next step's loads -- an L2 round trip per chunk.
struct StageRegs {
  u32x4 r[3];
  float lse;
};
template <bool COLS, bool LSE>
__device__ __forceinline__ void stage_load(const u32x4* __restrict__ rows, const u32x4* __restrict__ cols,
                                           const float* __restrict__ lse, int64_t n, int64_t chunk, int t, StageRegs& s) {
  // row fragments of the chunk's two 16-row tiles are adjacent ([tile][term][lane]); so are its column fragments
  const u32x4* pr = rows + chunk * 384;
  s.r[0] = pr[t];
  if (COLS) {
    const u32x4* pc = cols + chunk * 384;
    s.r[1] = *(t < 128 ? pr + t + 256 : pc + t - 128);
    s.r[2] = pc[t + 128];
  } else {
    s.r[1] = pr[(t & 127) + 256];
  }
  if (LSE) s.lse = lse[min(chunk * kChunk + (t & 31), n - 1)];  // (padding rows: any finite value, their column fragments are zero)
}
template <bool COLS, bool LSE>
__device__ __forceinline__ void stage_store(u32x4* __restrict__ s_buf, float* __restrict__ s_lse, int t, const StageRegs& s) {
  s_buf[t] = s.r[0];
  if (COLS) {
    s_buf[t + 256] = s.r[1];
    s_buf[t + 512] = s.r[2];
  } else {
    s_buf[(t & 127) + 256] = s.r[1];  // (threads t and t + 128 store the same piece: a store under a condition would take its
  }                                   //  load with it, to the end of the step)
  if (LSE) s_lse[t & 31] = s.lse;
}

// ---- forward: per own row the log-sum-exp state over this workgroup's share of the keys, in the log2 domain ------
// grid = (row tiles of the queries, splits).  part_m / part_l: [split][n_pad].
__global__ __launch_bounds__(256, 2) void nce_fwd_x3_kernel(const u32x4* __restrict__ img, const float* __restrict__ q,
                                                            const float* __restrict__ k, int64_t n, int64_t n_pad, int c,
                                                            float inv_T, int64_t span, float* __restrict__ part_m,
                                                            float* __restrict__ part_l, float* __restrict__ tile_loss,
                                                            unsigned* __restrict__ counters, float* __restrict__ lse,
                                                            float* __restrict__ loss) {
  __shared__ u32x4 s_st[2][384];
  __shared__ float s_red[2];
  __shared__ unsigned s_flag;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, kk = lane >> 4;
  const u32x4* __restrict__ own_rows = img;                           // q.rows
  const u32x4* __restrict__ oth_rows = img + 2 * image_pieces(n_pad);  // k.rows
  const int64_t a_base = (int64_t)blockIdx.x * kOwn + wave * 32;
  const float c1 = inv_T * kLog2e;
  u32x4 bo[2][3];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int term = 0; term < 3; ++term) bo[g][term] = own_rows[((a_base >> 4) + g) * 192 + term * 64 + lane];
  float m[2] = {kFloor, kFloor}, l[2] = {0.f, 0.f};
  const int64_t n32 = (n + kChunk - 1) / kChunk * kChunk;
  const int64_t cbeg = (int64_t)blockIdx.y * span, cend = min(n32, cbeg + span);
  const int64_t last_chunk = n32 / kChunk - 1;
  StageRegs sr;
  stage_load<false, false>(oth_rows, nullptr, nullptr, n, min(cbeg / kChunk, last_chunk), t, sr);
  stage_store<false, false>(s_st[0], nullptr, t, sr);
  __syncthreads();
  int buf = 0;
#if defined(PCMI_NCE_DIAG_NO_LOOP)
  for (int64_t b0 = cbeg; b0 < cbeg; b0 += kChunk, buf ^= 1) {
#else
  for (int64_t b0 = cbeg; b0 < cend; b0 += kChunk, buf ^= 1) {
#endif
    stage_load<false, false>(oth_rows, nullptr, nullptr, n, min(b0 / kChunk + 1, last_chunk), t, sr);
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the loads to their use at the end of the step)
    u32x4 ak[2][3];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int term = 0; term < 3; ++term) ak[tt][term] = s_st[buf][(tt * 3 + term) * 64 + lane];
    v4f sacc[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        sacc[g][tt] = v4f{0.f, 0.f, 0.f, 0.f};
        PCMI_NCE_SIX(sacc[g][tt], ak[tt], bo[g]);
      }
    const bool edge = b0 + kChunk > n;  // (wave-uniform) rows >= n of this chunk are padding: not keys
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float v[8];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = sacc[g][tt][r] * c1;
          if (edge && b0 + 16 * tt + 4 * kk + r >= n) x = kFloor;
          v[tt * 4 + r] = x;
        }
      float cm = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
      const float mn = fmaxf(m[g], cm);
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += __builtin_amdgcn_exp2f(v[e] - mn);
      l[g] = l[g] * __builtin_amdgcn_exp2f(m[g] - mn) + sum;
      m[g] = mn;
    }
    __builtin_amdgcn_sched_barrier(0);  // (... and hoists the stores, with their wait, to the top of it)
    stage_store<false, false>(s_st[buf ^ 1], nullptr, t, sr);
    __syncthreads();
  }
  // the four lane quads of a column hold disjoint key rows
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int d = 16; d < 64; d <<= 1) {
      const float om = __shfl_xor(m[g], d, 64), ol = __shfl_xor(l[g], d, 64);
      const float mn = fmaxf(m[g], om);
      l[g] = l[g] * __builtin_amdgcn_exp2f(m[g] - mn) + ol * __builtin_amdgcn_exp2f(om - mn);
      m[g] = mn;
    }
    if (kk == 0) {
      const int64_t o = (int64_t)blockIdx.y * n_pad + a_base + 16 * g + i;
      part_m[o] = m[g];
      part_l[o] = l[g];
    }
  }
  // ---- the last workgroup of the row tile: merge the splits (in split order), lse, the tile's share of the loss ----
  const int splits = (int)gridDim.y;
#if defined(PCMI_NCE_DIAG_NO_FINISH)  // timing diagnostic (no lse, no loss)
  return;
#endif
  if (!arrive_last(&counters[blockIdx.x], (unsigned)splits, &s_flag)) return;
  float contrib = 0.f;
  if (t < kOwn) {
    const int64_t a = (int64_t)blockIdx.x * kOwn + t;
    float mm = kFloor, ll = 0.f;
    for (int sp = 0; sp < splits; ++sp) {
      const float om = part_m[(int64_t)sp * n_pad + a], ol = part_l[(int64_t)sp * n_pad + a];
      const float mn = fmaxf(mm, om);
      ll = ll * __builtin_amdgcn_exp2f(mm - mn) + ol * __builtin_amdgcn_exp2f(om - mn);
      mm = mn;
    }
    if (a < n) {
      const float v = (mm + __builtin_amdgcn_logf(ll)) * kLn2;  // v_log_f32 = log2
      lse[a] = v;
      float dg = 0.f;
      for (int d = 0; d < c; ++d) dg = fmaf(q[a * c + d], k[a * c + d], dg);
      contrib = v - dg * inv_T;
    }
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) contrib += __shfl_xor(contrib, d, 64);
  if (lane == 0 && wave < 2) s_red[wave] = contrib;
  __syncthreads();
  if (t == 0) tile_loss[blockIdx.x] = s_red[0] + s_red[1];
  // ---- the last row tile to finish: the loss (tiles in order) ----
  const int tiles = (int)gridDim.x;
  if (!arrive_last(&counters[tiles], (unsigned)tiles, &s_flag)) return;
  if (t == 0) {
    float s = 0.f;
    for (int b = 0; b < tiles; ++b) s += tile_loss[b];
    *loss = s / (float)n;
  }
}

// ---- backward: d_own[a] = gs * sum_b (softmax - I) other_b ----------------------------------------------------------
// grid = (row tiles, splits, side): side 0: own = q, other = k, p = exp(s_ab - lse[a]);  side 1: own = k, other = q,
// p = exp(s_ba - lse[b]).  part: [side][split][n_pad][32] when splits > 1.
template <bool FOR_K, int C>
__device__ __forceinline__ void nce_bwd_body(const u32x4* __restrict__ own_rows, const u32x4* __restrict__ oth_rows,
                                             const u32x4* __restrict__ oth_cols, const float* __restrict__ lse, int64_t n,
                                             int64_t n_pad, float inv_T, const float* __restrict__ gscale, int64_t span,
                                             float* __restrict__ d_own, float* __restrict__ part,
                                             unsigned* __restrict__ counter, u32x4 (*s_st)[kStage], float (*s_lse)[kChunk],
                                             unsigned* s_flag) {
  constexpr int NCT = C / 16;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, kk = lane >> 4;
  const int64_t a_base = (int64_t)blockIdx.x * kOwn + wave * 32;
  const float c1 = inv_T * kLog2e;
  const float gs = (gscale ? *gscale : 1.f) * inv_T / (float)n;
  u32x4 bo[2][3];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int term = 0; term < 3; ++term) bo[g][term] = own_rows[((a_base >> 4) + g) * 192 + term * 64 + lane];
  float lo[2] = {0.f, 0.f};  // log2-domain lse of the own rows (side 0)
  if (!FOR_K) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int64_t a = a_base + 16 * g + i;
      lo[g] = lse[a < n ? a : n - 1] * kLog2e;
    }
  }
  v4f dacc[2][NCT];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) dacc[g][ct] = v4f{0.f, 0.f, 0.f, 0.f};
  const int64_t n32 = (n + kChunk - 1) / kChunk * kChunk;
  const int64_t cbeg = (int64_t)blockIdx.y * span, cend = min(n32, cbeg + span);
  const int64_t last_chunk = n32 / kChunk - 1;
  StageRegs sr;
  stage_load<true, FOR_K>(oth_rows, oth_cols, lse, n, min(cbeg / kChunk, last_chunk), t, sr);
  stage_store<true, FOR_K>(s_st[0], s_lse[0], t, sr);
  __syncthreads();
  int buf = 0;
#if defined(PCMI_NCE_DIAG_NO_LOOP)
  for (int64_t b0 = cbeg; b0 < cbeg; b0 += kChunk, buf ^= 1) {
#else
  for (int64_t b0 = cbeg; b0 < cend; b0 += kChunk, buf ^= 1) {
#endif
    stage_load<true, FOR_K>(oth_rows, oth_cols, lse, n, min(b0 / kChunk + 1, last_chunk), t, sr);
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the loads to their use at the end of the step)
    float lb[2][4];  // log2-domain lse of the other rows (side 1)
    if (FOR_K) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const v4f v = *reinterpret_cast<const v4f*>(&s_lse[buf][16 * tt + 4 * kk]);
#pragma unroll
        for (int r = 0; r < 4; ++r) lb[tt][r] = v[r] * kLog2e;
      }
    }
    u32x4 ak[2][3];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int term = 0; term < 3; ++term) ak[tt][term] = s_st[buf][(tt * 3 + term) * 64 + lane];
    v4f sacc[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        sacc[g][tt] = v4f{0.f, 0.f, 0.f, 0.f};
        PCMI_NCE_SIX(sacc[g][tt], ak[tt], bo[g]);
      }
    // the column fragments of the chunk: [term][column tile][lane] behind the 384 row-fragment pieces
    u32x4 bk[NCT][3];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int term = 0; term < 3; ++term) bk[ct][term] = s_st[buf][384 + (term * 2 + ct) * 64 + lane];
    const bool diag = b0 == a_base;  // (wave-uniform) the chunk holding this wave's own rows
    // (wave-uniform) the chunk reaches past row n: its padding rows b >= n have zero operands, so s = 0 and
    // p = exp2(-lse) -- with lse below about -88 (unnormalised features, a tiny T) that is inf, the three-term split of
    // inf has NaN terms, and NaN x the zero column fragments of the padding rows is NaN in the gradient of REAL rows.
    // The forward masks these rows with kFloor; here their weight is forced to zero.
    const bool edge = b0 + kChunk > n;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      v4f w0, w1;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(sacc[g][tt][r], c1, -(FOR_K ? lb[tt][r] : lo[g])));
          float w = p * gs;
          if (diag && 16 * tt + 4 * kk + r == 16 * g + i) w = (p - 1.f) * gs;
          if (edge && b0 + 16 * tt + 4 * kk + r >= n) w = 0.f;
          if (tt == 0) w0[r] = w; else w1[r] = w;
        }
      u32x4 wt[3];
      split3(w0, w1, wt[0], wt[1], wt[2]);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) { PCMI_NCE_SIX(dacc[g][ct], wt, bk[ct]); }
    }
    __builtin_amdgcn_sched_barrier(0);  // (... and hoists the stores, with their wait, to the top of it)
    stage_store<true, FOR_K>(s_st[buf ^ 1], s_lse[buf ^ 1], t, sr);
    __syncthreads();
  }
  // ---- D[row = 4 kk + r][col = i] of every 16 x 16 tile ----
  const int splits = (int)gridDim.y;
  if (splits == 1) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t a = a_base + 16 * g + 4 * kk + r;
        if (a < n) {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) d_own[a * C + 16 * ct + i] = dacc[g][ct][r];
        }
      }
    return;
  }
  float* __restrict__ mine = part + (int64_t)blockIdx.y * n_pad * 32;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t a = a_base + 16 * g + 4 * kk + r;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) mine[a * 32 + 16 * ct + i] = dacc[g][ct][r];
    }
#if defined(PCMI_NCE_DIAG_NO_FINISH)
  return;
#endif
  if (!arrive_last(counter, (unsigned)splits, s_flag)) return;
  // the last workgroup of the row tile: the shares in split order, four channels per thread
  constexpr int F4 = C / 4;  // float4 per row
  for (int e = t; e < kOwn * F4; e += 256) {
    const int row = e / F4, c4 = e % F4;
    const int64_t a = (int64_t)blockIdx.x * kOwn + row;
    if (a >= n) continue;
    v4f s = *reinterpret_cast<const v4f*>(part + a * 32 + c4 * 4);
    for (int sp = 1; sp < splits; ++sp) s += *reinterpret_cast<const v4f*>(part + ((int64_t)sp * n_pad + a) * 32 + c4 * 4);
    *reinterpret_cast<v4f*>(d_own + a * C + c4 * 4) = s;
  }
}

template <int C>
__global__ __launch_bounds__(256, 2) void nce_bwd_x3_kernel(const u32x4* __restrict__ img, const float* __restrict__ lse,
                                                            int64_t n, int64_t n_pad, float inv_T,
                                                            const float* __restrict__ gscale, int64_t span,
                                                            float* __restrict__ dq, float* __restrict__ dk,
                                                            float* __restrict__ part, unsigned* __restrict__ counters) {
  __shared__ u32x4 s_st[2][kStage];
  __shared__ __attribute__((aligned(16))) float s_lse[2][kChunk];
  __shared__ unsigned s_flag;
  const int64_t ip = image_pieces(n_pad);
  const int64_t side_part = (int64_t)gridDim.y * n_pad * 32;
  if (blockIdx.z == 0)
    nce_bwd_body<false, C>(img, img + 2 * ip, img + 3 * ip, lse, n, n_pad, inv_T, gscale, span, dq, part,
                           &counters[blockIdx.x], s_st, s_lse, &s_flag);
  else
    nce_bwd_body<true, C>(img + 2 * ip, img, img + ip, lse, n, n_pad, inv_T, gscale, span, dk, part + side_part,
                          &counters[gridDim.x + blockIdx.x], s_st, s_lse, &s_flag);
}

struct NcePlan {
  int64_t n_pad, span;
  int tiles, splits;
};
// splits: about two workgroups per CU over both sides of the backward (the forward uses the same decomposition with
// half the workgroups), at least four chunks per workgroup
NcePlan nce_x3_plan(int64_t n) {
  NcePlan p;
  p.n_pad = ceil_div(n, kOwn) * kOwn;
  p.tiles = (int)(p.n_pad / kOwn);
  const int64_t chunks = ceil_div(n, kChunk);
  const int64_t want = ceil_div((int64_t)num_cu(), p.tiles);
  p.splits = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, ceil_div(chunks, 4)), 32));
  p.span = ceil_div(chunks, p.splits) * kChunk;
  p.splits = (int)ceil_div(chunks * kChunk, p.span);
  return p;
}

}  // namespace

bool nce_x3_on() {
  static const bool on = [] {
    const char* e = getenv("PCMI_NCE_X3");
    return !(e && e[0] == '0');
  }();
  return on;
}

// workspace: the four operand images, then the split partials (forward: 2 x [splits][n_pad] + tile losses; backward:
// 2 x [splits][n_pad][32])
size_t nce_x3_workspace_bytes(int64_t n) {
  const NcePlan p = nce_x3_plan(n);
  const size_t images = (size_t)4 * image_pieces(p.n_pad) * sizeof(u32x4);
  const size_t fwd = ((size_t)2 * p.splits * p.n_pad + p.tiles) * sizeof(float);
  const size_t bwd = p.splits > 1 ? (size_t)2 * p.splits * p.n_pad * 32 * sizeof(float) : 0;
  return images + std::max(fwd, bwd) + 256;
}

static int nce_x3_pack(const float* q, const float* k, int64_t n, int c, const NcePlan& p, u32x4* img, hipStream_t st) {
  const int64_t threads = 4 * p.n_pad * 4;
  nce_pack_kernel<<<(unsigned)ceil_div(threads, 256), 256, 0, st>>>(q, k, n, p.n_pad, c, img);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int nce_x3_fwd(const float* q, const float* k, int64_t n, int c, float inv_T, float* lse, float* loss, void* ws, hipStream_t st) {
  const NcePlan p = nce_x3_plan(n);
  u32x4* img = (u32x4*)ws;
  float* pm = (float*)(img + 4 * image_pieces(p.n_pad));
  float* pl = pm + (size_t)p.splits * p.n_pad;
  float* tile_loss = pl + (size_t)p.splits * p.n_pad;
  unsigned* counters = stream_counters(st, (size_t)p.tiles + 1);
  if (!counters) return PCMI_ERR_HIP;
  if (int rc = nce_x3_pack(q, k, n, c, p, img, st)) return rc;
  nce_fwd_x3_kernel<<<dim3((unsigned)p.tiles, (unsigned)p.splits), 256, 0, st>>>(img, q, k, n, p.n_pad, c, inv_T, p.span, pm, pl,
                                                                                  tile_loss, counters, lse, loss);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int nce_x3_bwd(const float* q, const float* k, const float* lse, int64_t n, int c, float inv_T, const float* gscale, float* dq,
               float* dk, void* ws, hipStream_t st) {
  const NcePlan p = nce_x3_plan(n);
  u32x4* img = (u32x4*)ws;
  float* part = (float*)(img + 4 * image_pieces(p.n_pad));
  unsigned* counters = stream_counters(st, (size_t)2 * p.tiles);
  if (!counters) return PCMI_ERR_HIP;
  if (int rc = nce_x3_pack(q, k, n, c, p, img, st)) return rc;
  const dim3 grid((unsigned)p.tiles, (unsigned)p.splits, 2);
  if (c == 16)
    nce_bwd_x3_kernel<16><<<grid, 256, 0, st>>>(img, lse, n, p.n_pad, inv_T, gscale, p.span, dq, dk, part, counters);
  else
    nce_bwd_x3_kernel<32><<<grid, 256, 0, st>>>(img, lse, n, p.n_pad, inv_T, gscale, p.span, dq, dk, part, counters);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

}  // namespace pcmi
