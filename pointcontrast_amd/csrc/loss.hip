// Contrastive-loss blocks, row gather/scatter, and the fused SGD step.
//
//  * PointInfoNCE: the n x n logits q.k^T/T are produced tile by tile (64 x 64 per
//    workgroup step, 4 x 4 per lane out of LDS), consumed by an online log-sum-exp and never
//    written to memory; row reductions run across the 16 lanes that share a row
//    (__shfl_xor inside a quarter wave).  The backward recomputes the tile and contracts it
//    with the other operand out of LDS.
//  * Hardest-contrastive: the [P, S] distance matrix is likewise only ever a register tile;
//    min / arg-min are reduced across the quarter wave; the false-negative filter is a device
//    hash set of int64 pair keys.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "internal.h"

namespace pcmi {

constexpr int kTile = 64;

// load a [64 x C] tile of `m` starting at row r0 (zero padded) into d-major LDS: s[d][row]
template <int C>
__device__ inline void load_tile_dmajor(const float* __restrict__ m, int64_t n, int64_t r0,
                                        float (*s)[kTile + 4], int t) {
  for (int e = t; e < kTile * (C / 4); e += 256) {
    const int row = e / (C / 4), c4 = e % (C / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + row < n) v = *reinterpret_cast<const float4*>(m + (r0 + row) * C + c4 * 4);
    s[c4 * 4 + 0][row] = v.x;
    s[c4 * 4 + 1][row] = v.y;
    s[c4 * 4 + 2][row] = v.z;
    s[c4 * 4 + 3][row] = v.w;
  }
}

template <int C>
__device__ inline void load_tile_rowmajor(const float* __restrict__ m, int64_t n, int64_t r0,
                                          float (*s)[C + 4], int t) {
  for (int e = t; e < kTile * (C / 4); e += 256) {
    const int row = e / (C / 4), c4 = e % (C / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + row < n) v = *reinterpret_cast<const float4*>(m + (r0 + row) * C + c4 * 4);
    *reinterpret_cast<float4*>(&s[row][c4 * 4]) = v;
  }
}

// 4x4 register tile of A_tile . B_tile^T from d-major LDS tiles
template <int C>
__device__ inline void dot_tile(const float (*sa)[kTile + 4], const float (*sb)[kTile + 4], int tr, int tc,
                                float acc[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
#pragma unroll 8
  for (int d = 0; d < C; ++d) {
    const float4 a = *reinterpret_cast<const float4*>(&sa[d][tr * 4]);
    const float4 b = *reinterpret_cast<const float4*>(&sb[d][tc * 4]);
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
  }
}

// ---- NCE forward ------------------------------------------------------------------------------------
// grid = (row tiles, key splits): a row tile of 64 queries alone is 64 workgroups at n = 4096 -- a quarter of the
// chip -- so the key tiles are divided over gridDim.y workgroups, each leaving the online log-sum-exp state (m, l)
// and the diagonal logit of its share in part_m / part_l / part_d [split][n]; nce_combine_kernel merges the splits
// (same max-rescaled combine as across the 16 lanes of a row) into lse[i] and the loss partials.
template <int C>
__global__ __launch_bounds__(256) void nce_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      int64_t n, float inv_T, int64_t span, float* __restrict__ part_m,
                                                      float* __restrict__ part_l, float* __restrict__ part_d) {
  __shared__ __attribute__((aligned(16))) float sq[C][kTile + 4];
  __shared__ __attribute__((aligned(16))) float sk[C][kTile + 4];
  __shared__ float s_part[4];
  const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
  const int64_t r0 = (int64_t)blockIdx.x * kTile;
  load_tile_dmajor<C>(q, n, r0, sq, t);
  float m[4], l[4], diag[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m[i] = -INFINITY;
    l[i] = 0.f;
    diag[i] = 0.f;
  }
  const int64_t cbeg = (int64_t)blockIdx.y * span, cend = min(n, cbeg + span);
  for (int64_t c0 = cbeg; c0 < cend; c0 += kTile) {
    __syncthreads();
    load_tile_dmajor<C>(k, n, c0, sk, t);
    __syncthreads();
    float acc[4][4];
    dot_tile<C>(sq, sk, tr, tc, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float tmax = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t col = c0 + tc * 4 + j;
        const float s = acc[i][j] * inv_T;
        acc[i][j] = s;
        if (col < n) tmax = fmaxf(tmax, s);
        if (col == r0 + tr * 4 + i) diag[i] = s;
      }
      if (tmax > m[i]) {
        l[i] *= expf(m[i] - tmax);
        m[i] = tmax;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + tc * 4 + j < n) l[i] += expf(acc[i][j] - m[i]);
    }
  }
  // combine the 16 lanes that share the rows
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const float om = __shfl_xor(m[i], d, 64), ol = __shfl_xor(l[i], d, 64);
      const float nm = fmaxf(m[i], om);
      l[i] = (m[i] == -INFINITY ? 0.f : l[i] * expf(m[i] - nm)) + (om == -INFINITY ? 0.f : ol * expf(om - nm));
      m[i] = nm;
      diag[i] += __shfl_xor(diag[i], d, 64);
    }
    const int64_t row = r0 + tr * 4 + i;
    if (row < n && tc == 0) {
      const int64_t o = (int64_t)blockIdx.y * n + row;
      part_m[o] = m[i];
      part_l[o] = l[i];
      part_d[o] = diag[i];  // non-zero only in the split that holds column `row`
    }
  }
  (void)s_part;
}

// lse[i] = log sum_j exp(s_ij) from the per-split (m, l); part[block] = sum over the block's rows of (lse_i - s_ii)
__global__ __launch_bounds__(256) void nce_combine_kernel(const float* __restrict__ part_m, const float* __restrict__ part_l,
                                                          const float* __restrict__ part_d, int splits, int64_t n,
                                                          float* __restrict__ lse, float* __restrict__ part) {
  __shared__ float s_part[4];
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float contrib = 0.f;
  if (row < n) {
    float m = -INFINITY, l = 0.f, dg = 0.f;
    for (int sp = 0; sp < splits; ++sp) {
      const float om = part_m[(int64_t)sp * n + row], ol = part_l[(int64_t)sp * n + row];
      const float nm = fmaxf(m, om);
      l = (m == -INFINITY ? 0.f : l * expf(m - nm)) + (om == -INFINITY ? 0.f : ol * expf(om - nm));
      m = nm;
      dg += part_d[(int64_t)sp * n + row];
    }
    const float v = m + logf(l);
    lse[row] = v;
    contrib = v - dg;
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) contrib += __shfl_xor(contrib, d, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = contrib;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ void sum_scale_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += part[i];
    *out = s * scale;
  }
}

// ---- NCE backward: d_own[a] = gs * sum_b (softmax - I)[.,.] other_b ------------------------------
// FOR_K == false: own = q (rows a = i), other = k;   p = exp(s_ab - lse[a])
// FOR_K == true : own = k (rows a = j), other = q;   p = exp(s_ab - lse[b])
// grid = (row tiles, splits of the other operand's tiles); with gridDim.y > 1 every split writes its share of the sum to
// d_part[split][n][C] and nce_sum_splits_kernel adds the shares in split order (deterministic).
template <int C, bool FOR_K>
__global__ __launch_bounds__(256) void nce_bwd_kernel(const float* __restrict__ own, const float* __restrict__ other,
                                                      const float* __restrict__ lse, int64_t n, float inv_T,
                                                      const float* __restrict__ gscale, int64_t span,
                                                      float* __restrict__ d_own) {
  __shared__ __attribute__((aligned(16))) float so[C][kTile + 4];
  __shared__ __attribute__((aligned(16))) float sx[C][kTile + 4];
  __shared__ __attribute__((aligned(16))) float sxr[kTile][C + 4];
  __shared__ float sw[kTile][kTile + 1];
  const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
  const int64_t r0 = (int64_t)blockIdx.x * kTile;
  const float gs = (gscale ? *gscale : 1.f) * inv_T / (float)n;
  load_tile_dmajor<C>(own, n, r0, so, t);
  // second-GEMM mapping: thread -> (row, CD consecutive channels)
  constexpr int TPR = 4;  // threads per row
  constexpr int CD = C / TPR;
  const int orow = t / TPR, od0 = (t % TPR) * CD;
  float dacc[CD];
#pragma unroll
  for (int d = 0; d < CD; ++d) dacc[d] = 0.f;
  float lse_own[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) lse_own[i] = (!FOR_K && r0 + tr * 4 + i < n) ? lse[r0 + tr * 4 + i] : 0.f;
  const int64_t cbeg = (int64_t)blockIdx.y * span, cend = min(n, cbeg + span);
  d_own += (int64_t)blockIdx.y * n * C;  // this split's share (the caller passes d_part when gridDim.y > 1)
  for (int64_t c0 = cbeg; c0 < cend; c0 += kTile) {
    __syncthreads();
    load_tile_dmajor<C>(other, n, c0, sx, t);
    load_tile_rowmajor<C>(other, n, c0, sxr, t);
    __syncthreads();
    float acc[4][4];
    dot_tile<C>(so, sx, tr, tc, acc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t b = c0 + tc * 4 + j;
      const float lse_b = (FOR_K && b < n) ? lse[b] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t a = r0 + tr * 4 + i;
        float w = 0.f;
        if (a < n && b < n) {
          const float p = expf(acc[i][j] * inv_T - (FOR_K ? lse_b : lse_own[i]));
          w = (p - (a == b ? 1.f : 0.f)) * gs;
        }
        sw[tr * 4 + i][tc * 4 + j] = w;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int b = 0; b < kTile; ++b) {
      const float w = sw[orow][b];
#pragma unroll
      for (int d = 0; d < CD; ++d) dacc[d] = fmaf(w, sxr[b][od0 + d], dacc[d]);
    }
  }
  if (r0 + orow < n) {
#pragma unroll
    for (int d = 0; d < CD; ++d) d_own[(r0 + orow) * C + od0 + d] = dacc[d];
  }
}

// out[e] = sum over the splits of part[split][e] (float4 per thread, split order)
__global__ __launch_bounds__(256) void nce_sum_splits_kernel(const float* __restrict__ part, int splits, int64_t n4,
                                                             float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n4) return;
  float4 s = reinterpret_cast<const float4*>(part)[e];
  for (int sp = 1; sp < splits; ++sp) {
    const float4 v = reinterpret_cast<const float4*>(part)[(int64_t)sp * n4 + e];
    s.x += v.x;
    s.y += v.y;
    s.z += v.z;
    s.w += v.w;
  }
  reinterpret_cast<float4*>(out)[e] = s;
}

// ---- pdist + argmin ------------------------------------------------------------------------------
// TR rows of `a` per workgroup against tiles of TS rows of `b`; a thread holds a 4 x 4 block of distances and the
// TS / 4 threads that share its rows reduce (min, first arg-min) by shuffles.  The mining of the hardest-contrastive
// loss is p = 4096 positives against s = 1024 candidates: 64-row tiles are 64 workgroups on a 256-CU chip and 16 tile
// steps each (58 us); TR = 16, TS = 256 is 256 workgroups and 4 steps.
template <int C, int ROWS>
__device__ inline void load_rows_dmajor(const float* __restrict__ m, int64_t n, int64_t r0, float (*s)[ROWS + 4], int t) {
  for (int e = t; e < ROWS * (C / 4); e += 256) {
    const int row = e / (C / 4), c4 = e % (C / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + row < n) v = *reinterpret_cast<const float4*>(m + (r0 + row) * C + c4 * 4);
    s[c4 * 4 + 0][row] = v.x;
    s[c4 * 4 + 1][row] = v.y;
    s[c4 * 4 + 2][row] = v.z;
    s[c4 * 4 + 3][row] = v.w;
  }
}

template <int C, int TR, int TS>
__global__ __launch_bounds__(256) void pdist_argmin_kernel(const float* __restrict__ a, int64_t p,
                                                           const float* __restrict__ b, int64_t s,
                                                           float* __restrict__ dmin, int32_t* __restrict__ amin) {
  static_assert(TR * TS == 64 * 64 && TS >= 64, "256 threads x (4 x 4)");
  constexpr int LPR = TS / 4;  // lanes sharing a row group (16 or 64: inside one wave)
  __shared__ __attribute__((aligned(16))) float sa[C][TR + 4];
  __shared__ __attribute__((aligned(16))) float sb[C][TS + 4];
  const int t = threadIdx.x, tr = t / LPR, tc = t % LPR;
  const int64_t r0 = (int64_t)blockIdx.x * TR;
  load_rows_dmajor<C, TR>(a, p, r0, sa, t);
  float best[4];
  int32_t bidx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    best[i] = INFINITY;
    bidx[i] = 0x7fffffff;
  }
  for (int64_t c0 = 0; c0 < s; c0 += TS) {
    __syncthreads();
    load_rows_dmajor<C, TS>(b, s, c0, sb, t);
    __syncthreads();
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
#pragma unroll 8
    for (int d = 0; d < C; ++d) {
      const float4 av4 = *reinterpret_cast<const float4*>(&sa[d][tr * 4]);
      const float4 bv4 = *reinterpret_cast<const float4*>(&sb[d][tc * 4]);
      const float av[4] = {av4.x, av4.y, av4.z, av4.w}, bv[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float df = av[i] - bv[j];
          acc[i][j] = fmaf(df, df, acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t col = c0 + tc * 4 + j;
        if (col < s && acc[i][j] < best[i]) {
          best[i] = acc[i][j];
          bidx[i] = (int32_t)col;
        }
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int d = 1; d < LPR; d <<= 1) {
      const float ob = __shfl_xor(best[i], d, 64);
      const int32_t oi = __shfl_xor(bidx[i], d, 64);
      if (ob < best[i] || (ob == best[i] && oi < bidx[i])) {
        best[i] = ob;
        bidx[i] = oi;
      }
    }
    const int64_t row = r0 + tr * 4 + i;
    if (row < p && tc == 0) {
      dmin[row] = sqrtf(best[i] + 1e-7f);
      amin[row] = bidx[i];
    }
  }
}

// ---- int64 key set ---------------------------------------------------------------------------------
__global__ void keyset_build_kernel(const int32_t* __restrict__ pairs, int64_t n, int64_t M, uint64_t* keys,
                                    uint32_t mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = (uint64_t)((int64_t)pairs[2 * i] + (int64_t)pairs[2 * i + 1] * M);
  uint32_t slot = hash_key(key) & mask;
  while (true) {
    const unsigned long long prev =
        atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)kEmptyKey, (unsigned long long)key);
    if (prev == kEmptyKey || prev == key) return;
    slot = (slot + 1) & mask;
  }
}

__global__ void keyset_mask_kernel(const uint64_t* __restrict__ keys, uint32_t mask, const int64_t* __restrict__ a,
                                   const int64_t* __restrict__ b, int64_t n, int64_t M, uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = (uint64_t)(a[i] + b[i] * M);
  uint32_t slot = hash_key(key) & mask;
  uint8_t absent = 1;
  while (true) {
    const uint64_t kk = keys[slot];
    if (kk == key) {
      absent = 0;
      break;
    }
    if (kk == kEmptyKey) break;
    slot = (slot + 1) & mask;
  }
  out[i] = absent;
}

// ---- hardest-contrastive loss values + gradients ------------------------------------------------------
// stats[0] = sum relu(|a-b|^2 - pt); [1] = sum_mask0 relu(nt - d01)^2; [2] = count mask0;
// [3] = sum_mask1 relu(nt - d10)^2; [4] = count mask1
__global__ __launch_bounds__(256) void hardest_stats_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                            int64_t p, int c, const float* __restrict__ d01,
                                                            const uint8_t* __restrict__ m0,
                                                            const float* __restrict__ d10,
                                                            const uint8_t* __restrict__ m1, float pt, float nt,
                                                            float* __restrict__ part /* [blocks][5] */) {
  __shared__ float s_red[4][5];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < p) {
    float d2 = 0.f;
    for (int d = 0; d < c; ++d) {
      const float df = f0[i * c + d] - f1[i * c + d];
      d2 = fmaf(df, df, d2);
    }
    v[0] = fmaxf(d2 - pt, 0.f);
    if (m0[i]) {
      const float h = fmaxf(nt - d01[i], 0.f);
      v[1] = h * h;
      v[2] = 1.f;
    }
    if (m1[i]) {
      const float h = fmaxf(nt - d10[i], 0.f);
      v[3] = h * h;
      v[4] = 1.f;
    }
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v[q] += __shfl_xor(v[q], d, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 5)
    part[(int64_t)blockIdx.x * 5 + threadIdx.x] =
        s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
}

__global__ void hardest_final_kernel(const float* __restrict__ part, int nblocks, int64_t p, float* __restrict__ stats,
                                     float* __restrict__ losses) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < nblocks; ++b)
    for (int q = 0; q < 5; ++q) s[q] += part[b * 5 + q];
  for (int q = 0; q < 5; ++q) stats[q] = s[q];
  losses[0] = s[0] / (float)p;
  losses[1] = (s[1] / s[2] + s[3] / s[4]) * 0.5f;
}

// one thread per (positive pair, channel)
__global__ __launch_bounds__(256) void hardest_grad_kernel(
    const float* __restrict__ f0, const float* __restrict__ f1, int64_t p, int c, const float* __restrict__ sub0,
    const float* __restrict__ sub1, const float* __restrict__ d01, const int32_t* __restrict__ i01,
    const uint8_t* __restrict__ m0, const float* __restrict__ d10, const int32_t* __restrict__ i10,
    const uint8_t* __restrict__ m1, float pt, float nt, const float* __restrict__ stats,
    const float* __restrict__ gl, float* __restrict__ g0, float* __restrict__ g1) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p * c) return;
  const int64_t i = idx / c;
  const int d = (int)(idx - i * c);
  // positive term: every lane of the row recomputes |a-b|^2 (c is small)
  float d2 = 0.f;
  for (int e = 0; e < c; ++e) {
    const float df = f0[i * c + e] - f1[i * c + e];
    d2 = fmaf(df, df, d2);
  }
  const float a = f0[idx], b = f1[idx];
  const float up_pos = gl[0], up_neg = gl[1];
  float ga = 0.f, gb = 0.f;
  if (d2 - pt > 0.f) {
    const float gpos = up_pos * 2.f * (a - b) / (float)p;
    ga += gpos;
    gb -= gpos;
  }
  if (m0[i]) {
    const float h = nt - d01[i];
    if (h > 0.f) {
      const int32_t q = i01[i];
      const float coef = up_neg * -2.f * h / stats[2] * 0.5f;  // d neg / d D01
      const float gd = coef * (a - sub1[(int64_t)q * c + d]) / d01[i];
      ga += gd;  // (the matching -gd of the mined row: hardest_gsub_kernel)
    }
  }
  if (m1[i]) {
    const float h = nt - d10[i];
    if (h > 0.f) {
      const int32_t q = i10[i];
      const float coef = up_neg * -2.f * h / stats[4] * 0.5f;
      const float gd = coef * (b - sub0[(int64_t)q * c + d]) / d10[i];
      gb += gd;
    }
  }
  g0[idx] = ga;
  g1[idx] = gb;
}

// Gradient of the mined negatives, WITHOUT float atomics: several positives may have mined the same row q, and the sum
// of their contributions must not depend on the order the hardware retires atomics in (the step is bit-reproducible,
// tests/test_gpu_fullsize.py).  One wave per positive i: it OWNS row q = imin[i] if no earlier active positive mined q,
// and then adds the contributions of all positives that mined q in increasing i (lane = channel).  p^2 / 64 wave steps
// over three L1-resident arrays: a few microseconds at p = 4096.
//   side 0: fpos = f0, sub = sub1 (row q of the other cloud's candidates), dmin / imin / mask = d01 / i01 / m0, den = stats[2]
__global__ __launch_bounds__(256) void hardest_gsub_kernel(const float* __restrict__ fpos, int64_t p, int c,
                                                           const float* __restrict__ sub, const float* __restrict__ dmin,
                                                           const int32_t* __restrict__ imin, const uint8_t* __restrict__ mask,
                                                           float nt, const float* __restrict__ stats, int den_slot,
                                                           const float* __restrict__ gl, float* __restrict__ gsub) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= p) return;  // (wave-uniform)
  auto active = [&](int64_t j) { return j < p && mask[j] && nt - dmin[j] > 0.f; };
  if (!active(i)) return;
  const int32_t q = imin[i];
  constexpr int kScanU = 4;  // 64 x kScanU candidates per step, their loads issued together
  auto mined = [&](int64_t j, int64_t lim) {  // (lim >= 1; three independent loads, no short-circuit between them)
    const int64_t jc = j < lim ? j : lim - 1;
    const bool mk = mask[jc] != 0;
    const float dm = dmin[jc];
    const int32_t im = imin[jc];
    return (j < lim) & mk & (nt - dm > 0.f) & (im == q);
  };
  bool dup = false;
  for (int64_t b = 0; b < i; b += 64 * kScanU) {
    bool h[kScanU];
#pragma unroll
    for (int u = 0; u < kScanU; ++u) h[u] = mined(b + u * 64 + lane, i);
#pragma unroll
    for (int u = 0; u < kScanU; ++u) dup |= h[u];
  }
  if (__any(dup)) return;  // an earlier positive owns row q
  const float up_neg = gl[1], den = stats[den_slot];
  for (int d0 = 0; d0 < c; d0 += 64) {
    const int d = d0 + lane;
    const float sq = d < c ? sub[(int64_t)q * c + d] : 0.f;
    float sum = 0.f;
    for (int64_t b = i; b < p; b += 64 * kScanU) {
      bool h[kScanU];
#pragma unroll
      for (int u = 0; u < kScanU; ++u) h[u] = mined(b + u * 64 + lane, p);
#pragma unroll
      for (int u = 0; u < kScanU; ++u) {
        uint64_t m = __ballot(h[u]);
        while (m) {
          const int64_t jj = b + u * 64 + __builtin_ctzll(m);
          m &= m - 1;
          const float hh = nt - dmin[jj];
          const float coef = up_neg * -2.f * hh / den * 0.5f;
          if (d < c) sum -= coef * (fpos[jj * c + d] - sq) / dmin[jj];
        }
      }
    }
    if (d < c) gsub[(int64_t)q * c + d] += sum;
  }
}

// ---- rows gather / scatter-add ------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ src, int64_t src_ld, const int64_t* __restrict__ idx,
                                   int64_t n, int c4, float* __restrict__ dst, int64_t dst_ld) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * c4) return;
  const int64_t r = e / c4;
  const int col = (int)(e - r * c4);
  *reinterpret_cast<float4*>(dst + r * dst_ld + col * 4) =
      *reinterpret_cast<const float4*>(src + idx[r] * src_ld + col * 4);
}

// ---- rows of a list grouped by a key, without sorting and without float atomics -----------------------------------
// (scatter_add_rows_kernel: rows sharing a destination.)
// The first row with a key OWNS the group and adds the later rows' contributions in increasing row order, so the sum does
// not depend on the order the hardware would retire atomics in (the step is bit-reproducible, tests/test_gpu_fullsize.py).
// Every row has to see every key: a workgroup stages the keys through LDS (kGroupPass per pass) and each of its four
// waves compares a staged key with the keys of FOUR rows at once -- n / 64 LDS reads per wave and pass, where the first
// form of this kernel spent n / 64 global loads PER ROW (the same 13 us at n = 4096, but growing with n^2: 4 x per
// doubling against 2 x here).  hardest_gsub_kernel keeps the per-row form: its groups are few and large (hundreds of
// positives share a popular negative), the owner's arithmetic is what it waits for, and the staged form measured
// slower there (79 against 45 us, profiles/r04n_*).
constexpr int kGroupRows = 16;    // rows per workgroup (4 per wave)
constexpr int kGroupPass = 4096;  // keys staged per pass

// One staged pass [p0, p0 + len): for wave-row q (row r0 + q, key target[q]) sets dup[q] if an earlier row has the key and
// collects the later rows that have it, in increasing order, in the wave's list q; a full list -- and, from the caller,
// whatever is left at the end -- goes to consume(q, rows, count), which fetches the rows' contributions with all its
// loads in flight and adds them in list order (not a chain of dependent round trips when a group is large).  s_keys holds kGroupPass entries (the ones past the list's end never equal a target).
constexpr int kListCap = 64;
template <class Key, class Consume>
__device__ __forceinline__ void scan_same_key(const Key* __restrict__ s_keys, int len, int64_t p0, int64_t r0,
                                              const Key (&target)[4], bool (&dup)[4], int (&cnt)[4],
                                              int32_t (*list)[kListCap], int lane, Consume&& consume) {
  for (int b4 = 0; b4 < len; b4 += 256) {
    Key v4[4];  // four LDS reads in flight (the entries past `len` are staged and never match)
#pragma unroll
    for (int u = 0; u < 4; ++u) v4[u] = s_keys[b4 + u * 64 + lane];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const Key v = v4[u];
      const int64_t j0 = p0 + b4 + u * 64;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (dup[q]) continue;  // (wave-uniform) not the owner, or no row at all: nothing left to find out, nothing to add
        uint64_t m = __ballot(v == target[q]);
        if (m == 0) continue;  // (wave-uniform; the common case)
        const int64_t r = r0 + q;
        if (j0 < r) {  // matches in front of the row: it is not the owner
          const uint64_t before = r - j0 >= 64 ? ~0ull : ((1ull << (r - j0)) - 1ull);
          if (m & before) dup[q] = true;
        }
        if (!dup[q] && j0 + 63 > r) {  // matches behind it
          const uint64_t upto = r < j0 ? 0ull : (r - j0 >= 63 ? ~0ull : ((2ull << (r - j0)) - 1ull));  // bits <= r
          m &= ~upto;
          while (m) {
            if (lane == 0) list[q][cnt[q]] = (int32_t)(j0 + __builtin_ctzll(m));
            m &= m - 1;
            if (++cnt[q] == kListCap) {
              consume(q, list[q], kListCap);
              cnt[q] = 0;
            }
          }
        }
      }
    }
  }
}

// dst[idx[r]] += src[r] (idx may repeat: two positives matched to the same row); lane = column.  See scan_same_key.
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, int64_t src_ld,
                                                               const int64_t* __restrict__ idx, int64_t n, int c,
                                                               float* __restrict__ dst, int64_t dst_ld) {
  __shared__ int64_t s_idx[kGroupPass];
  __shared__ int32_t s_list[4][4][kListCap];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * kGroupRows + wave * 4;
  int64_t target[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) target[q] = r0 + q < n ? idx[r0 + q] : -1;
  for (int c0 = 0; c0 < c; c0 += 64) {
    const int col = c0 + lane;
    float sum[4], old[4];
    bool dup[4];
    int cnt[4] = {0, 0, 0, 0};
    auto consume = [&](int q, const int32_t* rows, int m) {
      constexpr int B = 16;
      for (int k0 = 0; k0 < m; k0 += B) {
        float a[B];
#pragma unroll
        for (int u = 0; u < B; ++u) a[u] = col < c ? src[(int64_t)rows[min(k0 + u, m - 1)] * src_ld + col] : 0.f;
#pragma unroll
        for (int u = 0; u < B; ++u)
          if (k0 + u < m) sum[q] += a[u];
      }
    };
    // every global access this row block needs is requested up front (the kernel is a chain of memory round trips, not
    // of instructions): own rows, the destination rows (read-modify-write), the first pass of the index array
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool live = r0 + q < n && col < c;
      sum[q] = live ? src[(r0 + q) * src_ld + col] : 0.f;
      old[q] = live ? dst[target[q] * dst_ld + col] : 0.f;
      dup[q] = r0 + q >= n;
    }
    for (int64_t p0 = 0; p0 < n; p0 += kGroupPass) {
      int64_t stage[kGroupPass / 256];
#pragma unroll
      for (int u = 0; u < kGroupPass / 256; ++u) stage[u] = p0 + u * 256 + t < n ? idx[p0 + u * 256 + t] : -2;  // (-2: never a target)
      __syncthreads();
#pragma unroll
      for (int u = 0; u < kGroupPass / 256; ++u) s_idx[u * 256 + t] = stage[u];
      __syncthreads();
      scan_same_key(s_idx, (int)min((int64_t)kGroupPass, n - p0), p0, r0, target, dup, cnt, s_list[wave], lane, consume);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (cnt[q]) consume(q, s_list[wave][q], cnt[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (!dup[q] && col < c) dst[target[q] * dst_ld + col] = old[q] + sum[q];
  }
}

// ---- softmax cross-entropy with ignore label (downstream reuse of the backbone) ---------------------------------
// loss = mean over the rows whose label != ignore of (logsumexp(x_r) - x_r[label_r]) -- torch.nn.CrossEntropyLoss(
// ignore_index=...) as used by downstream/semseg/lib/train.py:64,124.  One thread per row (c is the number of classes:
// a wave reads 64 consecutive rows, i.e. one contiguous run); per-block (sum, count) partials, finished by the
// last-arriving workgroup in block order (deterministic).  out[0] = loss, out[1] = number of counted rows.
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ x, int64_t ld, int64_t n, int c,
                                                     const int32_t* __restrict__ label, int ignore,
                                                     float* __restrict__ part, unsigned* counter, float* __restrict__ out) {
  __shared__ float s_red[4][2];
  __shared__ unsigned s_last;
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float v[2] = {0.f, 0.f};
  if (r < n) {
    const int32_t lb = label[r];
    if (lb != ignore && (lb < 0 || lb >= c)) {
      // torch.nn.CrossEntropyLoss raises for a label that is neither a class nor the ignore index; a kernel cannot
      // raise, so the loss (and every gradient behind it) becomes NaN instead of the row being dropped silently
      v[0] = __builtin_nanf("");
      v[1] = 1.f;
    } else if (lb != ignore) {
      const float* xr = x + r * ld;
      float m = -INFINITY;
      for (int j = 0; j < c; ++j) m = fmaxf(m, xr[j]);
      float se = 0.f;
      for (int j = 0; j < c; ++j) se += __expf(xr[j] - m);
      v[0] = m + __logf(se) - xr[lb];
      v[1] = 1.f;
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v[q] += __shfl_xor(v[q], d, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 2)
    part[(int64_t)blockIdx.x * 2 + threadIdx.x] =
        s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
  if (!arrive_last(counter, gridDim.x, &s_last)) return;
  if (threadIdx.x == 0) {
    float s0 = 0.f, s1 = 0.f;
    for (unsigned b = 0; b < gridDim.x; ++b) {
      s0 += part[2 * b];
      s1 += part[2 * b + 1];
    }
    out[0] = s1 > 0.f ? s0 / s1 : 0.f;
    out[1] = s1;
  }
}

// dx[r][j] = gloss / count * (softmax(x_r)[j] - [j == label_r])  (zero rows for ignored labels)
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ x, int64_t ld, int64_t n, int c,
                                                     const int32_t* __restrict__ label, int ignore,
                                                     const float* __restrict__ stats, const float* __restrict__ gloss,
                                                     float* __restrict__ dx, int64_t dx_ld) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const int32_t lb = label[r];
  float* dr = dx + r * dx_ld;
  if (lb != ignore && (lb < 0 || lb >= c)) {  // see ce_fwd_kernel: a mis-mapped label poisons, it is not dropped
    for (int j = 0; j < c; ++j) dr[j] = __builtin_nanf("");
    return;
  }
  if (lb == ignore || stats[1] <= 0.f) {
    for (int j = 0; j < c; ++j) dr[j] = 0.f;
    return;
  }
  const float* xr = x + r * ld;
  float m = -INFINITY;
  for (int j = 0; j < c; ++j) m = fmaxf(m, xr[j]);
  float se = 0.f;
  for (int j = 0; j < c; ++j) se += __expf(xr[j] - m);
  const float scale = gloss[0] / stats[1], inv = 1.f / se;
  for (int j = 0; j < c; ++j) dr[j] = scale * (__expf(xr[j] - m) * inv - (j == lb ? 1.f : 0.f));
}

// ---- SGD ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v,
                                                  int64_t n, float lr, float mu, float wd, float gscale, float gcoef) {
  // gcoef = 1 - dampening (1 on the first step: torch initialises the buffer with the gradient itself)
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 wv = reinterpret_cast<float4*>(w)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    vv.x = mu * vv.x + gcoef * (gscale * gv.x + wd * wv.x);
    vv.y = mu * vv.y + gcoef * (gscale * gv.y + wd * wv.y);
    vv.z = mu * vv.z + gcoef * (gscale * gv.z + wd * wv.z);
    vv.w = mu * vv.w + gcoef * (gscale * gv.w + wd * wv.w);
    wv.x -= lr * vv.x;
    wv.y -= lr * vv.y;
    wv.z -= lr * vv.z;
    wv.w -= lr * vv.w;
    reinterpret_cast<float4*>(v)[i] = vv;
    reinterpret_cast<float4*>(w)[i] = wv;
  }
  if (blockIdx.x == 0) {
    const int64_t i = n4 * 4 + threadIdx.x;
    if (i < n) {
      const float vv = mu * v[i] + gcoef * (gscale * g[i] + wd * w[i]);
      v[i] = vv;
      w[i] -= lr * vv;
    }
  }
}

static uint32_t keyset_cap(size_t bytes) {
  uint32_t cap = 1;
  while ((size_t)cap * 2 * sizeof(uint64_t) <= bytes) cap <<= 1;
  return cap;
}

}  // namespace pcmi

using namespace pcmi;

extern "C" {

int pcmi_gather_rows(const float* src, int64_t src_ld, const int64_t* idx, int64_t n, int c, float* dst,
                     int64_t dst_ld, pcmi_stream_t stream) {
  PCMI_REQUIRE(src && idx && dst && c % 4 == 0 && src_ld % 4 == 0 && dst_ld % 4 == 0 && (uintptr_t)src % 16 == 0 &&
                   (uintptr_t)dst % 16 == 0,
               PCMI_ERR_INVALID, "gather_rows: bad argument");
  if (n == 0) return PCMI_OK;
  gather_rows_kernel<<<dim3((unsigned)ceil_div(n * (c / 4), 256)), 256, 0, as_stream(stream)>>>(src, src_ld, idx, n, c / 4, dst,
                                                                                             dst_ld);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int pcmi_scatter_add_rows(const float* src, int64_t src_ld, const int64_t* idx, int64_t n, int c, float* dst,
                          int64_t dst_ld, pcmi_stream_t stream) {
  PCMI_REQUIRE(src && idx && dst && c > 0, PCMI_ERR_INVALID, "scatter_add_rows: bad argument");
  if (n == 0) return PCMI_OK;
  scatter_add_rows_kernel<<<dim3((unsigned)ceil_div(n, kGroupRows)), 256, 0, as_stream(stream)>>>(src, src_ld, idx, n, c, dst, dst_ld);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

}  // extern "C"

namespace pcmi {
// workgroups per row tile: enough to put ~2 workgroups on every CU, at most one split per other-operand tile
static int nce_splits(int64_t n) {
  const int64_t nb = ceil_div(n, kTile);
  const int64_t want = ceil_div(2 * (int64_t)num_cu(), nb);
  return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, nb), 32));
}
static int64_t nce_span(int64_t n, int splits) { return ceil_div(ceil_div(n, kTile), splits) * kTile; }
}  // namespace pcmi

extern "C" {

size_t pcmi_nce_workspace_bytes(int64_t n, int c) {
  const size_t sp = 32;  // upper bound of nce_splits
  // forward: 3 x [splits][n] + the block partials; backward: [splits][n][c] per operand
  const size_t valu = std::max(sp * (size_t)n * 3 * sizeof(float), sp * (size_t)n * c * sizeof(float)) +
                      (size_t)(ceil_div(n, 256) + 1) * sizeof(float) + 1024;
  return std::max(valu, nce_x3_workspace_bytes(n));
}

int pcmi_nce_fwd(const float* q, const float* k, int64_t n, int c, float inv_T, float* lse, float* loss, void* ws,
                 size_t ws_bytes, pcmi_stream_t stream) {
  PCMI_REQUIRE(q && k && lse && loss && n > 0, PCMI_ERR_INVALID, "nce_fwd: bad argument");
  PCMI_REQUIRE(c == 16 || c == 32, PCMI_ERR_UNSUPPORTED, "nce_fwd: feature width %d not in {16,32}", c);
  PCMI_REQUIRE(ws && ws_bytes >= pcmi_nce_workspace_bytes(n, c), PCMI_ERR_WORKSPACE, "nce_fwd: workspace too small");
  PCMI_REQUIRE((uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0, PCMI_ERR_INVALID, "nce_fwd: q/k must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  if (nce_x3_on()) return nce_x3_fwd(q, k, n, c, inv_T, lse, loss, ws, st);  // matrix-core form (nce_x3.hip)
  const int nb = (int)ceil_div(n, kTile), splits = nce_splits(n), nb2 = (int)ceil_div(n, 256);
  const int64_t span = nce_span(n, splits);
  float* pm = (float*)ws;
  float* pl = pm + (size_t)splits * n;
  float* pd = pl + (size_t)splits * n;
  float* part = pd + (size_t)splits * n;
  const dim3 grid((unsigned)nb, (unsigned)splits);
  if (c == 16) nce_fwd_kernel<16><<<grid, 256, 0, st>>>(q, k, n, inv_T, span, pm, pl, pd);
  if (c == 32) nce_fwd_kernel<32><<<grid, 256, 0, st>>>(q, k, n, inv_T, span, pm, pl, pd);
  PCMI_LAUNCH_CHECK();
  nce_combine_kernel<<<nb2, 256, 0, st>>>(pm, pl, pd, splits, n, lse, part);
  PCMI_LAUNCH_CHECK();
  sum_scale_kernel<<<1, 64, 0, st>>>(part, nb2, 1.0f / (float)n, loss);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int pcmi_nce_bwd(const float* q, const float* k, const float* lse, int64_t n, int c, float inv_T, const float* gscale,
                 float* dq, float* dk, void* ws, size_t ws_bytes, pcmi_stream_t stream) {
  PCMI_REQUIRE(q && k && lse && dq && dk && n > 0, PCMI_ERR_INVALID, "nce_bwd: bad argument");
  PCMI_REQUIRE(c == 16 || c == 32, PCMI_ERR_UNSUPPORTED, "nce_bwd: feature width %d not in {16,32}", c);
  hipStream_t st = as_stream(stream);
  const int nb = (int)ceil_div(n, kTile), splits = nce_splits(n);
  const int64_t span = nce_span(n, splits);
  PCMI_REQUIRE(splits == 1 || (ws && ws_bytes >= pcmi_nce_workspace_bytes(n, c)), PCMI_ERR_WORKSPACE, "nce_bwd: workspace too small");
  PCMI_REQUIRE((uintptr_t)dq % 16 == 0 && (uintptr_t)dk % 16 == 0, PCMI_ERR_INVALID, "nce_bwd: dq/dk must be 16-byte aligned");
  if (nce_x3_on()) {
    PCMI_REQUIRE(ws && ws_bytes >= pcmi_nce_workspace_bytes(n, c) && (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0,
                 PCMI_ERR_WORKSPACE, "nce_bwd: workspace too small (pcmi_nce_workspace_bytes) or q/k not 16-byte aligned");
    return nce_x3_bwd(q, k, lse, n, c, inv_T, gscale, dq, dk, ws, st);
  }
  const dim3 grid((unsigned)nb, (unsigned)splits);
  const int64_t n4 = n * c / 4;
  for (int which = 0; which < 2; ++which) {  // 0: dq, 1: dk
    float* dst = which ? dk : dq;
    float* target = splits > 1 ? (float*)ws : dst;
    if (c == 16 && !which) nce_bwd_kernel<16, false><<<grid, 256, 0, st>>>(q, k, lse, n, inv_T, gscale, span, target);
    if (c == 16 && which) nce_bwd_kernel<16, true><<<grid, 256, 0, st>>>(k, q, lse, n, inv_T, gscale, span, target);
    if (c == 32 && !which) nce_bwd_kernel<32, false><<<grid, 256, 0, st>>>(q, k, lse, n, inv_T, gscale, span, target);
    if (c == 32 && which) nce_bwd_kernel<32, true><<<grid, 256, 0, st>>>(k, q, lse, n, inv_T, gscale, span, target);
    PCMI_LAUNCH_CHECK();
    if (splits > 1) {
      nce_sum_splits_kernel<<<(unsigned)ceil_div(n4, 256), 256, 0, st>>>((const float*)ws, splits, n4, dst);
      PCMI_LAUNCH_CHECK();
    }
  }
  return PCMI_OK;
}

int pcmi_pdist_argmin(const float* a, int64_t p, const float* b, int64_t s, int c, float* dmin, int32_t* amin,
                      pcmi_stream_t stream) {
  PCMI_REQUIRE(a && b && dmin && amin && p > 0 && s > 0, PCMI_ERR_INVALID, "pdist_argmin: bad argument");
  PCMI_REQUIRE(c == 16 || c == 32 || c == 64, PCMI_ERR_UNSUPPORTED, "pdist_argmin: feature width %d not in {16,32,64}", c);
  hipStream_t st = as_stream(stream);
  // 16-row tiles when 64-row tiles would leave most of the chip without a workgroup (and the candidate tile fits the LDS)
  const bool narrow = c <= 32 && ceil_div(p, 64) < 2 * (int64_t)num_cu() && s > 64;
  const int nb = (int)ceil_div(p, narrow ? 16 : 64);
  if (c == 16 && narrow) pdist_argmin_kernel<16, 16, 256><<<nb, 256, 0, st>>>(a, p, b, s, dmin, amin);
  if (c == 16 && !narrow) pdist_argmin_kernel<16, 64, 64><<<nb, 256, 0, st>>>(a, p, b, s, dmin, amin);
  if (c == 32 && narrow) pdist_argmin_kernel<32, 16, 256><<<nb, 256, 0, st>>>(a, p, b, s, dmin, amin);
  if (c == 32 && !narrow) pdist_argmin_kernel<32, 64, 64><<<nb, 256, 0, st>>>(a, p, b, s, dmin, amin);
  if (c == 64) pdist_argmin_kernel<64, 64, 64><<<(unsigned)ceil_div(p, 64), 256, 0, st>>>(a, p, b, s, dmin, amin);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

size_t pcmi_keyset_bytes(int64_t n_keys) {
  size_t cap = 1024;
  while ((int64_t)cap < 2 * n_keys) cap <<= 1;
  return cap * sizeof(uint64_t);
}

int pcmi_keyset_build(const int32_t* pairs, int64_t n, int64_t M, void* set, size_t set_bytes, pcmi_stream_t stream) {
  PCMI_REQUIRE(pairs && set && set_bytes >= pcmi_keyset_bytes(n), PCMI_ERR_WORKSPACE, "keyset_build: set buffer too small");
  hipStream_t st = as_stream(stream);
  const uint32_t cap = keyset_cap(set_bytes);
  PCMI_HIP_CHECK(hipMemsetAsync(set, 0xFF, (size_t)cap * sizeof(uint64_t), st));
  if (n > 0) {
    keyset_build_kernel<<<dim3((unsigned)ceil_div(n, 256)), 256, 0, st>>>(pairs, n, M, (uint64_t*)set, cap - 1);
    PCMI_LAUNCH_CHECK();
  }
  return PCMI_OK;
}

int pcmi_keyset_mask_absent(const void* set, size_t set_bytes, const int64_t* a, const int64_t* b, int64_t n, int64_t M,
                            uint8_t* mask, pcmi_stream_t stream) {
  PCMI_REQUIRE(set && a && b && mask, PCMI_ERR_INVALID, "keyset_mask_absent: bad argument");
  if (n == 0) return PCMI_OK;
  const uint32_t cap = keyset_cap(set_bytes);
  keyset_mask_kernel<<<dim3((unsigned)ceil_div(n, 256)), 256, 0, as_stream(stream)>>>((const uint64_t*)set, cap - 1, a, b, n, M,
                                                                                   mask);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

size_t pcmi_hardest_workspace_bytes(int64_t p) { return (size_t)(ceil_div(p, 256) * 5 + 8) * sizeof(float) + 256; }

int pcmi_hardest_loss_fwd(const float* posF0, const float* posF1, int64_t p, int c, const float* d01min,
                          const uint8_t* mask0, const float* d10min, const uint8_t* mask1, float pos_thresh,
                          float neg_thresh, float* losses, float* stats, void* ws, size_t ws_bytes,
                          pcmi_stream_t stream) {
  PCMI_REQUIRE(posF0 && posF1 && d01min && mask0 && d10min && mask1 && losses && stats && p > 0 && c > 0, PCMI_ERR_INVALID,
               "hardest_loss_fwd: bad argument");
  PCMI_REQUIRE(ws && ws_bytes >= pcmi_hardest_workspace_bytes(p), PCMI_ERR_WORKSPACE, "hardest_loss_fwd: workspace too small");
  hipStream_t st = as_stream(stream);
  const int nb = (int)ceil_div(p, 256);
  float* part = (float*)ws;
  hardest_stats_kernel<<<nb, 256, 0, st>>>(posF0, posF1, p, c, d01min, mask0, d10min, mask1, pos_thresh, neg_thresh, part);
  PCMI_LAUNCH_CHECK();
  hardest_final_kernel<<<1, 64, 0, st>>>(part, nb, p, stats, losses);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int pcmi_hardest_loss_bwd(const float* posF0, const float* posF1, int64_t p, const float* subF0, const float* subF1, int c,
                          const float* d01min, const int32_t* d01ind, const uint8_t* mask0, const float* d10min,
                          const int32_t* d10ind, const uint8_t* mask1, float pos_thresh, float neg_thresh,
                          const float* stats, const float* gl, float* dposF0, float* dposF1, float* dsubF0,
                          float* dsubF1, pcmi_stream_t stream) {
  PCMI_REQUIRE(posF0 && posF1 && subF0 && subF1 && d01min && d01ind && mask0 && d10min && d10ind && mask1 && stats && gl &&
                   dposF0 && dposF1 && dsubF0 && dsubF1 && p > 0 && c > 0,
               PCMI_ERR_INVALID, "hardest_loss_bwd: bad argument");
  hipStream_t st = as_stream(stream);
  hardest_grad_kernel<<<dim3((unsigned)ceil_div(p * c, 256)), 256, 0, st>>>(
      posF0, posF1, p, c, subF0, subF1, d01min, d01ind, mask0, d10min, d10ind, mask1, pos_thresh, neg_thresh, stats, gl,
      dposF0, dposF1);
  PCMI_LAUNCH_CHECK();
  // the mined rows' gradients (+= into the caller's zeroed dsubF1 / dsubF0), in positive order: no float atomics
  const dim3 gw((unsigned)ceil_div(p, 4));
  hardest_gsub_kernel<<<gw, 256, 0, st>>>(posF0, p, c, subF1, d01min, d01ind, mask0, neg_thresh, stats, 2, gl, dsubF1);
  PCMI_LAUNCH_CHECK();
  hardest_gsub_kernel<<<gw, 256, 0, st>>>(posF1, p, c, subF0, d10min, d10ind, mask1, neg_thresh, stats, 4, gl, dsubF0);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

size_t pcmi_softmax_ce_workspace_bytes(int64_t n) { return (size_t)ceil_div(std::max<int64_t>(n, 1), 256) * 2 * sizeof(float) + 256; }

int pcmi_softmax_ce_fwd(const float* logits, int64_t ld, int64_t n, int c, const int32_t* labels, int ignore_label,
                        float* out2, void* ws, size_t ws_bytes, pcmi_stream_t stream) {
  PCMI_REQUIRE(logits && labels && out2 && n > 0 && c > 0 && ld >= c, PCMI_ERR_INVALID, "softmax_ce_fwd: bad argument");
  PCMI_REQUIRE(ws && ws_bytes >= pcmi_softmax_ce_workspace_bytes(n), PCMI_ERR_WORKSPACE, "softmax_ce_fwd: workspace too small");
  hipStream_t st = as_stream(stream);
  unsigned* counter = stream_counters(st, 1);
  if (!counter) return PCMI_ERR_HIP;
  ce_fwd_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(logits, ld, n, c, labels, ignore_label, (float*)ws, counter, out2);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int pcmi_softmax_ce_bwd(const float* logits, int64_t ld, int64_t n, int c, const int32_t* labels, int ignore_label,
                        const float* out2, const float* gloss, float* dlogits, int64_t d_ld, pcmi_stream_t stream) {
  PCMI_REQUIRE(logits && labels && out2 && gloss && dlogits && n > 0 && c > 0 && ld >= c && d_ld >= c, PCMI_ERR_INVALID,
               "softmax_ce_bwd: bad argument");
  ce_bwd_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, as_stream(stream)>>>(logits, ld, n, c, labels, ignore_label, out2, gloss,
                                                                           dlogits, d_ld);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int pcmi_sgd_step_dampened(float* w, const float* g, float* v, int64_t n, float lr, float momentum, float dampening,
                           float weight_decay, float grad_scale, int first_step, pcmi_stream_t stream) {
  PCMI_REQUIRE(w && g && v && n >= 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)g % 16 == 0 && (uintptr_t)v % 16 == 0,
               PCMI_ERR_INVALID, "sgd_step: buffers must be 16-byte aligned");
  PCMI_REQUIRE(dampening >= 0.f && dampening <= 1.f, PCMI_ERR_INVALID, "sgd_step: dampening %g outside [0, 1]", (double)dampening);
  if (n == 0) return PCMI_OK;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n / 4, 256), 256 * 8));
  sgd_kernel<<<grid, 256, 0, as_stream(stream)>>>(w, g, v, n, lr, momentum, weight_decay, grad_scale,
                                                  first_step ? 1.f : 1.f - dampening);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int pcmi_sgd_step(float* w, const float* g, float* v, int64_t n, float lr, float momentum, float weight_decay,
                  float grad_scale, pcmi_stream_t stream) {
  return pcmi_sgd_step_dampened(w, g, v, n, lr, momentum, 0.f, weight_decay, grad_scale, 1, stream);
}

}  // extern "C"
