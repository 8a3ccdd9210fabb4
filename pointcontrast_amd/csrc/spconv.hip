// Sparse convolution forward / backward-data on the fp32 matrix cores of gfx950.
//
// Formulation (output-stationary implicit GEMM -- no float atomics, deterministic):
//     out[r, :] = sum_k  X[ idx_k(r), : ] @ B_k          r = a tile of output rows
// where idx_k(r) comes from the neighbour table (map.nbr) or, for the one-offset-per-row
// direction of a stride-2 map, from the per-offset pair lists.  The same kernel serves
//   * 3^3 conv fwd            idx = nbr[k][r],          B_k = W[k]
//   * 3^3 conv bwd-data       idx = nbr[k][r],          B_k = W[mirror(k)]^T
//   * 2^3/s2 conv fwd         idx = child[k][r],        B_k = W[k]          (rows = coarse)
//   * 2^3/s2 conv-tr bwd-data idx = child[k][r],        B_k = W[k]^T        (rows = coarse)
//   * 2^3/s2 conv-tr fwd      pair mode, one k per tile, B = W[k]           (rows = fine)
//   * 2^3/s2 conv bwd-data    pair mode, one k per tile, B = W[k]^T         (rows = fine)
//   * 1x1 conv fwd / bwd-data idx = r,                  B = W or W^T
//
// Tiling (wave = 64, workgroup = 4 waves):
//   v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][kk = l>>5] and B[kk = l>>5][j = l&31].
//   A (gathered feature rows) goes STRAIGHT from global memory to the A operand: lane (i, h)
//   loads the float4 X[idx(i)][c0 + 8*blk + 4*h .. +3]; its four floats are the A values of
//   four consecutive MFMA steps, i.e. MFMA step s of block blk contracts over the physical
//   channel  c(s, h) = c0 + 8*blk + 4*h + s  -- any bijection works as long as B uses the same
//   one.  A half-wave therefore reads 32 rows x 16 B and both halves together cover 32 B
//   contiguous per row; successive blocks walk along the same 128-byte line.  No LDS round
//   trip, no barrier for A, absent neighbours are just zeros in the register.
//   B (weights, shared by every row of the tile) is staged through LDS in [32 x NS] chunks,
//   double buffered, one __syncthreads per chunk.
//   RW = number of 32-row groups per workgroup; the remaining 4/RW wave groups split the
//   eight-channel blocks of each chunk (intra-workgroup split of the contraction) and are
//   summed through LDS at the end.  RW=4: 128-row tiles for the big levels; RW=1: 32-row
//   tiles (+ optional split of the offset range over blockIdx.z into a partial buffer) so
//   that the small, wide levels still fill 256 CUs.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "internal.h"
#include "spconv_args.h"

namespace pcmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));  // first-class vector: HIP's float4 struct went through scratch here

constexpr int kKC = 32;  // contraction channels per staged weight chunk


// Weight chunk [KC x 32*NT] global -> registers -> LDS (NT * KC / 32 float4 per thread).  WT: B[c][n] = W[n][c]
// (backward-data).
template <int NT, bool WT, int KC = kKC>
__device__ __forceinline__ void load_b_regs(v4f (&breg)[NT * KC / 32], const float* __restrict__ wb, int64_t w_sc,
                                            int64_t w_sn, int c0, int n0, int t) {
  constexpr int NS = 32 * NT;
#pragma unroll
  for (int i = 0; i < NT * KC / 32; ++i) {
    const int e = t + i * 256;  // float4 index inside the chunk
    if (!WT) {
      const int c = e / (NS / 4), n4 = e % (NS / 4);
      breg[i] = *reinterpret_cast<const v4f*>(wb + (int64_t)(c0 + c) * w_sc + n0 + n4 * 4);
    } else {
      const int n = e / (KC / 4), c4 = e % (KC / 4);
      breg[i] = *reinterpret_cast<const v4f*>(wb + (int64_t)(n0 + n) * w_sn + c0 + c4 * 4);
    }
  }
}

template <int NT, bool WT, int LDB, int KC = kKC>
__device__ __forceinline__ void store_b_regs(const v4f (&breg)[NT * KC / 32], float* __restrict__ sb, int t) {
  constexpr int NS = 32 * NT;
#pragma unroll
  for (int i = 0; i < NT * KC / 32; ++i) {
    const int e = t + i * 256;
    if (!WT) {
      const int c = e / (NS / 4), n4 = e % (NS / 4);
      *reinterpret_cast<v4f*>(sb + c * LDB + n4 * 4) = breg[i];
    } else {
      const int n = e / (KC / 4), c4 = e % (KC / 4);
      sb[(c4 * 4 + 0) * LDB + n] = breg[i].x;
      sb[(c4 * 4 + 1) * LDB + n] = breg[i].y;
      sb[(c4 * 4 + 2) * LDB + n] = breg[i].z;
      sb[(c4 * 4 + 3) * LDB + n] = breg[i].w;
    }
  }
}

// KC: contraction channels per staged chunk (= per barrier).  32 everywhere except the narrow-slice launches of the
// small levels (NT = 1 with KC = 128, NT = 2 with KC = 64): there a 32-channel step is 16 MFMAs per wave -- 0.4 us,
// shorter than the L2 latency of the next step's gathers and weights, so the kernel ran at one memory latency per
// step (27 steps x ~0.9 us for a 256-channel conv over 1.3k rows); a deeper chunk gives every barrier 4x the matrix
// work and 4x the loads in flight.  Those variants trade the third wave per SIMD for the larger operand rings.
template <int NT, int RW, bool WT, bool PAIR, int KC = kKC>
// min 3 waves/SIMD: with this bound hipcc keeps the accumulators in plain VGPRs (<= 158 in total, no scratch);
// without it it split them into AGPRs at 170-220 registers total and 2 waves/SIMD
__global__ __launch_bounds__(256, KC > 32 ? 2 : 3) void spconv_mfma_kernel(ConvArgs a) {
  constexpr int TM = 32 * RW;        // rows per workgroup tile
  constexpr int KG = 4 / RW;         // wave groups splitting the contraction blocks
  constexpr int BPG = KC / 8 / KG;   // eight-channel blocks per wave group per chunk
  constexpr int NS = 32 * NT;        // output-channel slice of this workgroup
  constexpr int LDB = WT ? NS + 1 : NS;
  constexpr int KSLOTS = PAIR ? 1 : PCMI_MAX_KERNEL_VOLUME;
  constexpr int STAGE_FLOATS = 2 * KC * LDB;
  constexpr int RED_FLOATS = (KG > 1) ? (KG - 1) * RW * 32 * NS : 0;
  constexpr int LDS_FLOATS = STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS;

  __shared__ __attribute__((aligned(16))) float s_f[LDS_FLOATS];
  __shared__ int32_t s_idx[KSLOTS][TM];
  __shared__ int32_t s_orow[TM];
  __shared__ int32_t s_klist[KSLOTS];
  __shared__ int32_t s_nk;
  __shared__ int64_t s_tile[2];

  const int n0 = blockIdx.y * NS;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int rg = wave % RW, kg = wave / RW;
  // ---- tile descriptor -------------------------------------------------------------------
  int k_single = 0;
  if (PAIR) {
    if (t == 0) {
      int64_t tile = blockIdx.x, found_k = -1, p0 = 0;
      for (int k = 0; k < a.K; ++k) {
        const int64_t len = a.offs[k + 1] - a.offs[k];
        const int64_t nt = (len + TM - 1) / TM;
        if (tile < nt) {
          found_k = k;
          p0 = a.offs[k] + tile * TM;
          break;
        }
        tile -= nt;
      }
      s_tile[0] = found_k;
      s_tile[1] = p0;
    }
    __syncthreads();
    if (s_tile[0] < 0) return;
    k_single = (int)s_tile[0];
    const int64_t p0 = s_tile[1], pend = a.offs[k_single + 1];
    if (t < TM) {
      const int64_t p = p0 + t;
      s_idx[0][t] = p < pend ? a.pair_src[p] : -1;
      s_orow[t] = p < pend ? a.pair_dst[p] : -1;
    }
    if (t == 0) {
      s_klist[0] = 0;  // slot of s_idx; the weight slice comes from k_single
      s_nk = 1;
    }
  } else {
    // Workgroup b is observed to run on XCD b % 8 (never relied upon for correctness): give every XCD a
    // CONTIGUOUS range of tiles, so that the rows its tiles gather (scan-order neighbours) share one L2.
    int64_t tile = blockIdx.x;
    if (a.xcd_tiles > 0) {
      tile = (int64_t)(blockIdx.x & 7) * a.xcd_tiles + (blockIdx.x >> 3);
      if (tile * TM >= a.n_rows) return;
    }
    const int64_t row0 = tile * TM;
    const int kbeg = (int)((int64_t)a.K * blockIdx.z / a.ksplit);
    const int kend = (int)((int64_t)a.K * (blockIdx.z + 1) / a.ksplit);
    if (t < TM) s_orow[t] = (row0 + t < a.n_rows) ? (a.perm ? a.perm[row0 + t] : (int32_t)(row0 + t)) : -1;
    for (int p = t; p < (kend - kbeg) * TM; p += 256) {
      const int kk = p / TM, rr = p - kk * TM;
      const int64_t row = row0 + rr;
      int32_t v = -1;
      if (row < a.n_rows) v = a.nbr ? a.nbr[(int64_t)(kbeg + kk) * a.n_rows + row] : (int32_t)row;
      s_idx[kk][rr] = v;
    }
    __syncthreads();
    // compact list of offsets that have at least one neighbour in this tile
    if (t < 64) {
      int nk = 0;
      for (int kk = 0; kk < kend - kbeg; ++kk) {
        bool any = false;
        for (int rr = t; rr < TM; rr += 64) any |= (s_idx[kk][rr] >= 0);
        if (__any(any)) {
          if (t == 0) s_klist[nk] = kk;
          ++nk;
        }
      }
      if (t == 0) s_nk = nk;
    }
  }
  __syncthreads();
  const int nk = s_nk;
  const int nch = a.C / KC;
  const int nsteps = nk * nch;
  const int kbeg_blk = PAIR ? 0 : (int)((int64_t)a.K * blockIdx.z / a.ksplit);

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  // ---- staging helpers -------------------------------------------------------------------
  // (KC * NS / 4 float4 per chunk -> NT * KC / 32 per thread; kept in registers: the loops
  //  below are fully unrolled over a by-reference array, a lambda capture of it went to scratch)
  v4f breg[NT * KC / 32];
  auto load_b = [&](int step) {
    const int kslot = s_klist[step / nch];
    const int c0 = (step % nch) * KC;
    const int wk = PAIR ? a.wsel[k_single] : a.wsel[kbeg_blk + kslot];
    load_b_regs<NT, WT, KC>(breg, a.w + (int64_t)wk * a.w_kstride, a.w_sc, a.w_sn, c0, n0, t);
  };
  auto store_b = [&](int buf) { store_b_regs<NT, WT, LDB, KC>(breg, s_f + buf * (KC * LDB), t); };
  // A operands (gathers: the long-latency loads) and B chunks (weights: L2 hits shared by every workgroup, through
  // LDS) are both fetched one 32-channel step ahead.  (A two-steps-ahead register ring was measured: no gain once
  // the accumulators stayed in VGPRs, and one wave per SIMD less.)
  float4 a0[BPG], a1[BPG];
  bool v0 = false, v1 = false;
  auto load_a = [&](int step, float4* dst) -> bool {
    const int kslot = s_klist[step / nch];
    const int c0 = (step % nch) * KC;
    const int32_t idx = s_idx[kslot][rg * 32 + r];
    if (idx >= 0) {
      const float* xp = a.x + (int64_t)idx * a.x_ld + c0 + 4 * h;
#pragma unroll
      for (int b = 0; b < BPG; ++b)
        dst[b] = *reinterpret_cast<const float4*>(xp + 8 * (kg * BPG + b));
    } else {
#pragma unroll
      for (int b = 0; b < BPG; ++b) dst[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return __any(idx >= 0);
  };

  // ---- main loop --------------------------------------------------------------------------
  if (nsteps > 0) {
    load_b(0);
    v0 = load_a(0, a0);
    store_b(0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
      const bool more = step + 1 < nsteps;
      if (more) load_b(step + 1);
      if (more) v1 = load_a(step + 1, a1);
      if (v0) {
        // B fragments are read from LDS PF contraction steps AHEAD of the MFMAs that use them (register ring).
        // Written inline, hipcc emitted ds_read2 -> s_waitcnt lgkmcnt(0) -> 2 MFMAs, i.e. the full LDS latency in
        // front of every 128 cycles of matrix work: SQ_VALU_MFMA_BUSY_CYCLES showed the pipe 52 % busy.  The
        // sched_barriers pin the order (the scheduler otherwise sinks every read next to its use again).
        const float* sb = s_f + (step & 1) * (KC * LDB) + r + (8 * (kg * BPG) + 4 * h) * LDB;
        constexpr int QN = 4 * BPG;                                // contraction steps of this wave per chunk
        constexpr int PF0 = NT >= 3 ? 3 : (NT == 2 ? 4 : 6);       // >= ~380 cycles (6 MFMAs) between read and use
        constexpr int PF = PF0 < QN ? PF0 : QN;
        float bf[PF][NT];
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bf[q][nt] = sb[(8 * (q >> 2) + (q & 3)) * LDB + nt * 32];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const float4 ab = a0[q >> 2];
          const float av = (q & 3) == 0 ? ab.x : (q & 3) == 1 ? ab.y : (q & 3) == 2 ? ab.z : ab.w;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[q % PF][nt], acc[nt], 0, 0, 0);
          if (q + PF < QN) {
            const int qq = q + PF;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[q % PF][nt] = sb[(8 * (qq >> 2) + (qq & 3)) * LDB + nt * 32];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (more) store_b((step + 1) & 1);
      __syncthreads();
#pragma unroll
      for (int b = 0; b < BPG; ++b) a0[b] = a1[b];
      v0 = v1;
    }
  }

  // ---- reduce the contraction split across wave groups ----------------------------------------
  if (KG > 1) {
    // (the main loop ends with a barrier, so the staging area is free)
    if (kg > 0) {
      float* red = s_f + ((kg - 1) * RW + rg) * 32 * NS;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int i = (j & 3) + 8 * (j >> 2) + 4 * h;
          red[i * NS + nt * 32 + r] = acc[nt][j];
        }
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int g = 1; g < KG; ++g) {
        const float* red = s_f + ((g - 1) * RW + rg) * 32 * NS;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int i = (j & 3) + 8 * (j >> 2) + 4 * h;
            acc[nt][j] += red[i * NS + nt * 32 + r];
          }
      }
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------
  if (kg == 0) {
    float* outp = a.out + (PAIR ? 0 : (int64_t)blockIdx.z * a.split_stride);
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = a.bias ? a.bias[n0 + nt * 32 + r] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int i = (j & 3) + 8 * (j >> 2) + 4 * h;
      const int32_t orow = s_orow[rg * 32 + i];
      if (orow >= 0) {
        float* op = outp + (int64_t)orow * a.out_ld + n0 + r;
        if (a.accumulate) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) op[nt * 32] += acc[nt][j] + bv[nt];
        } else {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) op[nt * 32] = acc[nt][j] + bv[nt];
        }
      }
    }
  }
}

// ---- 16-row variant for the 128-row tiles ------------------------------------------------------------------------
// SK ("stream-K" over the offsets): a tile's cost is proportional to the number of offsets that occur in it (9..27
// after the mask sort), and 683 tiles on 768 resident workgroup slots is ONE round whose length the unluckiest CU
// sets -- the waves were alive for 68 % of the kernel.  In SK mode (these kernels and spconv16x_kernel) the launch is a fixed set of resident workgroups;
// the (tile, occupied offset) units of the whole level are numbered tile-major (map.tile_pref) and workgroup g
// takes units [g*per, (g+1)*per): whole tiles are written as usual, the at most two tiles a workgroup shares
// with its neighbours go to partial slots [g][0 = its first piece | 1 = its last piece] and
// sk_fixup_kernel adds the pieces of a split tile in workgroup order (deterministic).
// Same formulation, operands and LDS staging as spconv_mfma_kernel<NT, 4, WT, false>, but on
// v_mfma_f32_16x16x4_f32 (same peak): lane l = (i = l & 15, kk = l >> 4) supplies A[i][kk] and B[kk][j = i].  A wave
// still owns 32 rows, now as TWO independent 16-row groups, and skips an offset per GROUP: with 32-row groups the
// matrix pipe issued 1.18x the algorithmic FLOPs on the mask-sorted level-1 maps (PMC), the zero rows of waves whose
// 32 rows are only partly present at an offset.  The float4 a lane gathers (channels c0 + 16 blk + 4 kk .. +3 of its
// row: the four kk lanes of a row read 64 contiguous bytes) feeds four consecutive MFMA steps; step s contracts the
// channels {c0 + 16 blk + 4 kk + s}, and the B fragment uses the same bijection.  One B fragment (ds_read_b32, 2 NT
// per step) serves both row groups.  LDB = NS + 4 keeps those reads bank-conflict free (4 LDB = 16 mod 32).
typedef float f32x4 __attribute__((ext_vector_type(4)));
// waves per SIMD the 16-row kernel is compiled for: 4 for NT <= 3 (128 VGPRs, 2-5 dwords of scratch outside the
// MFMA loop) -- measured 3-5 % faster than 3 waves on the level-1 / level-2 convs (scripts/kbench.py); build with
// -DPCMI_CONV16_W4=0 (PCMI_EXTRA_HIPCC_FLAGS) for the 3-wave form
#ifndef PCMI_CONV16_W4
#define PCMI_CONV16_W4 1
#endif
__host__ __device__ constexpr int kConv16Waves(int nt) { return (PCMI_CONV16_W4 && nt <= 3) ? 4 : 3; }

// ---- the kernel (software-pipelined; round 2's unpipelined form -- same operands, bit-identical results -- is gone) ----
// Per-wave cycle accounting of the unpipelined form (profiles/r02_stall_attribution.txt): a wave spent
// 31 % of its life ISSUING the next step's loads (two dependent LDS look-ups, a kernarg look-up, 64-bit address
// arithmetic, exec-masked gathers), 33 % in its MFMA phase and 23 % at the per-step barrier; no phase of a wave
// overlapped another phase of the same wave, so the matrix pipe depended on the other three waves of the SIMD being
// in their MFMA phase (busy 83 % while all four were alive, 67 % of the kernel: the CU's workgroups finish staggered
// and the last ones run alone).  Here the next step's loads are issued from INSIDE the MFMA stream:
//   * the neighbour table is staged as 32-bit BYTE offsets (absent = 0x80000000) and the gathers / weight loads are raw
//     buffer loads: an absent row is an out-of-range offset that returns zeros -- no exec masking, no 64-bit address
//     arithmetic, the chunk offset rides in the scalar offset operand;
//   * one LDS look-up per offset ({table row, weight slice byte offset}), reused by the C / 32 chunk steps of the offset;
//   * the look-up, the weight loads and the gathers sit after the MFMAs of contraction steps 0, 1 and 2 of the running
//     chunk, each consuming what the previous stage requested, so none of them waits.
// Needs the gathered operand to be < 2 GiB (the launcher falls back to spconv_mfma_kernel otherwise).
template <int NT, bool WT, bool SK>
__global__ __launch_bounds__(256, kConv16Waves(NT)) void spconv16p_kernel(ConvArgs a) {
  constexpr int TM = 128, NS = 32 * NT, CTN = 2 * NT;
  constexpr int LDB = NS + 4;
  constexpr int KSLOTS = PCMI_MAX_KERNEL_VOLUME;
  constexpr uint32_t kAbsent = 0x80000000u;
  constexpr int kRsrcFlags = 0x00020000;  // raw buffer, 32-bit data format
  __shared__ __attribute__((aligned(16))) float s_f[2 * kKC * LDB];
  __shared__ uint32_t s_off[KSLOTS][TM];
  __shared__ int32_t s_orow[TM];
  __shared__ int2 s_kmeta[KSLOTS];  // per occupied offset: {row of s_off, byte offset of its weight slice}
  __shared__ int32_t s_kabs[KSLOTS];
  __shared__ int32_t s_nk;
  __shared__ int64_t s_tile[2];

  const int n0 = blockIdx.y * NS;
  int sk_g = 0, sk_tile = 0, sk_u = 0, sk_u1 = 0;
  bool sk_first = true;
  if constexpr (SK) {
    const int G = (int)gridDim.x;
    sk_g = (int)(blockIdx.x & 7) * (G / 8) + (int)(blockIdx.x >> 3);
    // shares are counted in chunk STEPS (units x C / 32): 15 001 units over 1024 workgroups is 15 units (45 steps) for
    // most against a mean of 43.95 steps
    const int U = a.sk_pref[a.sk_tiles] * (a.C / kKC);
    const int per = (U + G - 1) / G;
    sk_u = sk_g * per;
    sk_u1 = min(U, sk_u + per);
    if (sk_u >= sk_u1) return;
    if (threadIdx.x < 64) {  // last tile whose first unit is <= sk_u: 64-ary search (two dependent loads for <= 4096 tiles)
      int lo = 0, n = a.sk_tiles;  // answer in [lo, lo + n)
      while (n > 1) {
        const int stride = (n + 63) / 64;
        const int probe = lo + (int)threadIdx.x * stride;
        const bool le = probe < lo + n && a.sk_pref[probe] * (a.C / kKC) <= sk_u;
        const int cnt = __popcll(__ballot(le));  // probes are ordered: the first cnt are <=
        const int nlo = lo + max(cnt - 1, 0) * stride;
        n = min(stride, lo + n - nlo);
        lo = nlo;
      }
      if (threadIdx.x == 0) s_tile[0] = lo;
    }
    __syncthreads();
    sk_tile = __builtin_amdgcn_readfirstlane((int)s_tile[0]);
  }
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7FFFFFFF, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0x7FFFFFFF, kRsrcFlags);
  const uint32_t ld_bytes = (uint32_t)(a.x_ld * 4);
  const uint32_t wk_bytes = (uint32_t)(a.w_kstride * 4);
  const int sk_total = SK ? sk_u1 - sk_u : 0;  // chunk steps of this workgroup
  int sk_done = 0, prio_qtr = -1;
  for (;;) {  // one pass per tile piece (exactly one when !SK)
  bool sk_whole = true;
  int sk_next_u = 0, sk_c0 = 0, sk_steps = 0;  // SK: first chunk of the piece's first offset; steps of the piece
  int t = threadIdx.x;
  if constexpr (SK) asm volatile("" : "+v"(t));  // see spconv_mfma_kernel
  const int lane = t & 63, wave = t >> 6;
  const int i = lane & 15, kk = lane >> 4;
  if constexpr (SK) {
    const int64_t row0 = (int64_t)sk_tile * TM;
    const uint32_t tmask = a.sk_mask[sk_tile];
    const int nch_ = a.C / kKC;
    const int pref = a.sk_pref[sk_tile] * nch_, ns_t = __popc(tmask) * nch_;  // in steps
    const int jb = sk_u - pref, je = min(sk_u1 - pref, ns_t);                 // this piece: steps [jb, je) of the tile
    sk_whole = (jb == 0 && je == ns_t);
    sk_next_u = pref + je;
    const int o0 = jb / nch_, o1 = (je - 1) / nch_;  // occupied offsets (by rank) the piece touches
    sk_c0 = jb - o0 * nch_;
    sk_steps = je - jb;
    if (t < a.K && ((tmask >> t) & 1u)) {  // lane k: the rank of offset k among the occupied ones
      const int j = __popc(tmask & ((1u << t) - 1u));
      if (j >= o0 && j <= o1) {
        s_kabs[j - o0] = t;
        s_kmeta[j - o0] = make_int2(j - o0, (int)((uint32_t)a.wsel[t] * wk_bytes));
      }
    }
    if (t == 0) s_nk = sk_steps > 0 ? o1 - o0 + 1 : 0;
    if (t < TM) s_orow[t] = (row0 + t < a.n_rows) ? (a.perm ? a.perm[row0 + t] : (int32_t)(row0 + t)) : -1;
    __syncthreads();
    const int cnt = sk_steps > 0 ? o1 - o0 + 1 : 0;
    for (int p = t; p < cnt * TM; p += 256) {
      const int q = p / TM, rr = p - q * TM;
      const int64_t row = row0 + rr;
      const int32_t v = row < a.n_rows ? a.nbr[(int64_t)s_kabs[q] * a.n_rows + row] : -1;
      s_off[q][rr] = v >= 0 ? (uint32_t)v * ld_bytes : kAbsent;
    }
  } else {
    int64_t tile = blockIdx.x;
    if (a.xcd_tiles > 0) {
      tile = (int64_t)(blockIdx.x & 7) * a.xcd_tiles + (blockIdx.x >> 3);
      if (tile * TM >= a.n_rows) return;
    }
    const int64_t row0 = tile * TM;
    const int kbeg = (int)((int64_t)a.K * blockIdx.z / a.ksplit);
    const int kend = (int)((int64_t)a.K * (blockIdx.z + 1) / a.ksplit);
    if (t < TM) s_orow[t] = (row0 + t < a.n_rows) ? (a.perm ? a.perm[row0 + t] : (int32_t)(row0 + t)) : -1;
    for (int p = t; p < (kend - kbeg) * TM; p += 256) {
      const int q = p / TM, rr = p - q * TM;
      const int64_t row = row0 + rr;
      int32_t v = -1;
      if (row < a.n_rows) v = a.nbr ? a.nbr[(int64_t)(kbeg + q) * a.n_rows + row] : (int32_t)row;
      s_off[q][rr] = v >= 0 ? (uint32_t)v * ld_bytes : kAbsent;
    }
    __syncthreads();
    if (t < 64) {  // offsets with at least one neighbour in this tile
      int nk = 0;
      for (int q = 0; q < kend - kbeg; ++q) {
        bool any = false;
        for (int rr = t; rr < TM; rr += 64) any |= (s_off[q][rr] != kAbsent);
        if (__any(any)) {
          if (t == 0) s_kmeta[nk] = make_int2(q, (int)((uint32_t)a.wsel[kbeg + q] * wk_bytes));
          ++nk;
        }
      }
      if (t == 0) s_nk = nk;
    }
  }
  __syncthreads();
  const int nk = s_nk;
  const int nch = a.C / kKC;
  const int nsteps = SK ? sk_steps : nk * nch;

  f32x4 acc[2][CTN];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int ct = 0; ct < CTN; ++ct) acc[g][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this thread's float4s of a weight chunk: byte offsets relative to (slice, chunk)
  uint32_t bvo[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int e = t + q * 256;
    if (!WT) {
      const int c = e / (NS / 4), n4 = e % (NS / 4);
      bvo[q] = (uint32_t)(((int64_t)c * a.w_sc + n0 + n4 * 4) * 4);
    } else {
      const int n = e / (kKC / 4), c4 = e % (kKC / 4);
      bvo[q] = (uint32_t)(((int64_t)(n0 + n) * a.w_sn + c4 * 4) * 4);
    }
  }
  const uint32_t b_chunk_bytes = (uint32_t)((WT ? (int64_t)kKC : (int64_t)kKC * a.w_sc) * 4);
  v4f breg[NT];
  auto store_b = [&](int buf) { store_b_regs<NT, WT, LDB>(breg, s_f + buf * (kKC * LDB), t); };
  // load side: the (offset, chunk) whose operands are being requested -- one step ahead of the MFMAs
  v4f a0[2][2], a1[2][2];
  int va0 = 0, va1 = 0;
  // The three stages run on EVERY step, unconditionally (the s_waitcnt counts the compiler derives are then the same
  // on every path: a conditional load made it wait for the loads it had just issued); after the last step they
  // request out-of-range offsets, which cost no memory traffic.
  int lj = 0, lc = sk_c0 - 1;       // uniform
  int2 l_meta = make_int2(0, 0);    // s_kmeta[lj]
  uint32_t l_voff[2] = {kAbsent, kAbsent};  // byte offset of this lane's row (+ 16 kk) for the two groups
  const uint32_t off_lane = (uint32_t)(wave * 32 + i) * 4;
  auto stage_meta = [&]() {  // advance to the next (offset, chunk); its look-up
    ++lc;
    if (lc == nch) {
      lc = 0;
      ++lj;
    }
    lc = __builtin_amdgcn_readfirstlane(lc);  // (uniform by construction; keeps them in SGPRs)
    lj = __builtin_amdgcn_readfirstlane(lj);
    l_meta = s_kmeta[min(lj, KSLOTS - 1)];
  };
  auto stage_b = [&](bool live) {  // weights of (lj, lc); this lane's rows of the offset table
    const int krow = __builtin_amdgcn_readfirstlane(l_meta.x) & 31;
    const uint32_t wofs = (uint32_t)__builtin_amdgcn_readfirstlane(l_meta.y);
    const char* row = reinterpret_cast<const char*>(&s_off[0][0]) + min(krow, KSLOTS - 1) * (TM * 4) + off_lane;
    l_voff[0] = *reinterpret_cast<const uint32_t*>(row);
    l_voff[1] = *reinterpret_cast<const uint32_t*>(row + 64);
    const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wofs + (uint32_t)lc * b_chunk_bytes));
#pragma unroll
    for (int q = 0; q < NT; ++q)
      breg[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wr, live ? bvo[q] : kAbsent, live ? soff : 0u, 0));
  };
  auto stage_a = [&](v4f (&dst)[2][2], bool live) -> int {  // gathers of (lj, lc)
    const uint32_t v0 = live ? l_voff[0] : kAbsent, v1 = live ? l_voff[1] : kAbsent;
    const int va = (__any(v0 != kAbsent) ? 1 : 0) | (__any(v1 != kAbsent) ? 2 : 0);
    const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(lc * (kKC * 4));
    const uint32_t o0 = v0 + 16 * kk, o1 = v1 + 16 * kk;  // absent stays out of range
    dst[0][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, o0, soff, 0));
    dst[1][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, o1, soff, 0));
    dst[0][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, o0 + 64, soff, 0));
    dst[1][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, o1 + 64, soff, 0));
    return va;
  };

  if (nsteps > 0) {
    stage_meta();
    stage_b(true);
    va0 = stage_a(a0, true);
    store_b(0);
    __syncthreads();
    // one chunk step on operand set `cur`, requesting the next one into `nxt` (two sets, swapped every step: no copies,
    // and the wait for a gather sits at its first use)
    auto do_step = [&](int step, v4f (&cur)[2][2], int va_cur, v4f (&nxt)[2][2], int& va_nxt) {
      const bool more = step + 1 < nsteps;
      if constexpr (SK) {
        // issue priority falls with the workgroup's progress through its (equal) share of units: the four workgroups
        // of a CU otherwise finish staggered (oldest first), and the last ones run with a half-empty matrix pipe
        const int qtr = __builtin_amdgcn_readfirstlane(((sk_done + step) * 4) / max(sk_total, 1));
        if (qtr != prio_qtr) {
          prio_qtr = qtr;
          if (qtr <= 0) __builtin_amdgcn_s_setprio(3);
          else if (qtr == 1) __builtin_amdgcn_s_setprio(2);
          else if (qtr == 2) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
      }
      const float* sb = s_f + (step & 1) * (kKC * LDB) + i + (4 * kk) * LDB;
      constexpr int QN = 8, PF = 2;
      float bf[PF][CTN];
      if (va_cur) {
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
          for (int ct = 0; ct < CTN; ++ct) bf[q][ct] = sb[(16 * (q >> 2) + (q & 3)) * LDB + ct * 16];
      }
      __builtin_amdgcn_sched_barrier(0);
      const bool g0 = (va_cur & 1) != 0, g1 = (va_cur & 2) != 0;
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        const float av0 = cur[0][q >> 2][q & 3], av1 = cur[1][q >> 2][q & 3];
        if (g0) {
#pragma unroll
          for (int ct = 0; ct < CTN; ++ct)
            acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bf[q % PF][ct], acc[0][ct], 0, 0, 0);
        }
        if (g1) {
#pragma unroll
          for (int ct = 0; ct < CTN; ++ct)
            acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bf[q % PF][ct], acc[1][ct], 0, 0, 0);
        }
        if (va_cur && q + PF < QN) {
          const int qq = q + PF;
#pragma unroll
          for (int ct = 0; ct < CTN; ++ct) bf[q % PF][ct] = sb[(16 * (qq >> 2) + (qq & 3)) * LDB + ct * 16];
        }
        // the next step's operands, requested in the shadow of this step's MFMAs
        if (q == 0) stage_meta();
        if (q == 1) stage_b(more);
        if (q == 2) va_nxt = stage_a(nxt, more);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) store_b((step + 1) & 1);
      __syncthreads();
    };
    for (int step = 0; step < nsteps; step += 2) {
      do_step(step, a0, va0, a1, va1);
      if (step + 1 < nsteps) do_step(step + 1, a1, va1, a0, va0);
    }
  }

  // ---- epilogue: D[row = 4 kk + r][col = i] of every 16x16 tile --------------------------------------------------
  if (SK && !sk_whole) {
    float* pp = a.sk_part + ((int64_t)(2 * sk_g + (sk_first ? 0 : 1)) * TM + wave * 32) * a.N + n0 + i;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = g * 16 + 4 * kk + r;
#pragma unroll
        for (int ct = 0; ct < CTN; ++ct) pp[(int64_t)rl * a.N + ct * 16] = acc[g][ct][r];
      }
  } else {
    float* outp = a.out + (int64_t)blockIdx.z * a.split_stride;
    float bv[CTN];
#pragma unroll
    for (int ct = 0; ct < CTN; ++ct) bv[ct] = a.bias ? a.bias[n0 + ct * 16 + i] : 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int32_t orow = s_orow[wave * 32 + g * 16 + 4 * kk + r];
        if (orow >= 0) {
          float* op = outp + (int64_t)orow * a.out_ld + n0 + i;
          if (a.accumulate) {
#pragma unroll
            for (int ct = 0; ct < CTN; ++ct) op[ct * 16] += acc[g][ct][r] + bv[ct];
          } else {
#pragma unroll
            for (int ct = 0; ct < CTN; ++ct) op[ct * 16] = acc[g][ct][r] + bv[ct];
          }
        }
      }
  }
  if constexpr (!SK) {
    break;
  } else {
    sk_u = sk_next_u;
    ++sk_tile;
    sk_first = false;
    sk_done += nsteps;
    if (sk_u >= sk_u1) break;
    __syncthreads();  // s_off / s_orow / the staging area are rewritten by the next piece
  }
  }  // for (;;)
}

// Sums the pieces of the tiles that the SK launch split between workgroups (see spconv_mfma_kernel) into the
// output rows, in workgroup order.  One workgroup per tile; tiles written whole by one workgroup are skipped.
__global__ __launch_bounds__(256) void sk_fixup_kernel(const float* __restrict__ part, const int32_t* __restrict__ pref,
                                                       int n_tiles, int G, const int32_t* __restrict__ perm, int64_t n_rows,
                                                       int N, const float* __restrict__ bias, float* __restrict__ out,
                                                       int64_t out_ld, int accumulate, int sub) {
  constexpr int TM = 128;
  const int tile = blockIdx.x;
  // sub: shares counted in sub-steps of a unit (spconv16p_kernel: C / 32 chunk steps), 1 = whole units
  const int U = pref[n_tiles] * sub, per = (U + G - 1) / G;
  const int u0 = pref[tile] * sub, u1 = pref[tile + 1] * sub;
  if (u1 <= u0) return;
  const int g_first = u0 / per, g_last = (u1 - 1) / per;
  if (g_first == g_last) return;  // one workgroup had the whole tile and wrote it itself
  const int n4 = N / 4;
  const int64_t row0 = (int64_t)tile * TM;
  for (int e = threadIdx.x; e < TM * n4; e += 256) {
    const int rr = e / n4, c = (e - rr * n4) * 4;
    const int64_t row = row0 + rr;
    if (row >= n_rows) continue;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = g_first; g <= g_last; ++g) {
      const int slot = (u0 <= g * per) ? 0 : 1;  // the tile holds the first unit of g: g's first piece
      const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)(2 * g + slot) * TM + rr) * N + c);
      sum.x += v.x;
      sum.y += v.y;
      sum.z += v.z;
      sum.w += v.w;
    }
    if (bias) {
      sum.x += bias[c];
      sum.y += bias[c + 1];
      sum.z += bias[c + 2];
      sum.w += bias[c + 3];
    }
    const int64_t orow = perm ? perm[row] : row;
    float4* dst = reinterpret_cast<float4*>(out + orow * out_ld + c);
    if (accumulate) {
      const float4 o = *dst;
      sum.x += o.x;
      sum.y += o.y;
      sum.z += o.z;
      sum.w += o.w;
    }
    *dst = sum;
  }
}

// out[row, n] = sum_s partial[s][row, n] (+ bias)
// (Round 6 measured this kernel, sk_fixup_kernel above and wgrad_slab_sum_kernel with every load of a batch really in flight --
//  hipcc re-rolls the batch below into one load + s_waitcnt vmcnt(0) per partial tensor -- through clamped indices and template
//  flags: in the STEP they were not faster (split_reduce 0.76 -> 0.78, sk_fixup 0.45 -> 0.51, slab_sum 0.46 -> 0.55 ms per step of
//  kernel time, the step itself unchanged: profiles/r06h_*): beside the other streams these passes wait for bandwidth, not
//  for their own latency.  The forms that did get faster are kept: wgrad_reduce_kernel, colsum_partial_kernel and the
//  unit-balanced launch's table build in spconv16x_kernel.)
__global__ void split_reduce_kernel(const float* __restrict__ part, int64_t split_stride, int ksplit,
                                    int64_t n_rows, int N, const float* __restrict__ bias,
                                    float* __restrict__ out, int64_t out_ld, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
  const int n4 = N / 4;
  if (idx >= n_rows * n4) return;
  const int64_t row = idx / n4;
  const int c = (int)(idx % n4) * 4;
  // partials in batches of 8: all loads of a batch are in flight together (one dependent load per partial was a
  // chain of up to 27 L2 latencies); the additions stay in partial order -> same sums as before, deterministic
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i0 = 0; i0 < ksplit; i0 += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      v[u] = i0 + u < ksplit ? *reinterpret_cast<const float4*>(part + (int64_t)(i0 + u) * split_stride + row * N + c)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s.x += v[u].x;
      s.y += v[u].y;
      s.z += v[u].z;
      s.w += v[u].w;
    }
  }
  if (bias) {
    s.x += bias[c];
    s.y += bias[c + 1];
    s.z += bias[c + 2];
    s.w += bias[c + 3];
  }
  float4* dst = reinterpret_cast<float4*>(out + row * out_ld + c);
  if (accumulate) {
    const float4 o = *dst;
    s.x += o.x;
    s.y += o.y;
    s.z += o.z;
    s.w += o.w;
  }
  *dst = s;
}

// ---- tiny-channel stem (cin < 8: conv0p1s1 has cin = 3): plain VALU, HBM-bound -----------------
template <int CIN>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, int64_t x_ld,
                                                       const float* __restrict__ w, int cout,
                                                       const int32_t* __restrict__ nbr, int K,
                                                       int64_t n_rows, const float* __restrict__ bias,
                                                       float* __restrict__ out, int64_t out_ld) {
  extern __shared__ float s_w[];  // [K][CIN][cout]
  for (int i = threadIdx.x; i < K * CIN * cout; i += 256) s_w[i] = w[i];
  __syncthreads();
  const int rows_per_block = 256 / 32;
  const int n = threadIdx.x & 31;
  const int64_t row = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x >> 5);
  if (row >= n_rows) return;
  for (int nn = n; nn < cout; nn += 32) {
    float acc = bias ? bias[nn] : 0.f;
    for (int k = 0; k < K; ++k) {
      const int32_t idx = nbr ? nbr[(int64_t)k * n_rows + row] : (int32_t)row;
      if (idx < 0) continue;
      const float* xp = x + (int64_t)idx * x_ld;
#pragma unroll
      for (int c = 0; c < CIN; ++c) acc = fmaf(xp[c], s_w[(k * CIN + c) * cout + nn], acc);
    }
    out[row * out_ld + nn] = acc;
  }
}

// 3-channel stem, 32 output channels: ONE LANE PER OUTPUT ROW.  The neighbour table is read coalesced (consecutive
// lanes = consecutive rows of nbr[k][*]), the 12-byte input rows are gathered once per (row, offset), the 27 x 3 x 32
// weights are wave-uniform (scalar loads: the address does not depend on the lane) and every lane keeps its 32
// accumulators in registers and writes one contiguous 128-byte output row.  HBM-bound: 27 x 4 B of table + <= 27 x 12 B
// of input + 128 B of output per row.  (stem_fwd_kernel above spends a 10 KB LDS weight staging per 8 rows and reads
// every table entry / input value 32 times.)
template <int K>
__global__ __launch_bounds__(256) void stem32_fwd_kernel(const float* __restrict__ x, int64_t x_ld,
                                                         const float* __restrict__ w /* [K][3][32] */,
                                                         const int32_t* __restrict__ nbr, int64_t n_rows,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int64_t out_ld) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n_rows) return;
  // all K table entries first, then all K input rows: two memory latencies per output row instead of 2 K
  int32_t idx[K];
#pragma unroll
  for (int k = 0; k < K; ++k) idx[k] = nbr ? nbr[(int64_t)k * n_rows + row] : (int32_t)row;
  float xv[K][3];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float* xp = x + (int64_t)max(idx[k], 0) * x_ld;
    const bool ok = idx[k] >= 0;
    xv[k][0] = ok ? xp[0] : 0.f;
    xv[k][1] = ok ? xp[1] : 0.f;
    xv[k][2] = ok ? xp[2] : 0.f;
  }
  float acc[32];
#pragma unroll
  for (int n = 0; n < 32; ++n) acc[n] = bias ? bias[n] : 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (idx[k] < 0) continue;  // (an absent neighbour must not contribute: 0 * w could still be -0 / NaN-propagating)
    const float* wk = w + k * 96;
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = fmaf(xv[k][2], wk[64 + n], fmaf(xv[k][1], wk[32 + n], fmaf(xv[k][0], wk[n], acc[n])));
  }
  float4* op = reinterpret_cast<float4*>(out + row * out_ld);
#pragma unroll
  for (int q = 0; q < 8; ++q) op[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
}

template <int NT, int RW, bool WT, bool PAIR>
static int launch_one(const ConvArgs& a, dim3 grid, hipStream_t st) {
  spconv_mfma_kernel<NT, RW, WT, PAIR><<<grid, 256, 0, st>>>(a);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

constexpr int kStreamKDefaultMinTiles = 256;  // unit-balanced launch from 256 tiles (32768 rows); PCMI_SPCONV_STREAMK=0 turns it off

// PCMI_SPCONV_STREAMK: minimum number of 128-row tiles for the unit-balanced launch (0 = never).
static int64_t sk_min_tiles() {  // read per call: the parity test compares both launches in one process
  const char* e = getenv("PCMI_SPCONV_STREAMK");
  return e ? (int64_t)atoll(e) : (int64_t)kStreamKDefaultMinTiles;
}
// resident workgroups of the unit-balanced launch (3 or 4 waves per SIMD, see kConv16Waves), multiple of 8
static int sk_workgroups(int NT = 4) { return kConv16Waves(NT) * num_cu() / 8 * 8; }
static int sk_workgroups_max() { return 4 * num_cu() / 8 * 8; }
static bool sk_rows_eligible(int64_t rows, int K) {
  const int64_t mt = sk_min_tiles();
  return K > 1 && mt > 0 && rows >= 4096 /* kSortRowsMin: such maps carry tile units */ && ceil_div(rows, 128) >= mt;
}
static size_t sk_partial_bytes(int64_t rows, int N, int K) {
  return sk_rows_eligible(rows, K) ? (size_t)sk_workgroups_max() * 2 * 128 * N * sizeof(float) : 0;
}

// The 16-row kernels take the 128-row tiles of levels with at least PCMI_CONV16 rows (0 = never, 1 = always).  Default
// 512 since round 3: with the weights of every layer packed for its level's slice width ahead of the launches
// (x3_plan_nt), the split-precision kernel also wins on the coarse levels -- 240 -> 247 pairs/s at 2048, 249 at 512
// (profiles/r03e_bench_ab_*.txt); round 2's 8192 was measured with the fp32 form, which lost below it.
static int64_t conv16_min_rows() {
  const char* e = getenv("PCMI_CONV16");
  return e ? atoll(e) : 512;
}
static bool conv16_enabled(int64_t n_rows, int64_t x_bytes) {
  const int64_t min_rows = conv16_min_rows();
  // the pipelined form addresses the gathered operand with 32-bit byte offsets (absent = 2^31)
  return min_rows > 0 && n_rows >= min_rows && x_bytes <= 0x7FFFFF00ll;
}

// The split-precision form (spconv_x3.hip: fp32 operands as three bf16 terms on the bf16 matrix cores) takes the
// matrix-bound launches of the 16-row kernel (>= 64 channels on both sides); PCMI_CONV16_X3=0: the fp32-MFMA kernel.
static bool conv16_x3_on() {
  const char* e = getenv("PCMI_CONV16_X3");
  return !e || atoi(e) != 0;
}
static bool conv16_x3(int NT, int C, int N) { return conv16_x3_on() && NT >= 2 && NT <= 4 && C >= 64 && N >= 64; }
// resident workgroups per CU of spconv16x_kernel (LDS: 2 weight blocks of 6 KiB x NT + the 13.5 KiB offset table)
static int x3_workgroups(int NT) { return (NT <= 3 ? 3 : 2) * num_cu() / 8 * 8; }

template <bool WT, bool SK>
static int launch16(int NT, const ConvArgs& a, dim3 grid, hipStream_t st) {
  switch (NT) {
    case 1: spconv16p_kernel<1, WT, SK><<<grid, 256, 0, st>>>(a); break;
    case 2: spconv16p_kernel<2, WT, SK><<<grid, 256, 0, st>>>(a); break;
    case 3: spconv16p_kernel<3, WT, SK><<<grid, 256, 0, st>>>(a); break;
    case 4: spconv16p_kernel<4, WT, SK><<<grid, 256, 0, st>>>(a); break;
    default: set_error("spconv: bad NT %d", NT); return PCMI_ERR_INVALID;
  }
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

// 128-row tiles, narrow slices (NT <= 2, levels under 8192 rows): 64-channel chunks when the contraction size allows --
// twice the matrix work and loads in flight per barrier (NT = 1 keeps 4 waves per SIMD, NT = 2 keeps 3; 128-channel
// chunks cost a wave per SIMD and lost).  false = not taken.
template <bool WT>
static bool launch_deep(int NT, const ConvArgs& a, dim3 grid, hipStream_t st) {
  if (NT == 1 && a.C % 64 == 0) {
    spconv_mfma_kernel<1, 4, WT, false, 64><<<grid, 256, 0, st>>>(a);
    return true;
  }
  if (NT == 2 && a.C % 64 == 0) {
    spconv_mfma_kernel<2, 4, WT, false, 64><<<grid, 256, 0, st>>>(a);
    return true;
  }
  return false;
}

template <int RW, bool WT, bool PAIR>
static int launch_nt(int NT, const ConvArgs& a, dim3 grid, hipStream_t st) {
  switch (NT) {
    case 1: return launch_one<1, RW, WT, PAIR>(a, grid, st);
    case 2: return launch_one<2, RW, WT, PAIR>(a, grid, st);
    case 3: if constexpr (RW > 1) return launch_one<3, RW, WT, PAIR>(a, grid, st); else break;
    case 4: if constexpr (RW > 1) return launch_one<4, RW, WT, PAIR>(a, grid, st); else break;
  }
  set_error("spconv: bad NT %d", NT);
  return PCMI_ERR_INVALID;
}

template <bool WT, bool PAIR>
static int launch_rw(int RW, int NT, const ConvArgs& a, dim3 grid, hipStream_t st) {
  switch (RW) {
    case 4: return launch_nt<4, WT, PAIR>(NT, a, grid, st);
    case 2: return launch_nt<2, WT, PAIR>(NT, a, grid, st);
    case 1: return launch_nt<1, WT, PAIR>(NT, a, grid, st);
  }
  set_error("spconv: bad RW %d", RW);
  return PCMI_ERR_INVALID;
}

struct Plan {
  int RW, NT, ksplit;
};
constexpr int kMaxKSplit = 27;

// rows: output rows (or pairs in pair mode); N: output channels; K: offsets.
// Row tiles stay as large as the row count allows (a tile streams its weight slice once, so small
// tiles multiply the weight traffic: measured 10-20 TFLOP/s with 32-row tiles on the 256-channel
// levels); the parallelism the small levels lack comes from splitting the offset range over
// blockIdx.z into partial sums instead.
// wide: the contraction has >= 64 channels (with N >= 64 the launch can take the split-precision kernel, whose slices
// are at least 64 wide)
// C: the contraction size when the caller knows it (0: the round-2 rule alone)
static Plan make_plan(int64_t rows, int N, int K, bool pair, bool wide = false, int C = 0) {
  Plan p;
  const int nt_all = N / 32;
  p.NT = nt_all % 4 == 0 ? 4 : (nt_all % 3 == 0 ? 3 : (nt_all % 2 == 0 ? 2 : 1));
  p.ksplit = 1;
  p.RW = rows >= 96 ? 4 : (rows >= 48 ? 2 : 1);
  if (p.RW == 1 && p.NT > 2) p.NT = (nt_all % 2 == 0) ? 2 : 1;  // the cross-wave reduction lives in LDS
  // small levels: narrow output slices partition the weights (no extra weight traffic) and multiply the
  // number of resident workgroups; the re-gathered rows are L2-resident at these sizes
  const int64_t min16 = conv16_min_rows();
  if (rows < 2048)
    p.NT = (wide && !pair && K > 1 && N >= 64 && nt_all % 2 == 0 && conv16_x3_on() && min16 > 0 && rows >= min16 && p.RW == 4) ? 2 : 1;
  else if (rows < 8192 && p.NT > 2)
    p.NT = (nt_all % 2 == 0) ? 2 : 1;
  if (!pair && K > 1) {
    const int64_t wgs = ceil_div(rows, 32 * p.RW) * (nt_all / p.NT);
    // workgroups aimed at: 2.5 per CU (measured again under the round-3 default: 1.0 -> 240.3, 1.5 -> 247.0, 2.5 -> 249.1,
    // 4.0 -> 244.6 pairs/s, profiles/r03h_bench_ab_*.txt)
    const int64_t target = (25 * (int64_t)num_cu()) / 10;
    if (wgs < target) p.ksplit = (int)std::min<int64_t>(std::min<int64_t>(K, kMaxKSplit), ceil_div(target, wgs));
    // Round 6: the split by RESIDENCY ROUNDS, for the launches of the 16-row split-precision kernel (whose residency is
    // known: x3_workgroups).  The kernel's run time is that of its longest workgroup chain: rounds x (offsets per workgroup
    // x chunks + a fixed share), and "2.5 workgroups per CU" ignores both factors -- on the bench batch it gives the
    // stride-16 level (28 workgroups per split) ksplit 23, i.e. 4 of the 23 z-slices carry TWO offsets and every launch
    // takes 16 chunk steps where ksplit 27 takes 8; the stride-4 128-channel launches (79 tiles) ksplit 9 = 711 workgroups
    // on 512 slots, a full round and a 40 %-filled second one, where ksplit 6 is ONE round.  cost(ks), in chunk
    // steps: rounds x (ceil(K / ks) C / 32 + 3) x (a penalty under 2 workgroups per CU: nobody hides a step's latency)
    // + the partial tensors' write + read-back (8 B per element and split at ~3 TB/s against ~2 us per step).  Taken only
    // when it beats the rule above by 10 % of its own estimate, and only under 16384 rows: on the stride-2 level it picks
    // ksplit 2, which measured equal in the forward pass and 20 % SLOWER in the backward pass, beside the weight-gradient
    // stream (profiles/r06d_*: 632 workgroups of 42 steps leave nothing for the other stream to slip into).
    // PCMI_KSPLIT_RULE=0: the rule above alone (read per call).
    const char* re = getenv("PCMI_KSPLIT_RULE");
    if (C >= 64 && wide && p.RW == 4 && wgs < target && rows < 16384 && !(re && re[0] == '0') && conv16_x3(p.NT, C, N) && min16 > 0 &&
        rows >= min16) {
      const double slots = (double)x3_workgroups(p.NT), nch = (double)(C / kKC);
      const double traffic = (double)rows * N * 8.0 / 3.0e6 / 2.0;  // chunk steps per partial tensor
      auto cost = [&](int ks) {
        const double w = (double)wgs * ks;
        const double rounds = std::ceil(w / slots);
        const double thin = std::max(1.0, 2.0 * num_cu() / std::min(w, slots));
        return rounds * ((double)ceil_div((int64_t)K, (int64_t)ks) * nch + 3.0) * thin + traffic * ks;
      };
      int best = p.ksplit;
      const int kmax = (int)std::min<int64_t>(K, kMaxKSplit);
      for (int ks = 1; ks <= kmax; ++ks)
        if (cost(ks) < cost(best) - 1e-9) best = ks;
      if (cost(best) < 0.9 * cost(p.ksplit)) p.ksplit = best;
    }
  }
  return p;
}

static size_t partial_bytes(int64_t rows, int N, int K) {
  if (K <= 1 || rows <= 0 || N % 32 != 0) return 0;
  int ks = std::max(make_plan(rows, N, K, false, false).ksplit, make_plan(rows, N, K, false, true).ksplit);
  for (int C = 64; C <= 512; C += 32) ks = std::max(ks, make_plan(rows, N, K, false, true, C).ksplit);  // (any contraction size)
  return ks > 1 ? (size_t)ks * rows * N * sizeof(float) : 0;
}

// Output slice width (units of 32 channels) of the table launch over `rows` output rows if it takes the split-precision
// kernel, else 0: what the executor packs the weights for ahead of the launches (engine.hip: x3_prepack).
int x3_plan_nt(int64_t rows, int C, int N, int K) {
  if (C % 32 != 0 || N % 32 != 0 || C < 64 || N < 64 || K <= 1 || !conv16_x3_on()) return 0;
  const Plan p = make_plan(rows, N, K, false, true);
  const int64_t min16 = conv16_min_rows();
  if (p.RW != 4 || min16 <= 0 || rows < min16) return 0;
  return conv16_x3(p.NT, C, N) ? p.NT : 0;
}

// One gathered GEMM:  out[rows, N] = sum_k x[idx_k(rows)] @ B_k
static int run_gathered(const float* x, int64_t x_ld, int64_t x_rows, int C, const float* w, int cin, int cout,
                        bool w_transposed, int N, const pcmi_kmap_t* map, bool pair_mode,
                        bool swap_pairs, const int32_t* wsel, const float* bias, float* out,
                        int64_t out_ld, int64_t n_rows, int accumulate, void* ws, size_t ws_bytes,
                        hipStream_t st) {
  PCMI_REQUIRE(C % 32 == 0 && N % 32 == 0, PCMI_ERR_UNSUPPORTED,
               "spconv: channels (%d -> %d) must be multiples of 32 on the MFMA path", C, N);
  PCMI_REQUIRE(x_ld % 4 == 0 && out_ld >= N && ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0),
               PCMI_ERR_INVALID, "spconv: operands must be 16-byte aligned with ld %% 4 == 0");
  if (n_rows == 0) return PCMI_OK;
  ConvArgs a;
  a.x = x;
  a.x_ld = x_ld;
  a.C = C;
  a.w = w;
  a.w_kstride = (int64_t)cin * cout;
  if (!w_transposed) {  // B[c][n] = W[c][n], contraction over cin
    a.w_sc = cout;
    a.w_sn = 1;
  } else {  // B[c][n] = W[n][c], contraction over cout
    a.w_sc = 1;
    a.w_sn = cout;
  }
  a.N = N;
  a.K = map ? map->K : 1;
  for (int k = 0; k < PCMI_MAX_KERNEL_VOLUME; ++k) a.wsel[k] = wsel ? wsel[k] : k;
  a.n_rows = n_rows;
  a.bias = bias;
  a.nbr = nullptr;
  a.perm = nullptr;
  a.pair_src = a.pair_dst = nullptr;
  a.offs = nullptr;
  a.ksplit = 1;
  a.split_stride = 0;
  a.xcd_tiles = 0;
  a.accumulate = accumulate;
  a.out = out;
  a.out_ld = out_ld;
  a.sk_mask = nullptr;
  a.sk_pref = nullptr;
  a.sk_tiles = 0;
  a.sk_part = nullptr;
  a.wpack = nullptr;
  if (pair_mode) {
    a.pair_src = swap_pairs ? map->pair_in : map->pair_out;
    a.pair_dst = swap_pairs ? map->pair_out : map->pair_in;
    a.offs = map->offs;
    Plan p = make_plan(n_rows, N, a.K, true);
    const int TM = 32 * p.RW;
    // tiles of all offsets: exact when the map's counts are on the host, else their bound (every pair is one fine row
    // and every offset adds at most one ragged tile); a workgroup past the last tile leaves at once
    int64_t tiles = 0;
    if (map->M >= 0)
      for (int k = 0; k < a.K; ++k) tiles += ceil_div(map->offs_host[k + 1] - map->offs_host[k], TM);
    else
      tiles = ceil_div(kmap_pairs_bound(*map), TM) + a.K;
    if (tiles == 0) return PCMI_OK;
    dim3 grid((unsigned)tiles, (unsigned)(N / (32 * p.NT)), 1);
    return w_transposed ? launch_rw<true, true>(p.RW, p.NT, a, grid, st)
                        : launch_rw<false, true>(p.RW, p.NT, a, grid, st);
  }
  a.nbr = map ? map->nbr : nullptr;
  if (map && map->perm && map->nbr_perm) {  // same results, rows visited in mask-sorted order
    a.nbr = map->nbr_perm;
    a.perm = map->perm;
  }
  // 32 -> 32 channels: all weight slices resident in LDS, a wave per 16-row group (spconv32r.hip)
  if (conv32r_eligible(a, x_rows * x_ld * 4)) return conv32r_launch(w_transposed, a, st);
  Plan p = make_plan(n_rows, N, a.K, false, C >= 64, C);
  // (32-channel convs are HBM/latency-bound: the partial tiles cost them more than the balance gains -- measured)
  // (the unit-balanced launch exists for the 16-row kernels: an operand of >= 2 GiB, which they cannot address, takes the
  //  whole-tile launch of spconv_mfma_kernel below)
  // (Round 6 measured this launch also on the levels that split their offsets over blockIdx.z -- ksplit > 1: the stride-2 level
  //  of the bench batch, 316 tiles, neutral; the stride-4 level as well, 78 tiles cut into 3-10 pieces each, 0.5 ms per step
  //  SLOWER: profiles/r06a_wgrad_stream_cost_and_sk_mid_ab.txt.  The offset split + split_reduce_kernel stays there.)
  if (map && map->tile_pref && map->perm && p.RW == 4 && p.ksplit == 1 && sk_rows_eligible(n_rows, a.K) &&
      map->n_tiles == ceil_div(n_rows, 128) && C >= 64 && N >= 64 && conv16_enabled(n_rows, x_rows * x_ld * 4)) {
    const bool x3 = conv16_x3(p.NT, C, N);
    const int G = x3 ? x3_workgroups(p.NT) : sk_workgroups(p.NT);
    const size_t part = sk_partial_bytes(n_rows, N, a.K);
    const size_t need = part + (x3 ? x3_pack_bytes(a.K, C, N) : 0);
    PCMI_REQUIRE(ws && ws_bytes >= need, PCMI_ERR_WORKSPACE, "spconv: workspace %zu < %zu bytes", ws_bytes, need);
    a.sk_mask = map->tile_mask;
    a.sk_pref = map->tile_pref;
    a.sk_tiles = (int)map->n_tiles;
    a.sk_part = (float*)ws;
    dim3 grid((unsigned)G, (unsigned)(N / (32 * p.NT)), 1);
    int rc;
    if (x3) {
      a.wpack = x3_find_prepacked(w, w_transposed, p.NT);  // the executor's once-per-pass pack, if there is one
      rc = PCMI_OK;
      if (!a.wpack) {
        a.wpack = (char*)ws + part;
        rc = x3_pack_weights(a, p.NT, const_cast<void*>(a.wpack), st);
      }
      if (rc == PCMI_OK) rc = x3_launch(p.NT, true, a, grid, st);
    } else
      rc = w_transposed ? launch16<true, true>(p.NT, a, grid, st) : launch16<false, true>(p.NT, a, grid, st);
    if (rc) return rc;
    const int sub = C / kKC;  // the shares of both kernels are counted in chunk steps
    sk_fixup_kernel<<<dim3((unsigned)a.sk_tiles), 256, 0, st>>>(a.sk_part, a.sk_pref, a.sk_tiles, G, a.perm, n_rows, N, bias,
                                                              out, out_ld, accumulate, sub);
    PCMI_LAUNCH_CHECK();
    return PCMI_OK;
  }
  const bool x3 = p.RW == 4 && map && conv16_enabled(n_rows, x_rows * x_ld * 4) && conv16_x3(p.NT, C, N);
  const size_t split_bytes = p.ksplit > 1 ? align_up((size_t)p.ksplit * n_rows * N * sizeof(float), 256) : 0;
  if (x3) {
    PCMI_REQUIRE(ws && ws_bytes >= split_bytes + x3_pack_bytes(a.K, C, N), PCMI_ERR_WORKSPACE,
                 "spconv: workspace %zu < %zu bytes", ws_bytes, split_bytes + x3_pack_bytes(a.K, C, N));
    a.wpack = x3_find_prepacked(w, w_transposed, p.NT);
    if (!a.wpack) {
      a.wpack = (char*)ws + split_bytes;
      const int rc_pack = x3_pack_weights(a, p.NT, const_cast<void*>(a.wpack), st);
      if (rc_pack) return rc_pack;
    }
  }
  if (p.ksplit > 1) {
    const size_t need = (size_t)p.ksplit * n_rows * N * sizeof(float);
    PCMI_REQUIRE(ws && ws_bytes >= need, PCMI_ERR_WORKSPACE, "spconv: workspace %zu < %zu bytes", ws_bytes, need);
    a.ksplit = p.ksplit;
    a.split_stride = n_rows * N;
    a.out = (float*)ws;
    a.out_ld = N;
    a.bias = nullptr;
    a.accumulate = 0;
  }
  int64_t tiles = ceil_div(n_rows, 32 * p.RW);
  if (tiles >= 64) {  // XCD-contiguous tile order (for RW == 4 it coincides with the chunks of the mask sort)
    a.xcd_tiles = (int)ceil_div(tiles, 8);
    tiles = (int64_t)a.xcd_tiles * 8;
  }
  dim3 grid((unsigned)tiles, (unsigned)(N / (32 * p.NT)), (unsigned)p.ksplit);
  int rc;
  if (x3) {
    rc = x3_launch(p.NT, false, a, grid, st);
  } else if (p.RW == 4 && conv16_enabled(n_rows, x_rows * x_ld * 4)) {
    rc = w_transposed ? launch16<true, false>(p.NT, a, grid, st) : launch16<false, false>(p.NT, a, grid, st);
  } else if (p.RW == 4 && (w_transposed ? launch_deep<true>(p.NT, a, grid, st) : launch_deep<false>(p.NT, a, grid, st))) {
    rc = PCMI_OK;
    PCMI_LAUNCH_CHECK();
  } else
    rc = w_transposed ? launch_rw<true, false>(p.RW, p.NT, a, grid, st) : launch_rw<false, false>(p.RW, p.NT, a, grid, st);
  if (rc) return rc;
  if (p.ksplit > 1) {
    const int64_t n4 = n_rows * (N / 4);
    split_reduce_kernel<<<dim3((unsigned)ceil_div(n4, 256)), 256, 0, st>>>((const float*)ws, a.split_stride, p.ksplit,
                                                                         n_rows, N, bias, out, out_ld, accumulate);
    PCMI_LAUNCH_CHECK();
  }
  return PCMI_OK;
}

size_t spconv_fwd_bwd_workspace(int64_t n_in, int64_t n_out, int cin, int cout, int K) {
  const size_t ks = std::max(partial_bytes(n_out, cout, K), partial_bytes(n_in, cin, K));
  const size_t sk = n_in == n_out ? std::max(sk_partial_bytes(n_out, cout, K), sk_partial_bytes(n_in, cin, K)) : 0;
  // the packed weights of the split-precision form sit behind the partial tiles (whether or not it is switched on:
  // <= 10.6 MB for the widest layer)
  const size_t pack = (K > 1 && cin % 32 == 0 && cout % 32 == 0) ? x3_pack_bytes(K, cin, cout) : 0;
  return align_up(std::max(ks, sk), 256) + pack;
}

}  // namespace pcmi

namespace pcmi {

int spconv_forward_m32(const float* in, int64_t in_ld, int64_t n_in, int cin, const float* weight, int cout,
                       const pcmi_kmap_t* map, int transpose, const float* bias, float* out, int64_t out_ld,
                       int64_t n_out, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  PCMI_REQUIRE(in && weight && out && cin > 0 && cout > 0, PCMI_ERR_INVALID, "spconv_fwd: null/empty argument");
  if (map) {
    const int64_t mi = transpose ? map->n_out : map->n_in, mo = transpose ? map->n_in : map->n_out;
    PCMI_REQUIRE(mi == n_in && mo == n_out, PCMI_ERR_INVALID, "spconv_fwd: rows (%lld -> %lld) do not match the map (%lld -> %lld)",
                 (long long)n_in, (long long)n_out, (long long)mi, (long long)mo);
  } else {
    PCMI_REQUIRE(n_in == n_out, PCMI_ERR_INVALID, "spconv_fwd: dense path needs n_in == n_out");
  }
  if (cin < 8) {
    PCMI_REQUIRE(cin == 3 && !transpose && !accumulate && (!map || map->stride == 1), PCMI_ERR_UNSUPPORTED,
                 "spconv_fwd: cin=%d only supported as the 3-channel stride-1 stem", cin);
    if (n_out == 0) return PCMI_OK;
    const int K = map ? map->K : 1;
    if (cout == 32 && K == 27 && out_ld % 4 == 0 && (uintptr_t)out % 16 == 0) {
      stem32_fwd_kernel<27><<<dim3((unsigned)ceil_div(n_out, 256)), 256, 0, st>>>(in, in_ld, weight, map->nbr, n_out, bias, out,
                                                                                 out_ld);
      PCMI_LAUNCH_CHECK();
      return PCMI_OK;
    }
    const size_t lds = sizeof(float) * K * 3 * cout;
    stem_fwd_kernel<3><<<dim3((unsigned)ceil_div(n_out, 8)), 256, lds, st>>>(in, in_ld, weight, cout, map ? map->nbr : nullptr,
                                                                            K, n_out, bias, out, out_ld);
    PCMI_LAUNCH_CHECK();
    return PCMI_OK;
  }
  // rows = output rows.  transposed conv over a stride-2 map: one offset per fine row -> pair mode
  const bool pair_mode = map && transpose;
  PCMI_REQUIRE(!(map && transpose && map->stride != 2), PCMI_ERR_UNSUPPORTED, "spconv_fwd: transposed conv needs a stride-2 map");
  return run_gathered(in, in_ld, n_in, cin, weight, cin, cout, false, cout, map, pair_mode, false, nullptr, bias, out,
                      out_ld, n_out, accumulate, ws, ws_bytes, st);
}

int spconv_backward_data_m32(const float* gout, int64_t gout_ld, int64_t n_out, int cout, const float* weight, int cin,
                             const pcmi_kmap_t* map, int transpose, float* gin, int64_t gin_ld, int64_t n_in,
                             int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  PCMI_REQUIRE(gout && weight && gin && cin >= 8 && cout > 0, PCMI_ERR_INVALID, "spconv_bwd_data: bad argument (cin=%d)", cin);
  if (!map) {
    PCMI_REQUIRE(n_in == n_out, PCMI_ERR_INVALID, "spconv_bwd_data: dense path needs n_in == n_out");
    return run_gathered(gout, gout_ld, n_out, cout, weight, cin, cout, true, cin, nullptr, false, false, nullptr, nullptr,
                        gin, gin_ld, n_in, accumulate, ws, ws_bytes, st);
  }
  const int64_t mi = transpose ? map->n_out : map->n_in, mo = transpose ? map->n_in : map->n_out;
  PCMI_REQUIRE(mi == n_in && mo == n_out, PCMI_ERR_INVALID, "spconv_bwd_data: rows do not match the map");
  if (map->stride == 1) {
    // gin[i] = sum_k gout[nbr[k][i]] @ W[mirror(k)]^T  (in and out rows coincide)
    PCMI_REQUIRE(!transpose, PCMI_ERR_UNSUPPORTED, "spconv_bwd_data: transposed stride-1 conv is not on the hot path");
    return run_gathered(gout, gout_ld, n_out, cout, weight, cin, cout, true, cin, map, false, false, map->mirror, nullptr,
                        gin, gin_ld, n_in, accumulate, ws, ws_bytes, st);
  }
  if (!transpose) {
    // strided conv: every fine (input) row has exactly one (coarse row, k): pair mode, rows = fine
    return run_gathered(gout, gout_ld, n_out, cout, weight, cin, cout, true, cin, map, true, false, nullptr, nullptr, gin,
                        gin_ld, n_in, accumulate, ws, ws_bytes, st);
  }
  // transposed conv: gin (coarse) gathers its children: nbr table, rows = coarse
  return run_gathered(gout, gout_ld, n_out, cout, weight, cin, cout, true, cin, map, false, false, nullptr, nullptr, gin,
                      gin_ld, n_in, accumulate, ws, ws_bytes, st);
}

}  // namespace pcmi

using namespace pcmi;

extern "C" {

int pcmi_spconv_fwd(const float* in, int64_t in_ld, int64_t n_in, int cin, const float* weight, int cout,
                    const pcmi_kmap_t* map, int transpose, const float* bias, float* out, int64_t out_ld,
                    int64_t n_out, void* ws, size_t ws_bytes, pcmi_stream_t stream) {
  return spconv_forward(in, in_ld, n_in, cin, weight, cout, map, transpose, bias, out, out_ld, n_out, 0, ws, ws_bytes,
                        as_stream(stream));
}

int pcmi_spconv_bwd_data(const float* gout, int64_t gout_ld, int64_t n_out, int cout, const float* weight, int cin,
                         const pcmi_kmap_t* map, int transpose, float* gin, int64_t gin_ld, int64_t n_in,
                         void* ws, size_t ws_bytes, pcmi_stream_t stream) {
  return spconv_backward_data(gout, gout_ld, n_out, cout, weight, cin, map, transpose, gin, gin_ld, n_in, 0, ws,
                              ws_bytes, as_stream(stream));
}

int pcmi_spconv_split_precision(void) { return conv16_x3_on() ? 1 : 0; }


}  // extern "C"
