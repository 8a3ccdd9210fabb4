// Weight gradients of the 3^3 / stride-1 convolutions on the BF16 matrix cores, output-tile stationary:
//     gW[k][c][n] = sum over the output rows j with a neighbour at offset k of  X[nbr[k][j]][c] * G[j][n]
//
// Why a second kernel.  wgrad_mfma_kernel (spconv_wgrad.hip) walks the pair list of ONE offset per workgroup with the
// fp32 MFMA instruction and reads both operand rows of every pair from global memory: a row is fetched once per offset
// it occurs in (~17 times; PMC: 2.25 GB per level-1 launch at 4.4 TB/s for 134 MB of unique data), and the fp32
// instruction runs at 1/16 of the bf16 rate.  Here
//   * a workgroup owns a range of output rows (in the map's mask-sorted processing order) and a GROUP of KG offsets whose
//     accumulators it keeps in registers over its whole row range: the G rows of a 64-row tile are staged ONCE per tile
//     and group, the X rows are gathered per (tile, occupied offset) -- absent offsets of a tile are skipped;
//   * both operands go through LDS as three bf16 terms each (x = h + m + l exactly, x3_split.h) in a "contraction-packed"
//     layout: one 16-byte cell = the 8 consecutive tile rows of ONE channel, so that the fragment lane (i, kk) of
//     v_mfma_f32_16x16x32_bf16 needs -- A[i][8 kk .. 8 kk + 7] = 8 rows of channel i -- is ONE ds_read_b128.  The
//     transposition (memory is row-major, the contraction runs over rows) happens in registers for free: a staging
//     lane gathers the float4 of 8 rows and then holds 8 rows x 4 channels;
//   * cells sit at [term][row group rg][channel ^ rg]: the XOR makes the 8 lanes of a ds_write_b128 service group (8 row
//     groups, same channel) and the 16 lanes of a ds_read_b128 service group (MI355X_MICROARCH.md, LDS) hit distinct
//     16-byte slots -- both conflict-free;
//   * six bf16 MFMAs per fp32-equivalent tile, small terms first (as spconv16x_kernel); fp32 accumulation.
// The row-block partial sums leave as slabs [k][row block][cin][cout] and are added in row-block order by
// wgrad_slab_sum_kernel (deterministic, no float atomics).  Two workgroups per CU (<= 80 KB of LDS, <= 256 registers):
// one stages while the other multiplies.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "internal.h"
#include "x3_split.h"

namespace pcmi {

struct WgradTArgs {
  const float* x;       // [*, x_ld] convolution input (rows addressed by the table)
  int64_t x_ld;
  const float* g;       // [*, g_ld] gradient of the convolution output
  int64_t g_ld;
  const int32_t* nbr;   // [K][n_rows] neighbour table in processing order (nbr_perm when perm is set); nullptr: identity (K = 1)
  const int32_t* perm;  // nullable: position -> output row
  int64_t n_rows;
  int K, C, N;
  int RB, NG;           // row blocks (multiple of 8), offset groups
  int tiles_per_rb;     // 64-row tiles per row block
  float* slabs;         // [K][RB][C][N]
};

template <int MTW, int NTW, int KG>
__global__ __launch_bounds__(256, 2) void wgrad_x3t_kernel(WgradTArgs a) {
  constexpr int TR = 64, RG = TR / 8;          // rows per tile, groups of 8 rows
  constexpr int CB = 32 * MTW, NB = 32 * NTW;  // channel block of the workgroup: 2 waves x MTW (NTW) tiles of 16
  constexpr uint32_t kAbsent = 0x80000000u;
  constexpr int kRsrcFlags = 0x00020000;       // raw buffer, 32-bit data format
  __shared__ __attribute__((aligned(16))) u32x4 s_x[3 * RG * CB];
  __shared__ __attribute__((aligned(16))) u32x4 s_g[3 * RG * NB];
  __shared__ uint32_t s_xoff[KG][TR];
  __shared__ uint32_t s_goff[TR];
  __shared__ int s_any[KG];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  // workgroup -> (row block, offset group): the groups of one row block share an XCD (b % 8, observed placement, speed
  // only) and are dispatched next to each other, so the rows they all read are fetched into that L2 once
  const int b = blockIdx.x;
  const int og = (b >> 3) % a.NG;
  const int rb = (b & 7) + 8 * (b / (8 * a.NG));
  const int kbase = og * KG;
  const int c0 = blockIdx.y * CB, n0 = blockIdx.z * NB;
  const int n_tiles = (int)((a.n_rows + TR - 1) / TR);
  const int t0 = rb * a.tiles_per_rb, t1 = min(t0 + a.tiles_per_rb, n_tiles);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7FFFFFFF, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, 0x7FFFFFFF, kRsrcFlags);
  const uint32_t xld = (uint32_t)(a.x_ld * 4), gld = (uint32_t)(a.g_ld * 4);

  f32x4 acc[KG][MTW][NTW];
#pragma unroll
  for (int s = 0; s < KG; ++s)
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[s][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging task of a thread: row group rg (8 consecutive tile rows) x channel quad q -- 8 consecutive lanes take the 8
  // row groups of one quad (the LDS write pattern the XOR above is made for)
  const int rg_s = t & 7, q_s = t >> 3;
  auto stage = [&](u32x4* dst, const __amdgpu_buffer_rsrc_t& rsrc, const uint32_t* offs, int ch0, int width) {
    if (q_s < width / 4) {
      v4f v[8];
      const uint32_t col = (uint32_t)(ch0 + 4 * q_s) * 4u;
#pragma unroll
      for (int e = 0; e < 8; ++e)  // an absent row has an offset >= 2^31: out of range, the load returns zeros
#if defined(PCMI_X3_DIAG_NO_GATHER)  // timing diagnostic (wrong results): every row out of range = no memory traffic
        v[e] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, kAbsent + (offs[8 * rg_s + e] & 0u) + col, 0, 0));
#else
        v[e] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, offs[8 * rg_s + e] + col, 0, 0));
#endif
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const v4f x0 = {v[0][e4], v[1][e4], v[2][e4], v[3][e4]}, x1 = {v[4][e4], v[5][e4], v[6][e4], v[7][e4]};
        u32x4 h, m, l;
#if defined(PCMI_X3_DIAG_NO_SPLIT)  // timing diagnostic (wrong results)
        h = __builtin_bit_cast(u32x4, x0); m = __builtin_bit_cast(u32x4, x1); l = h;
#else
        split3(x0, x1, h, m, l);
#endif
        const int cell = rg_s * width + ((4 * q_s + e4) ^ rg_s);
        dst[cell] = h;
        dst[RG * width + cell] = m;
        dst[2 * RG * width + cell] = l;
      }
    }
  };

  for (int tile = t0; tile < t1; ++tile) {
    const int64_t p0 = (int64_t)tile * TR;
    __syncthreads();  // the previous tile's products are done with the offsets and the staged operands
    if (t < TR) {
      const int64_t pos = p0 + t;
      uint32_t o = kAbsent;
      if (pos < a.n_rows) o = (uint32_t)(a.perm ? a.perm[pos] : (int32_t)pos) * gld;
      s_goff[t] = o;
    }
    for (int idx = t; idx < KG * TR; idx += 256) {  // a wave handles the 64 rows of one offset per round
      const int s = idx >> 6, rr = idx & 63, k = kbase + s;
      const int64_t pos = p0 + rr;
      int32_t v = -1;
      if (k < a.K && pos < a.n_rows) v = a.nbr ? a.nbr[(int64_t)k * a.n_rows + pos] : (int32_t)pos;  // (nullptr: the 1x1 convolution)
      s_xoff[s][rr] = v >= 0 ? (uint32_t)v * xld : kAbsent;
      const bool any = __ballot(v >= 0) != 0ull;
      if (lane == 0) s_any[s] = any ? 1 : 0;
    }
    __syncthreads();
    bool any_k = false;
#pragma unroll
    for (int s = 0; s < KG; ++s) any_k |= s_any[s] != 0;
    if (!any_k) continue;  // (uniform) no offset of this group occurs in the tile
    stage(s_g, gr, s_goff, n0, NB);
#pragma unroll
    for (int s = 0; s < KG; ++s) {
      if (s_any[s] == 0) continue;  // (uniform)
      stage(s_x, xr, s_xoff[s], c0, CB);
      __syncthreads();
#pragma unroll
      for (int step = 0; step < TR / 32; ++step) {
        const int rgq = 4 * step + kk;  // the 8 rows this lane quad contracts in this MFMA
        u32x4 ah[MTW], am[MTW], al[MTW];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const int cell = rgq * CB + ((16 * (wm * MTW + mt) + i) ^ rgq);
          ah[mt] = s_x[cell];
          am[mt] = s_x[RG * CB + cell];
          al[mt] = s_x[2 * RG * CB + cell];
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int cell = rgq * NB + ((16 * (wn * NTW + nt) + i) ^ rgq);
          const u32x4 bh = s_g[cell], bm = s_g[RG * NB + cell], bl = s_g[2 * RG * NB + cell];
#if defined(PCMI_X3_DIAG_NO_MFMA)  // timing diagnostic (wrong results)
#define PCMI_WX3_MFMA(AT, BT) asm volatile("" ::"v"(AT[mt]), "v"(BT))
#else
#define PCMI_WX3_MFMA(AT, BT)                                                                                              \
  acc[s][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AT[mt]), __builtin_bit_cast(bf16x8, BT), \
                                                           acc[s][mt][nt], 0, 0, 0)
#endif
          // six products per tile, the small ones first, back to back on the tile's accumulator, tile after tile (round 3
          // alternated the MTW tiles of a term: same sums, same order per accumulator, a little slower -- profiles/r04e_*)
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            PCMI_WX3_MFMA(al, bh);
            PCMI_WX3_MFMA(ah, bl);
            PCMI_WX3_MFMA(am, bm);
            PCMI_WX3_MFMA(am, bh);
            PCMI_WX3_MFMA(ah, bm);
            PCMI_WX3_MFMA(ah, bh);
          }
#undef PCMI_WX3_MFMA
        }
      }
      __syncthreads();  // s_x is restaged for the next offset (s_g / the offsets for the next tile)
    }
  }

  // ---- slabs: D[row = 4 kk + r][col = i] of every 16x16 tile; row = input channel, col = output channel ------------
#pragma unroll
  for (int s = 0; s < KG; ++s) {
    const int k = kbase + s;
    if (k >= a.K) continue;
    float* slab = a.slabs + ((int64_t)k * a.RB + rb) * a.C * a.N;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = c0 + 16 * (wm * MTW + mt) + 4 * kk + r, n = n0 + 16 * (wn * NTW + nt) + i;
          slab[(int64_t)c * a.N + n] = acc[s][mt][nt][r];
        }
  }
}

// ---- producer / consumer form (round 4) -------------------------------------------------------------------------------
// wgrad_x3t_kernel above stages and multiplies with the SAME four waves: gather (8 loads per thread) -> wait -> split -> LDS
// -> barrier -> MFMAs -> barrier, per (tile, offset).  Timing with components compiled out
// (profiles/r04e_kernel_component_removal.txt): without the gathers it runs 25 % shorter, without the MFMAs 20 % -- it is
// bound by the exposed staging phase, and its 248 registers leave no room to request the next unit's rows early (round 3
// tried: 3 instead of 4 offsets per workgroup, slower).  Here the two jobs belong to different waves of an 8-wave workgroup
// (one per CU): waves 0-3 only multiply (they own the accumulators), waves 4-7 only stage -- no accumulators, so a producer
// keeps the rows of THREE units in flight in registers (the gathers of slot q + 3 are issued before slot q + 1 is converted)
// and writes the packed image of the next unit into the other half of a double-buffered LDS while the consumers multiply
// the current one.  One workgroup barrier per (tile, offset slot); absent slots cost a barrier and nothing else.
//   slots   q = tile * KG + s; X(q) lives in s_x[q & 1] (KG is even: = s & 1), G(tile) in s_g[tile & 1]
//   tables  (row offsets of a tile's KG offsets, the G row offsets, which slots are present) for tile T are loaded by the
//           producers during tile T - 2 into a ring of three -- visible long before anybody plans with them
//   step q  producers: [s == 0: request the table of tile + 2] request X(q + 3); convert + write X(q + 1); [s == 1: convert +
//           write G(tile + 1)]; [s == 2: write the table of tile + 2]; [s == 3: request G(tile + 2)]
//           consumers: multiply slot q if present      all: barrier
// Same cells, same fragment reads, same six products in the same order per accumulator, same slabs as wgrad_x3t_kernel: the
// results are bit-identical to it for the same row-block count.
#if defined(PCMI_X3_DIAG_STAMP)  // timing diagnostic: shader-clock stamps around every per-slot barrier of workgroup 0 (one consumer, one producer wave)
__device__ unsigned long long g_x3p_stamp[2][2 * 1024];
#define PCMI_X3P_BARRIER(Q, ROLE)                                                                                  \
  do {                                                                                                             \
    const unsigned long long t_in = __builtin_readcyclecounter();                                                  \
    __syncthreads();                                                                                               \
    const unsigned long long t_out = __builtin_readcyclecounter();                                                 \
    if (blockIdx.x == 17 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && (wave == 0 || wave == 4) && (Q) < 1024) { \
      g_x3p_stamp[ROLE][2 * (Q)] = t_in;                                                                           \
      g_x3p_stamp[ROLE][2 * (Q) + 1] = t_out;                                                                      \
    }                                                                                                              \
  } while (0)
__device__ unsigned long long g_x3p_phase[4];  // producer wave 4 of that workgroup: cycles in (gather issue | X conversion + writes | G / tables | barrier)
#define PCMI_X3P_PHASE(P)                                          \
  do {                                                             \
    __builtin_amdgcn_sched_barrier(0);                             \
    const unsigned long long t_ph = __builtin_readcyclecounter();  \
    __builtin_amdgcn_sched_barrier(0);                             \
    ph_sum[((P) + 3) & 3] += t_ph - ph_last;                       \
    ph_last = t_ph;                                                \
  } while (0)
#elif defined(PCMI_X3_DIAG_NO_BARRIER)  // timing diagnostic (racy)
#define PCMI_X3P_BARRIER(Q, ROLE) do {} while (0)
#define PCMI_X3P_PHASE(P) do {} while (0)
#else
#define PCMI_X3P_BARRIER(Q, ROLE) __syncthreads()
#define PCMI_X3P_PHASE(P) do {} while (0)
#endif
template <int MTW, int NTW, int KG>
__global__ __launch_bounds__(512, 2) void wgrad_x3p_kernel(WgradTArgs a) {
  static_assert(KG % 2 == 0 && KG >= 2 && KG <= 4, "slot parity = offset parity; one producer wave per offset slot");
  constexpr int TR = 64, RG = TR / 8;
  constexpr int CB = 32 * MTW, NB = 32 * NTW;
  constexpr uint32_t kAbsent = 0x80000000u;
  constexpr int kRsrcFlags = 0x00020000;
  __shared__ __attribute__((aligned(16))) u32x4 s_x[2][3 * RG * CB];
  __shared__ __attribute__((aligned(16))) u32x4 s_g[2][3 * RG * NB];
  __shared__ uint32_t s_xoff[3][KG][TR];
  __shared__ uint32_t s_goff[3][TR];
  __shared__ int s_any[3][KG];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool producer = wave >= 4;
  const int b = blockIdx.x;
  const int og = (b >> 3) % a.NG;
  const int rb = (b & 7) + 8 * (b / (8 * a.NG));
  const int kbase = og * KG;
  const int c0 = blockIdx.y * CB, n0 = blockIdx.z * NB;
  const int n_tiles = (int)((a.n_rows + TR - 1) / TR);
  const int t0 = rb * a.tiles_per_rb, t1 = min(t0 + a.tiles_per_rb, n_tiles);
  const int nt_tiles = max(t1 - t0, 0);
  const uint32_t xld = (uint32_t)(a.x_ld * 4), gld = (uint32_t)(a.g_ld * 4);

  if (producer) {
#if defined(PCMI_X3P_PRIO_P)  // A/B: wave priority of the staging waves
    __builtin_amdgcn_s_setprio(PCMI_X3P_PRIO_P);
#endif
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7FFFFFFF, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, 0x7FFFFFFF, kRsrcFlags);
    const int pt = t - 256;
    // tables of relative tile `tl` (tile t0 + tl; past the range: everything absent) into ring slot tl % 3, in two halves:
    // the global loads are issued in one step and their results written to LDS two steps later, so that the producer
    // never waits for a table entry it has just requested (wave 4 + sx looks after offset slot sx; wave 4 also after G)
    int32_t tab_v[(KG + 3) / 4];
    int32_t tab_g = -1;
    auto table_issue = [&](int tl) {
      const int64_t p0 = (int64_t)(t0 + tl) * TR;
      const bool in_range = tl < nt_tiles;
      const int64_t pos = p0 + lane;
#pragma unroll
      for (int j = 0; j < (KG + 3) / 4; ++j) {
        const int sx = wave - 4 + 4 * j, k = kbase + sx;
        tab_v[j] = -1;
        if (sx < KG && in_range && k < a.K && pos < a.n_rows) tab_v[j] = a.nbr ? a.nbr[(int64_t)k * a.n_rows + pos] : (int32_t)pos;
      }
      // (unconditional reset + ONE guarded load: a reset inside `if (wave == 4)` becomes a select on the register's old
      //  value -- the result of the load two tiles ago as far as the compiler knows -- and with it an s_waitcnt vmcnt(0))
      tab_g = -1;
      if (wave == 4 && in_range && pos < a.n_rows) tab_g = a.perm ? a.perm[pos] : (int32_t)pos;
    };
    auto table_commit = [&](int tl) {
      const int sl = tl % 3;
#pragma unroll
      for (int j = 0; j < (KG + 3) / 4; ++j) {
        const int sx = wave - 4 + 4 * j;
        if (sx < KG) {
          s_xoff[sl][sx][lane] = tab_v[j] >= 0 ? (uint32_t)tab_v[j] * xld : kAbsent;
          const bool any = __ballot(tab_v[j] >= 0) != 0ull;
          if (lane == 0) s_any[sl][sx] = any ? 1 : 0;
        }
      }
      if (wave == 4) s_goff[sl][lane] = tab_g >= 0 ? (uint32_t)tab_g * gld : kAbsent;
    };
    // Staging task of a producer thread: row group rg_s (8 consecutive tile rows) x W / 32 consecutive channels -- ALL 256
    // producer threads have one (with channel quads only 24 x 8 = 192 of them did at 96 channels: the fourth producer wave
    // idle and the operand split -- what bounds this kernel -- on three SIMDs instead of four).  8 consecutive lanes take the
    // 8 row groups of the same channels: the LDS write pattern the XOR in the cell address is made for.
    const int rg_s = pt & 7, q_s = pt >> 3;
    auto issue = [&](auto& v, const __amdgpu_buffer_rsrc_t& rsrc, const uint32_t* offs, int ch0, auto wtag, auto e0tag, auto e1tag) {
      constexpr int W = decltype(wtag)::value, CPT = W / 32, E0 = decltype(e0tag)::value, E1 = decltype(e1tag)::value;
#if defined(PCMI_X3_DIAG_NO_GATHER)  // timing diagnostic (wrong results): every row out of range = no memory traffic
      const uint32_t col = kAbsent;
#else
      const uint32_t col = (uint32_t)(ch0 + CPT * q_s) * 4u;
#endif
#pragma unroll
      for (int e = E0; e < E1; ++e) {  // an absent row has an offset >= 2^31: out of range, the load returns zeros
#if defined(PCMI_X3_DIAG_NO_GATHER)
        const uint32_t o = (offs[8 * rg_s + e] & 0xFFFFu) | col;
#else
        const uint32_t o = offs[8 * rg_s + e] + col;
#endif
        if constexpr (CPT == 3) {
          const auto t3 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, o, 0, 0);
          static_assert(sizeof(t3) >= 12, "b96");
          __builtin_memcpy(&v[e], &t3, 12);
        } else {
          const auto t2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o, 0, 0);
          static_assert(sizeof(t2) == 8, "b64");
          __builtin_memcpy(&v[e], &t2, 8);
        }
      }
    };
    auto finish = [&](u32x4* dst, const auto& v, auto wtag, auto c0tag, auto c1tag) {
      constexpr int W = decltype(wtag)::value, CPT = W / 32, EC0 = decltype(c0tag)::value, EC1 = decltype(c1tag)::value;
#pragma unroll
      for (int ec = EC0; ec < (EC1 < CPT ? EC1 : CPT); ++ec) {
        const v4f x0 = {v[0].f[ec], v[1].f[ec], v[2].f[ec], v[3].f[ec]}, x1 = {v[4].f[ec], v[5].f[ec], v[6].f[ec], v[7].f[ec]};
        u32x4 h, m, l;
#if defined(PCMI_X3_DIAG_NO_SPLIT)  // timing diagnostic (wrong results)
        h = __builtin_bit_cast(u32x4, x0); m = __builtin_bit_cast(u32x4, x1); l = h;
#else
        split3(x0, x1, h, m, l);
#endif
#if defined(PCMI_X3_DIAG_NO_LDSW)  // timing diagnostic (wrong results): the packed cells are not written
        asm volatile("" ::"v"(h), "v"(m), "v"(l));
#else
        const int cell = rg_s * W + ((CPT * q_s + ec) ^ rg_s);
        dst[cell] = h;
        dst[RG * W + cell] = m;
        dst[2 * RG * W + cell] = l;
#endif
      }
    };
    struct Row { float f[3]; };  // the (2 or 3) channels of one gathered row
    constexpr std::integral_constant<int, CB> kWX{};
    constexpr std::integral_constant<int, NB> kWG{};
    constexpr std::integral_constant<int, 0> k0{};
    constexpr std::integral_constant<int, 8> k8{};
    // FOUR register sets for the gathered rows (the producers own no accumulators): the rows of slot q + 3 are requested in
    // step q and converted in step q + 2 -- two full steps in flight (with two sets / one step the step lasted as long as
    // a gather under load: 0.54 ms per level-0 launch, profiles/r04j_*); slot s of a tile uses set s (KG = 4).  The G rows
    // of tile + 2 are requested in a tile's last step and converted in the next tile's second.
    static_assert(KG == 4, "one register set per offset slot");
    Row r0[8], r1[8], r2[8], r3[8], rgv[8];
    // Requests and conversions are UNCONDITIONAL: an absent slot has nothing but out-of-range offsets in its table (no
    // memory traffic, zeros come back) and is converted like any other -- a few hundred wasted VALU cycles in waves that
    // have slack.  With the requests under `if (present)` the compiler cannot pair a request with its conversion (two
    // reads of the same flag), assumes requests that were never consumed and guards every reuse of the ring registers
    // with s_waitcnt vmcnt(0) -- which exposes the latency of the gathers it has just issued.
    // Round 5 measured the other half: ONLY the conversion under a wave-uniform `if (slot present)` (requests still
    // unconditional; hipcc -S: no vmcnt(0), the loop's waits stay vmcnt(16..31), 230 registers as before).  39 % of the
    // slots of the bench batch are absent, so this removes 39 % of the producers' split work -- and the kernel got
    // SLOWER: level-1 96 -> 96 0.498 -> 0.537 ms, level 2 0.130 -> 0.138 ms, the step unchanged (264.1 vs 265.0
    // pairs/s, inside the run-to-run spread): profiles/r05b_wgrad_x3p_skip_absent_conversion_ab.txt.  The conversion of
    // zeros in an absent slot is NOT what the kernel waits for (the consumers have nothing to do in such a slot and
    // the producers' next request has been in flight for two steps); the present slots are, where a producer's
    // conversion and a consumer's 36 fragment reads + 108 products share a SIMD and the LDS pipe.
    table_issue(0);
    table_commit(0);
    table_issue(1);
    table_commit(1);
    __syncthreads();  // (B0) the first two tables are visible
    issue(r0, xr, s_xoff[0][0], c0, kWX, k0, k8);
    issue(rgv, gr, s_goff[0], n0, kWG, k0, k8);
    issue(r1, xr, s_xoff[0][1], c0, kWX, k0, k8);
    issue(r2, xr, s_xoff[0][2], c0, kWX, k0, k8);
    finish(s_x[0], r0, kWX, k0, k8);
    finish(s_g[0], rgv, kWG, k0, k8);
    issue(rgv, gr, s_goff[1], n0, kWG, k0, k8);  // G rows of the second tile (converted in step 1)
    __syncthreads();  // (B1) slot 0 and the G rows of the first tile are staged
#define PCMI_X3P_SET(j) ((j) == 0 ? r0 : ((j) == 1 ? r1 : ((j) == 2 ? r2 : r3)))
#if defined(PCMI_X3_DIAG_STAMP)
    unsigned long long ph_sum[4] = {0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#endif
    for (int tl = 0; tl < nt_tiles; ++tl) {
#pragma unroll
      for (int sx = 0; sx < KG; ++sx) {
        PCMI_X3P_PHASE(0);
        if (sx == 0) table_issue(tl + 2);
        {  // rows of slot q + 3 into the register set slot q - 1 has left (converted two steps ago)
          const int s3 = (sx + 3) % KG, tl3 = tl + (sx + 3) / KG;
          issue(PCMI_X3P_SET(s3), xr, s_xoff[tl3 % 3][s3], c0, kWX, k0, k8);
        }
        PCMI_X3P_PHASE(1);
        // slot q + 1 (requested two steps ago): convert and write into the X buffer the consumers are not reading.
        // (Round 6 measured the order of the two: the 8 requests of a wave take ~780 of a step's 2830 cycles -- four waves'
        //  gathers queue on the CU's one address unit -- and neither the conversion first nor a channel's conversion behind
        //  every few requests hides that: 0.491 / 0.436 against 0.435 ms per level-1 launch, profiles/r06pq_*.)
        finish(s_x[(sx + 1) & 1], PCMI_X3P_SET((sx + 1) % KG), kWX, k0, k8);
        PCMI_X3P_PHASE(2);
        if (sx == 1) finish(s_g[(tl + 1) & 1], rgv, kWG, k0, k8);
        if (sx == KG - 2) table_commit(tl + 2);  // (first read one step on: a barrier away)
        if (sx == KG - 1) issue(rgv, gr, s_goff[(tl + 2) % 3], n0, kWG, k0, k8);  // (its table: written one step ago)
        PCMI_X3P_PHASE(3);
        PCMI_X3P_BARRIER(tl * KG + sx, 1);
      }
    }
#if defined(PCMI_X3_DIAG_STAMP)
    if (blockIdx.x == 17 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && wave == 4)
      for (int e = 0; e < 4; ++e) g_x3p_phase[e] = ph_sum[e];
#endif
#undef PCMI_X3P_SET
    return;
  }

  // ---- consumers: waves 0-3, 2 x 2 over the [CB x NB] block, MTW x NTW tiles of 16 x 16 each ---------------------------
#if defined(PCMI_X3P_PRIO_C)  // A/B: wave priority of the multiplying waves
  __builtin_amdgcn_s_setprio(PCMI_X3P_PRIO_C);
#endif
  const int i = lane & 15, kk = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  f32x4 acc[KG][MTW][NTW];
#pragma unroll
  for (int sx = 0; sx < KG; ++sx)
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[sx][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();  // (B0)
  __syncthreads();  // (B1)
  for (int tl = 0; tl < nt_tiles; ++tl) {
    const u32x4* sg = s_g[tl & 1];
#pragma unroll
    for (int sx = 0; sx < KG; ++sx) {
      if (s_any[tl % 3][sx] != 0) {  // (uniform)
        const u32x4* sxp = s_x[sx & 1];
        // One consumer wave per SIMD: nobody else covers an LDS round trip, so the fragment reads are pipelined by hand --
        // the B fragments of column tile nt + 1 (of the next 32-row step behind the last one) are requested before the 18
        // MFMAs of tile nt are issued, pinned with sched_barrier (left to the compiler: ~18 lgkmcnt(0) waits per unit right
        // behind their reads, ~2000 stall cycles against 1836 of MFMA issue).  The A fragments of a step stay live across
        // its column tiles; the second step's are read behind the first step's last tile (no registers to hold both).
        u32x4 ah[MTW], am[MTW], al[MTW];
        u32x4 bh[2], bm[2], bl[2];
        auto read_a = [&](int step) {
          const int rgq = 4 * step + kk;  // the 8 rows this lane quad contracts in this MFMA
#if defined(PCMI_X3_DIAG_NO_FRAG)  // timing diagnostic (wrong results): no fragment reads from LDS
          if (step >= 0) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) asm volatile("" : "+v"(ah[mt]), "+v"(am[mt]), "+v"(al[mt]));
            return;
          }
#endif
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            const int cell = rgq * CB + ((16 * (wm * MTW + mt) + i) ^ rgq);
            ah[mt] = sxp[cell];
            am[mt] = sxp[RG * CB + cell];
            al[mt] = sxp[2 * RG * CB + cell];
          }
        };
        auto read_b = [&](int step, int nt, int slot) {
          const int rgq = 4 * step + kk;
#if defined(PCMI_X3_DIAG_NO_FRAG)
          if (step >= 0) {
            asm volatile("" : "+v"(bh[slot]), "+v"(bm[slot]), "+v"(bl[slot]));
            return;
          }
#endif
          const int cell = rgq * NB + ((16 * (wn * NTW + nt) + i) ^ rgq);
          bh[slot] = sg[cell];
          bm[slot] = sg[RG * NB + cell];
          bl[slot] = sg[2 * RG * NB + cell];
        };
        constexpr int STEPS = TR / 32;
        read_a(0);
        read_b(0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < STEPS * NTW; ++it) {
          const int step = it / NTW, nt = it % NTW, slot = it & 1;
          if (it + 1 < STEPS * NTW) read_b((it + 1) / NTW, (it + 1) % NTW, slot ^ 1);
          __builtin_amdgcn_sched_barrier(0);
#if defined(PCMI_X3_DIAG_NO_MFMA)  // timing diagnostic (wrong results)
#define PCMI_WX3P_MFMA(AT, BT) asm volatile("" ::"v"(AT[mt]), "v"(BT[slot]))
#else
#define PCMI_WX3P_MFMA(AT, BT)                                                                                                      \
  acc[sx][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AT[mt]), __builtin_bit_cast(bf16x8, BT[slot]), \
                                                            acc[sx][mt][nt], 0, 0, 0)
#endif
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            PCMI_WX3P_MFMA(al, bh);
            PCMI_WX3P_MFMA(ah, bl);
            PCMI_WX3P_MFMA(am, bm);
            PCMI_WX3P_MFMA(am, bh);
            PCMI_WX3P_MFMA(ah, bm);
            PCMI_WX3P_MFMA(ah, bh);
#if defined(PCMI_X3_DIAG_B2B)  // A/B: the six products of an accumulator back to back (the compiler interleaves the three)
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
#undef PCMI_WX3P_MFMA
          __builtin_amdgcn_sched_barrier(0);
          if (nt == NTW - 1 && step + 1 < STEPS) {
            read_a(step + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      PCMI_X3P_BARRIER(tl * KG + sx, 0);
    }
  }
  // ---- slabs: D[row = 4 kk + r][col = i] of every 16x16 tile; row = input channel, col = output channel ------------
#pragma unroll
  for (int sx = 0; sx < KG; ++sx) {
    const int k = kbase + sx;
    if (k >= a.K) continue;
    float* slab = a.slabs + ((int64_t)k * a.RB + rb) * a.C * a.N;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = c0 + 16 * (wm * MTW + mt) + 4 * kk + r, n = n0 + 16 * (wn * NTW + nt) + i;
          slab[(int64_t)c * a.N + n] = acc[sx][mt][nt][r];
        }
  }
}

// Round 6 measured where a slot of wgrad_x3p_kernel goes (shader-clock stamps, -DPCMI_X3_DIAG_STAMP, profiles/r06l_*): at level 1,
// 96 -> 96, a slot lasts 3400 cycles of which the consumers multiply for 2050 (108 MFMAs = 1836) and WAIT for 1360; the four
// producer waves need 3300 -- 750 to read the offsets and issue 8 gathers, 1860 to convert 24 elements per lane (132 VALU
// operations) and write 9 cells, 440 for their share of the G rows and the tables.  The kernel is bound by the producers' VALU
// work, at ~7 cycles per instruction beside the consumers' MFMA stream.  What followed from that:
//   * -fno-slp-vectorize (build.py): hipcc had packed the split's subtractions into v_pk_add_f32 (+ two v_mov per pair), which
//     costs +13 cycles each beside MFMAs: 0.504 -> 0.438 ms per level-1 launch, the step -2.4 % (profiles/r06n_*);
//   * a twelve-wave form (EIGHT producer waves, an offset PAIR per workgroup so that the consumers' accumulators fit 168
//     registers) was built, passed the parity tests and was SLOWER (0.503 against 0.438 ms alone, the step +3 %): the VALU work
//     per SIMD is the same whichever wave carries it, and the G rows are staged for twice as many offset groups.  Removed;
//   * with NO producer work at all (split, gather traffic and LDS writes compiled out) the step gains 0.3 ms of 14.0
//     (profiles/r06j_*): the bound on anything a pre-split operand image could buy, before its own passes are paid for.

// gW[k][e] (+)= sum over the row blocks of slab[k][rb][e], in row-block order.  32 elements x 8 row-block lanes per
// workgroup, folded through LDS (the chain of dependent loads is RB / 8 long).
__global__ __launch_bounds__(256) void wgrad_slab_sum_kernel(const float* __restrict__ slabs, int RB, int64_t per_k,
                                                             float* __restrict__ gw, int accumulate) {
  __shared__ float s_part[8][33];
  const int k = blockIdx.y;
  const int el = threadIdx.x & 31, cl = threadIdx.x >> 5;
  const int64_t e = (int64_t)blockIdx.x * 32 + el;
  float s = 0.f;
  if (e < per_k)
    for (int r = cl; r < RB; r += 8) s += slabs[((int64_t)k * RB + r) * per_k + e];
  s_part[cl][el] = s;
  __syncthreads();
  if (cl != 0 || e >= per_k) return;
#pragma unroll
  for (int q = 1; q < 8; ++q) s += s_part[q][el];
  float* dst = gw + (int64_t)k * per_k + e;
  *dst = accumulate ? *dst + s : s;
}

// ---- host side ----------------------------------------------------------------------------------------------------
constexpr int kWgradTKG = 4;     // offsets whose accumulators a workgroup holds
constexpr int kWgradTMaxRB = 128;

static int wgrad_x3t_tw(int c) { return c % 96 == 0 ? 3 : (c % 64 == 0 ? 2 : 0); }  // 16-wide tiles per wave and axis

// PCMI_WGRAD_X3T: minimum number of rows for the tile-stationary split-precision kernel (0 = never; 1 = every size).
// Read per call: the parity test runs both kernels in one process.
// Measured on the bench batch, ms per launch stand-alone, pair-list fp32 kernel -> this one (profiles/r03b_kbench_*):
//   40k rows   96->96 0.205 -> 0.139, 128->96 0.253 -> 0.223
//   175k rows  96->96 0.509 -> 0.514, 128->96 0.732 -> 0.901   (both kernels are bound by the gathered-row traffic there:
//              2.3 GB at 4.4 TB/s against 1.8 GB at 3.4 TB/s; this one re-stages the G tile once per offset group)
//   9.9k rows  64->64 0.026 -> 0.048, 128->128 0.073 -> 0.098, 192->128 0.160 -> 0.121
// and in the training step, where it shares the chip with the backward chain (profiles/r03i_bench_ab_x3t_window.txt, 3
// processes each): rows in [16384, 100000] 248.7-250.3 pairs/s, >= 8192 251.4-252.5, >= 16384 **252.0-252.9** -- at 175k
// rows it is no faster alone but asks a quarter less of the memory system, which the chain's kernels get.
// Round 4, with the producer / consumer form (profiles/r04n_*, 2 processes each): >= 100000 rows 256.0, >= 16384 257.9,
// >= 8192 **258.5**, >= 2048 255.7 pairs/s.
static int64_t wgrad_x3t_min_rows() {
  const char* e = getenv("PCMI_WGRAD_X3T");
  return e ? (int64_t)atoll(e) : (int64_t)8192;
}

// PCMI_WGRAD_X3T_DENSE=0: the 1x1 convolutions (no map: K = 1, identity table) stay on the pair-list kernel (A/B)
static bool wgrad_x3t_dense_on() {
  const char* e = getenv("PCMI_WGRAD_X3T_DENSE");
  return !(e && e[0] == '0');
}

// map == nullptr: a 1x1 convolution, gW = X^T G over all rows -- the same kernel with K = 1 and the identity table
// (round 5: the level-1 128 -> 96 downsample gradient took 350 us in the step on the fp32 pair-list kernel, 12 TFLOP/s).
bool wgrad_x3t_eligible(const pcmi_kmap_t* map, int64_t n_in, int64_t n_out, int cin, int cout, int64_t in_ld, int64_t gout_ld) {
  const int64_t mr = wgrad_x3t_min_rows();
  const bool shape = map ? (map->kernel_size == 3 && map->stride == 1) : wgrad_x3t_dense_on();
  return shape && mr > 0 && n_out >= mr && n_in == n_out && cin >= 64 && cout >= 64 &&
         wgrad_x3t_tw(cin) > 0 && wgrad_x3t_tw(cout) > 0 && n_in * in_ld * 4 <= 0x7FFFFF00ll && n_out * gout_ld * 4 <= 0x7FFFFF00ll;
}

// PCMI_WGRAD_X3P: 1 = the producer / consumer form (wgrad_x3p_kernel: 8 waves, one workgroup per CU; default), 0 = one role
// (wgrad_x3t_kernel); read per call
static int wgrad_x3p_mode() {
  const char* e = getenv("PCMI_WGRAD_X3P");
  return e ? (atoi(e) != 0 ? 1 : 0) : 1;
}

static int wgrad_x3t_rb(int64_t n_rows, int gy, int gz, int per_cu = 2, int K = PCMI_MAX_KERNEL_VOLUME, int kg = kWgradTKG) {
  const int NG = (K + kg - 1) / kg;
  const int64_t n_tiles = ceil_div(n_rows, 64);
  // `per_cu` resident workgroups per CU (wgrad_x3t_kernel: two -- one per CU: 238.6 against 240 pairs/s in the step; the
  // producer / consumer form is one 8-wave workgroup per CU); one round of them, at least 4 tiles per workgroup, a
  // multiple of 8 row blocks
  int64_t rb = (int64_t)per_cu * num_cu() / ((int64_t)NG * gy * gz);
  rb = std::min<int64_t>(rb, n_tiles / 4);
  rb = std::max<int64_t>(8, std::min<int64_t>(kWgradTMaxRB, rb / 8 * 8));
  return (int)rb;
}

size_t wgrad_x3t_workspace(int64_t n_rows, int cin, int cout) {
  const int MTW = wgrad_x3t_tw(cin), NTW = wgrad_x3t_tw(cout);
  const int64_t mr = wgrad_x3t_min_rows();
  if (MTW == 0 || NTW == 0 || mr <= 0 || n_rows < mr || cin < 64 || cout < 64) return 0;
  return (size_t)PCMI_MAX_KERNEL_VOLUME * wgrad_x3t_rb(n_rows, cin / (32 * MTW), cout / (32 * NTW)) * cin * cout * sizeof(float);
}

template <int MTW, int NTW>
static void launch_x3t(const WgradTArgs& a, dim3 grid, int mode, hipStream_t st) {
  if (mode == 1) wgrad_x3p_kernel<MTW, NTW, kWgradTKG><<<grid, 512, 0, st>>>(a);
  else wgrad_x3t_kernel<MTW, NTW, kWgradTKG><<<grid, 256, 0, st>>>(a);
}

int wgrad_x3t_run(const float* in, int64_t in_ld, const float* gout, int64_t gout_ld, int64_t n_rows, int cin, int cout,
                  const pcmi_kmap_t* map, float* gweight, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  const int MTW = wgrad_x3t_tw(cin), NTW = wgrad_x3t_tw(cout);
  PCMI_REQUIRE(MTW > 0 && NTW > 0 && (!map || map->K <= PCMI_MAX_KERNEL_VOLUME), PCMI_ERR_UNSUPPORTED, "wgrad x3t: channels (%d, %d)", cin, cout);
  WgradTArgs a;
  a.x = in;
  a.x_ld = in_ld;
  a.g = gout;
  a.g_ld = gout_ld;
  a.nbr = !map ? nullptr : ((map->perm && map->nbr_perm) ? map->nbr_perm : map->nbr);
  a.perm = (map && map->perm && map->nbr_perm) ? map->perm : nullptr;
  a.n_rows = n_rows;
  const int K = map ? map->K : 1;
  a.K = K;
  a.C = cin;
  a.N = cout;
  const int gy = cin / (32 * MTW), gz = cout / (32 * NTW);
  const int mode = wgrad_x3p_mode();
  [[maybe_unused]] const bool pc = mode == 1;
  const int kg = kWgradTKG;
  a.NG = (K + kg - 1) / kg;
  a.RB = wgrad_x3t_rb(n_rows, gy, gz, mode >= 1 ? 1 : 2, K, kg);
  // (the producer / consumer form takes 8 x NG x (RB / 8) = 224 of the 256 CUs at NG = 7: with every CU used -- 36 row
  //  blocks, 252 workgroups -- it is 11 % faster alone (0.501 against 0.560 ms) and SLOWER in the step (253.1-254.6 against
  //  256.6-256.9 pairs/s): the CUs it leaves are where the backward chain's kernels run meanwhile; fewer row blocks lose
  //  again -- 24: 254.6, 16: 252.7.  profiles/r04jk_wgrad_producer_consumer.txt)
  a.tiles_per_rb = (int)ceil_div(ceil_div(n_rows, 64), a.RB);
  const size_t need = (size_t)K * a.RB * cin * cout * sizeof(float);
  PCMI_REQUIRE(ws && ws_bytes >= need, PCMI_ERR_WORKSPACE, "wgrad x3t: workspace %zu < %zu bytes", ws_bytes, need);
  a.slabs = (float*)ws;
  const dim3 grid((unsigned)(a.RB * a.NG), (unsigned)gy, (unsigned)gz);
  switch (MTW * 4 + NTW) {
    case 3 * 4 + 3: launch_x3t<3, 3>(a, grid, mode, st); break;
    case 3 * 4 + 2: launch_x3t<3, 2>(a, grid, mode, st); break;
    case 2 * 4 + 3: launch_x3t<2, 3>(a, grid, mode, st); break;
    default: launch_x3t<2, 2>(a, grid, mode, st); break;
  }
  PCMI_LAUNCH_CHECK();
#if defined(PCMI_X3_DIAG_STAMP)
  if (pc) {
    static unsigned long long h[2][2048];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_x3p_stamp), sizeof(h));
    const int nq = std::min(1024, a.tiles_per_rb * kWgradTKG);
    double work[2] = {0, 0}, wait[2] = {0, 0};
    int cnt = 0;
    for (int q = 8; q + 8 < nq; ++q, ++cnt)
      for (int r = 0; r < 2; ++r) {
        work[r] += (double)(h[r][2 * q] - h[r][2 * q - 1]);
        wait[r] += (double)(h[r][2 * q + 1] - h[r][2 * q]);
      }
    unsigned long long php[4];
    (void)hipMemcpyFromSymbol(php, HIP_SYMBOL(g_x3p_phase), sizeof(php));
    const double nsl = (double)a.tiles_per_rb * kWgradTKG;
    fprintf(stderr, "x3p phases: C %d rows %lld | per slot: table + gather issue %.0f | X convert+write %.0f | G/tables %.0f | barrier %.0f\n", cin, (long long)n_rows,
            php[0] / nsl, php[1] / nsl, php[2] / nsl, php[3] / nsl);
    if (cnt > 0)
      fprintf(stderr, "x3p stamp: rows %lld C %d N %d K %d | slots %d | consumer work %.0f wait %.0f | producer work %.0f wait %.0f | slot %.0f cycles\n",
              (long long)n_rows, cin, cout, K, cnt, work[0] / cnt, wait[0] / cnt, work[1] / cnt, wait[1] / cnt,
              (double)(h[0][2 * (nq - 9)] - h[0][2 * 8]) / (nq - 17));
  }
#endif
  const int64_t per_k = (int64_t)cin * cout;
  wgrad_slab_sum_kernel<<<dim3((unsigned)ceil_div(per_k, 32), (unsigned)K), 256, 0, st>>>(a.slabs, a.RB, per_k, gweight, accumulate);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

}  // namespace pcmi
