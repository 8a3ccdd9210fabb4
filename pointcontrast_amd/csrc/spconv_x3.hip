// fp32 sparse convolution on the BF16 matrix cores of gfx950: three-term operand split ("bf16x3").
//
// Why.  gfx950 has no reduced-precision fast path for fp32 inputs (no xf32): v_mfma_f32_16x16x4_f32 runs at the fp32
// VECTOR rate, 1/16 of the bf16 rate (MI355X_MICROARCH.md: 157 TFLOP/s against 2.5 PFLOP/s).  The reference's
// arithmetic is fp32 and north_star holds the features to 1e-4, so plain bf16 is not an option -- but an fp32 number
// is the EXACT sum of three bf16 numbers,
//     x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)        (round-to-nearest-even, 3 x 8 bits)
// and a product of two such sums, truncated after the terms of relative size 2^-16,
//     a b  ~  a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_m b_m + a_l b_h)       (dropped: <= 2^-23 |a b|)
// is six bf16 MFMAs with fp32 accumulation: 6/16 of the matrix-pipe time of the fp32 instruction for the same tile,
// at fp32-class accuracy (every bf16 x bf16 product is exact in fp32; one rounding per MFMA instead of one per
// product).  tests/test_gpu_parity.py::test_conv16_x3_* measures both forms against an fp64 contraction.
//
// What.  spconv16x_kernel is spconv16p_kernel (spconv.hip: 128-row tiles, two 16-row groups per wave, per-group offset
// skipping, byte-offset neighbour table, raw buffer loads issued from inside the MFMA stream, unit-balanced launch)
// with the inner product replaced:
//   * weights: x3_pack_kernel splits them ONCE per launch into the kernel's LDS image -- per (slice k, 32-channel chunk,
//     NS-wide output slice) a block [term h|m|l][16-column tile][lane quad kk][column][8 bf16] -- so staging a chunk is a
//     linear 16-byte copy and a B fragment (8 bf16 = the 8 channels lane (j, kk) contracts) is ONE ds_read_b128, conflict-
//     free by the service groups of that instruction (x3_piece);
//   * gathered rows: fp32 in memory as before (same HBM / L2 traffic); the 8 floats a lane holds per group and chunk are
//     split in registers (v_cvt_pk_bf16_f32, 5.5 VALU operations per element, issued beside the other waves' MFMAs);
//   * v_mfma_f32_16x16x32_bf16: lane (i = l & 15, kk = l >> 4) supplies A[i][8 kk .. 8 kk + 7] and B[8 kk ..][j = i];
//     which eight channels those are is free as long as A and B agree: channel(kk, e) = 4 kk + (e & 3) + 16 (e >> 2)
//     (the two float4 the lane gathers).  C/D layout as the fp32 16x16 form: D[row = 4 kk + r][col = i].
// Default for the matrix-bound launches of the 16-row kernel (>= 64 channels on both sides); PCMI_CONV16_X3=0 restores
// the fp32-MFMA kernel.  Measured (profiles/r02_x3_*): level-1 96->96 on the 175k-row pair tensor 0.530 -> 0.343 ms
// (161 TFLOP/s of fp32-equivalent work: above the 157.3 TFLOP/s peak of the fp32 instruction), 128->128 at level 2
// 0.341 -> 0.170 ms; max |difference| to the fp32 kernel 0.9-2.9e-6 of the largest output.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "internal.h"
#include "spconv_args.h"
#include "x3_split.h"

namespace pcmi {

// 16-byte piece of column n (of the NS-wide output slice) and lane quad kk inside one term of a weight block:
// [16-column tile][kk][column in the tile].  The fragment read of lane (i, kk) for column tile ct is then piece
// 64 ct + 16 kk + i: the 16 lanes of every ds_read_b128 service group ({0-3, 12-15, 20-27}, ... -- MI355X_MICROARCH.md,
// LDS) cover 16 different 16-byte slots.  (Round 2's order [column][kk] put lanes i and i + 4 of a quad on the same
// slot: every fragment read was a 2-way bank conflict.)
__host__ __device__ constexpr int x3_piece(int n, int kk) { return (n >> 4) * 64 + kk * 16 + (n & 15); }

// channel of chunk-local element e of lane quad kk (see the header)
__host__ __device__ constexpr int x3_channel(int kk, int e) { return 4 * kk + (e & 3) + 16 * (e >> 2); }

// One thread per (slice, chunk, output slice, column, lane quad): the three 16-byte pieces of that lane's B fragment.
__global__ __launch_bounds__(256) void x3_pack_kernel(const float* __restrict__ w, int64_t w_kstride, int64_t w_sc, int64_t w_sn,
                                                      int K, int C, int N, int NS, u32x4* __restrict__ out) {
  const int nch = C / kKC, nns = N / NS;
  const int64_t total = (int64_t)K * nch * nns * NS * 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int kk = (int)(idx & 3);
  int64_t r = idx >> 2;
  const int nl = (int)(r % NS);
  r /= NS;
  const int ns = (int)(r % nns);
  r /= nns;
  const int cc = (int)(r % nch);
  const int wk = (int)(r / nch);
  const float* wb = w + (int64_t)wk * w_kstride + (int64_t)(ns * NS + nl) * w_sn;
  v4f x0, x1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    x0[e] = wb[(int64_t)(cc * kKC + x3_channel(kk, e)) * w_sc];
    x1[e] = wb[(int64_t)(cc * kKC + x3_channel(kk, e + 4)) * w_sc];
  }
  u32x4 h, m, l;
  split3(x0, x1, h, m, l);
  u32x4* blk = out + (((int64_t)wk * nch + cc) * nns + ns) * (3 * NS * 4);
  const int piece = x3_piece(nl, kk);
  blk[0 * NS * 4 + piece] = h;
  blk[1 * NS * 4 + piece] = m;
  blk[2 * NS * 4 + piece] = l;
}

// DMA: the weight block of the next step goes global -> LDS directly (buffer_load_dwordx4 ... lds: the block is a
// linear image, a wave instruction copies 1 KiB) instead of through 4 x BR staging registers per thread -- at NT = 3
// that is what lets the kernel fit 3 waves per SIMD without spilling (185 -> 165 VGPRs).
// The same for many (layer, orientation) jobs in one launch: a thread finds its job by bisection over the item prefix.
__global__ __launch_bounds__(256) void x3_pack_many_kernel(const X3PackJob* __restrict__ jobs, int n_jobs, int64_t total,
                                                           int64_t item0) {
  const int64_t idx = item0 + (int64_t)blockIdx.x * 256 + threadIdx.x;  // items [item0, total) of the job list
  if (idx >= total) return;
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {  // last job whose first item is <= idx
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_item <= idx) lo = mid; else hi = mid - 1;
  }
  const X3PackJob jb = jobs[lo];
  const int64_t loc = idx - jb.first_item;
  const int nch = jb.C / kKC, nns = jb.N / jb.NS;
  const int kk = (int)(loc & 3);
  int64_t r = loc >> 2;
  const int nl = (int)(r % jb.NS);
  r /= jb.NS;
  const int ns = (int)(r % nns);
  r /= nns;
  const int cc = (int)(r % nch);
  const int wk = (int)(r / nch);
  const float* wb = jb.w + (int64_t)wk * jb.w_kstride + (int64_t)(ns * jb.NS + nl) * jb.w_sn;
  v4f x0, x1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    x0[e] = wb[(int64_t)(cc * kKC + x3_channel(kk, e)) * jb.w_sc];
    x1[e] = wb[(int64_t)(cc * kKC + x3_channel(kk, e + 4)) * jb.w_sc];
  }
  u32x4 h, m, l;
  split3(x0, x1, h, m, l);
  u32x4* blk = reinterpret_cast<u32x4*>(jb.out) + (((int64_t)wk * nch + cc) * nns + ns) * (3 * jb.NS * 4);
  const int piece = x3_piece(nl, kk);
  blk[0 * jb.NS * 4 + piece] = h;
  blk[1 * jb.NS * 4 + piece] = m;
  blk[2 * jb.NS * 4 + piece] = l;
}

// -DPCMI_X3_DIAG_NO_{GATHER,DMA,SPLIT,BFRAG,MFMA,BARRIER}: timing diagnostics -- one component compiled out, wrong results, for
// stand-alone timing only (pointcontrast_amd.build.build_variant + PCMI_LIB; profiles/r04e_kernel_component_removal.txt
// is what they showed: no single component bounds the kernel, and it is NOT memory latency -- requesting the gathers, or
// gathers and weight blocks, two steps ahead changed nothing: profiles/r04b_*, r04f_*).
#if defined(PCMI_X3_DIAG_STAMP)  // timing diagnostic: shader-clock totals of one wave's phases (workgroup 17, wave 0)
__device__ unsigned long long g_x3c_phase[8];
#define PCMI_X3C_PHASE(P)                                          \
  do {                                                             \
    __builtin_amdgcn_sched_barrier(0);                             \
    const unsigned long long t_ph = __builtin_readcyclecounter();  \
    __builtin_amdgcn_sched_barrier(0);                             \
    ph_sum[P] += t_ph - ph_last;                                   \
    ph_last = t_ph;                                                \
  } while (0)
#else
#define PCMI_X3C_PHASE(P) do {} while (0)
#endif
template <int NT, bool SK, bool DMA>
__global__ __launch_bounds__(256, NT == 4 ? 2 : 3) void spconv16x_kernel(ConvArgs a) {
  constexpr int TM = 128, NS = 32 * NT, CTN = 2 * NT;  // rows per tile, output slice, 16-wide column tiles
  constexpr int KSLOTS = PCMI_MAX_KERNEL_VOLUME;
  constexpr uint32_t kAbsent = 0x80000000u;
  constexpr int kRsrcFlags = 0x00020000;  // raw buffer, 32-bit data format
  constexpr int PIECES = 3 * NS * 4;      // 16-byte pieces of one weight block
  constexpr int BLOCK_BYTES = PIECES * 16;
  constexpr int BR = (PIECES + 255) / 256;  // staging rounds of a thread
  static_assert(CTN >= 3, "the three load stages sit behind the MFMAs of column tiles 0, 1, 2");
  __shared__ __attribute__((aligned(16))) u32x4 s_b[2][PIECES];
  __shared__ uint32_t s_off[KSLOTS][TM];
  __shared__ int32_t s_orow[TM];
  __shared__ int2 s_kmeta[KSLOTS];  // per occupied offset: {row of s_off, byte offset of its first weight block}
  __shared__ int32_t s_kabs[KSLOTS];
  __shared__ int32_t s_nk;
  __shared__ int64_t s_tile[2];

  const int n0 = blockIdx.y * NS;
  const int nch = a.C / kKC;
  const uint32_t chunk_bytes = (uint32_t)gridDim.y * BLOCK_BYTES;  // blocks of one chunk: one per output slice
  const uint32_t wk_bytes = (uint32_t)nch * chunk_bytes;
  const uint32_t slice_bytes = (uint32_t)blockIdx.y * BLOCK_BYTES;
  int sk_g = 0, sk_tile = 0, sk_u = 0, sk_u1 = 0;
  bool sk_first = true;
  if constexpr (SK) {
    const int G = (int)gridDim.x;
    sk_g = (int)(blockIdx.x & 7) * (G / 8) + (int)(blockIdx.x >> 3);
    const int U = a.sk_pref[a.sk_tiles] * nch;  // shares are counted in chunk steps (see spconv16p_kernel)
    const int per = (U + G - 1) / G;
    sk_u = sk_g * per;
    sk_u1 = min(U, sk_u + per);
    if (sk_u >= sk_u1) return;
    if (threadIdx.x < 64) {  // last tile whose first step is <= sk_u: 64-ary search
      int lo = 0, n = a.sk_tiles;
      while (n > 1) {
        const int stride = (n + 63) / 64;
        const int probe = lo + (int)threadIdx.x * stride;
        const bool le = probe < lo + n && a.sk_pref[probe] * nch <= sk_u;
        const int cnt = __popcll(__ballot(le));
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
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wpack), 0, 0x7FFFFFFF, kRsrcFlags);
  const uint32_t ld_bytes = (uint32_t)(a.x_ld * 4);
  const int sk_total = SK ? sk_u1 - sk_u : 0;
  int sk_done = 0, prio_qtr = -1;
#if defined(PCMI_X3_DIAG_STAMP)
  unsigned long long ph_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#endif
  for (;;) {  // one pass per tile piece (exactly one when !SK)
  bool sk_whole = true;
  int sk_next_u = 0, sk_c0 = 0, sk_steps = 0;
  int t = threadIdx.x;
  if constexpr (SK) asm volatile("" : "+v"(t));  // see spconv_mfma_kernel
  const int lane = t & 63, wave = t >> 6;
  const int i = lane & 15, kk = lane >> 4;
  if constexpr (SK) {
    const int64_t row0 = (int64_t)sk_tile * TM;
    const uint32_t tmask = a.sk_mask[sk_tile];
    const int pref = a.sk_pref[sk_tile] * nch, ns_t = __popc(tmask) * nch;
    const int jb = sk_u - pref, je = min(sk_u1 - pref, ns_t);
    sk_whole = (jb == 0 && je == ns_t);
    sk_next_u = pref + je;
    const int o0 = jb / nch, o1 = (je - 1) / nch;
    sk_c0 = jb - o0 * nch;
    sk_steps = je - jb;
    if (t < a.K && ((tmask >> t) & 1u)) {
      const int j = __popc(tmask & ((1u << t) - 1u));
      if (j >= o0 && j <= o1) {
        s_kabs[j - o0] = t;
        s_kmeta[j - o0] = make_int2(j - o0, (int)((uint32_t)a.wsel[t] * wk_bytes + slice_bytes));
      }
    }
    if (t == 0) s_nk = sk_steps > 0 ? o1 - o0 + 1 : 0;
    if (t < TM) s_orow[t] = (row0 + t < a.n_rows) ? (a.perm ? a.perm[row0 + t] : (int32_t)(row0 + t)) : -1;
    __syncthreads();
    const int cnt = sk_steps > 0 ? o1 - o0 + 1 : 0;
    {  // the piece's table: thread t has row t mod 128 of the offsets t / 128, + 2, + 4, ...  Four table entries are requested at
       // a time, from clamped (offset, row) indices, and the out-of-range ones dropped afterwards -- round 6: as `row < n ?
       // nbr[...] : -1` in a one-entry loop this was a load, s_waitcnt vmcnt(0), ds_write per entry (hipcc -S): up to 14
       // dependent L2 round trips in front of every tile piece of the level-1 launches.
      const int rr = t & (TM - 1);
      const int64_t row = row0 + rr;
      const bool row_ok = row < a.n_rows;
      const int64_t row_c = row_ok ? row : a.n_rows - 1;
      for (int q0 = t >> 7; q0 < cnt; q0 += 8) {
        int32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = a.nbr[(int64_t)s_kabs[min(q0 + 2 * u, cnt - 1)] * a.n_rows + row_c];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (q0 + 2 * u < cnt) s_off[q0 + 2 * u][rr] = (row_ok && v[u] >= 0) ? (uint32_t)v[u] * ld_bytes : kAbsent;
      }
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
          if (t == 0) s_kmeta[nk] = make_int2(q, (int)((uint32_t)a.wsel[kbeg + q] * wk_bytes + slice_bytes));
          ++nk;
        }
      }
      if (t == 0) s_nk = nk;
    }
  }
  __syncthreads();
  const int nk = s_nk;
  const int nsteps = SK ? sk_steps : nk * nch;

  f32x4 acc[2][CTN];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int ct = 0; ct < CTN; ++ct) acc[g][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this thread's pieces of a weight block (linear copy: piece p of the block -> s_b[buf][p]); DMA: per wave, 64 pieces
  // = 1 KiB per instruction, wave w takes the wave-pieces w, w + 4, ...
  constexpr int BRR = DMA ? 1 : BR;
  uint32_t bvo[BRR];
#pragma unroll
  for (int q = 0; q < BRR; ++q) bvo[q] = (t + q * 256 < PIECES) ? (uint32_t)(t + q * 256) * 16u : kAbsent;
  u32x4 breg[BRR];
  auto store_b = [&](int buf) {
    if constexpr (!DMA) {
#pragma unroll
      for (int q = 0; q < BR; ++q)
        if ((q + 1) * 256 <= PIECES || t + q * 256 < PIECES) s_b[buf][t + q * 256] = breg[q];
    }
  };
  constexpr int WPIECES = PIECES / 64, WR = (WPIECES + 3) / 4;  // wave-pieces of a block, rounds of a wave
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_b = [&](int buf, uint32_t soff) {
    if constexpr (DMA) {
#pragma unroll
      for (int r = 0; r < WR; ++r) {
        const int wp = wave_u + 4 * r;
        if (wp < WPIECES)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)&s_b[buf][wp * 64], 16,
                                                   (uint32_t)lane * 16u,
                                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(soff + (uint32_t)wp * 1024u)), 0, 0);
      }
    }
  };
  // load side: the (offset, chunk) whose operands are being requested -- one step ahead of the MFMAs (spconv16p_kernel)
  v4f a0[2][2], a1[2][2];
  int va0 = 0, va1 = 0;
  int lj = 0, lc = sk_c0 - 1;
  int2 l_meta = make_int2(0, 0);
  uint32_t l_voff[2] = {kAbsent, kAbsent};
  const uint32_t off_lane = (uint32_t)(wave * 32 + i) * 4;
  auto stage_meta = [&]() {
    ++lc;
    if (lc == nch) {
      lc = 0;
      ++lj;
    }
    lc = __builtin_amdgcn_readfirstlane(lc);
    lj = __builtin_amdgcn_readfirstlane(lj);
    l_meta = s_kmeta[min(lj, KSLOTS - 1)];
  };
  auto stage_b = [&](bool live, int nbuf) {  // nbuf: the LDS buffer the block is for (DMA)
    const int krow = __builtin_amdgcn_readfirstlane(l_meta.x) & 31;
    const uint32_t wofs = (uint32_t)__builtin_amdgcn_readfirstlane(l_meta.y);
    const char* row = reinterpret_cast<const char*>(&s_off[0][0]) + min(krow, KSLOTS - 1) * (TM * 4) + off_lane;
    l_voff[0] = *reinterpret_cast<const uint32_t*>(row);
    l_voff[1] = *reinterpret_cast<const uint32_t*>(row + 64);
    const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wofs + (uint32_t)lc * chunk_bytes));
    if constexpr (DMA) {
#if !defined(PCMI_X3_DIAG_NO_DMA)  // timing diagnostic (wrong results): the weight blocks are never staged
      if (live) dma_b(nbuf, soff);
#endif
    } else {
#pragma unroll
      for (int q = 0; q < BR; ++q)
        breg[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, live ? bvo[q] : kAbsent, live ? soff : 0u, 0));
    }
  };
  auto stage_a = [&](v4f (&dst)[2][2], bool live) -> int {
    const uint32_t v0 = live ? l_voff[0] : kAbsent, v1 = live ? l_voff[1] : kAbsent;
    const int va = (__any(v0 != kAbsent) ? 1 : 0) | (__any(v1 != kAbsent) ? 2 : 0);
    const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(lc * (kKC * 4));
#if defined(PCMI_X3_DIAG_NO_GATHER)  // timing diagnostic (wrong results): every gather out of range = no memory traffic
    const uint32_t o0 = kAbsent, o1 = kAbsent;
#else
    const uint32_t o0 = v0 + 16 * kk, o1 = v1 + 16 * kk;  // absent stays out of range
#endif
    dst[0][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, o0, soff, 0));
    dst[1][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, o1, soff, 0));
    dst[0][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, o0 + 64, soff, 0));
    dst[1][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, o1 + 64, soff, 0));
    return va;
  };

  if (nsteps > 0) {
    stage_meta();
    stage_b(true, 0);
    va0 = stage_a(a0, true);
    store_b(0);
    __syncthreads();  // (DMA: the barrier's fence waits for the block to have landed)
    auto do_step = [&](int step, v4f (&cur)[2][2], int va_cur, v4f (&nxt)[2][2], int& va_nxt) {
      PCMI_X3C_PHASE(0);  // everything between two steps (tile prologue / epilogue, table build)
      const bool more = step + 1 < nsteps;
      if constexpr (SK) {  // issue priority falls with the workgroup's progress through its share (spconv16p_kernel)
        const int qtr = __builtin_amdgcn_readfirstlane(((sk_done + step) * 4) / max(sk_total, 1));
        if (qtr != prio_qtr) {
          prio_qtr = qtr;
          if (qtr <= 0) __builtin_amdgcn_s_setprio(3);
          else if (qtr == 1) __builtin_amdgcn_s_setprio(2);
          else if (qtr == 2) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
      }
      const bool g0 = (va_cur & 1) != 0, g1 = (va_cur & 2) != 0;
      // the gathered rows of this step as three bf16 terms
      u32x4 ah[2], am[2], al[2];
#if defined(PCMI_X3_DIAG_NO_SPLIT)  // timing diagnostic (wrong results): no operand split, the raw bits stand in for the terms
      ah[0] = __builtin_bit_cast(u32x4, cur[0][0]); am[0] = __builtin_bit_cast(u32x4, cur[0][1]); al[0] = ah[0];
      ah[1] = __builtin_bit_cast(u32x4, cur[1][0]); am[1] = __builtin_bit_cast(u32x4, cur[1][1]); al[1] = ah[1];
#else
      if (g0) split3(cur[0][0], cur[0][1], ah[0], am[0], al[0]);
      if (g1) split3(cur[1][0], cur[1][1], ah[1], am[1], al[1]);
#endif
      PCMI_X3C_PHASE(1);  // priority + operand split
      // B fragments of column tile ct: piece (term, n = 16 ct + i, kk); a two-tile register ring, read one tile ahead
      const u32x4* sb = &s_b[step & 1][kk * 16 + i];  // x3_piece(16 ct + i, kk) = 64 ct + this
      u32x4 bh[2], bm[2], bl[2];
#if defined(PCMI_X3_DIAG_NO_BFRAG)  // timing diagnostic (wrong results): no fragment reads from LDS, a register stands in
      bh[0] = bm[0] = bl[0] = bh[1] = bm[1] = bl[1] = ah[0];
      (void)sb;
#else
      if (va_cur) {
        bh[0] = sb[0 * NS * 4];
        bm[0] = sb[1 * NS * 4];
        bl[0] = sb[2 * NS * 4];
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < CTN; ++ct) {
        const int rb = ct & 1;
#if !defined(PCMI_X3_DIAG_NO_BFRAG)
        if (va_cur && ct + 1 < CTN) {
          bh[rb ^ 1] = sb[(0 * NS + 16 * (ct + 1)) * 4];
          bm[rb ^ 1] = sb[(1 * NS + 16 * (ct + 1)) * 4];
          bl[rb ^ 1] = sb[(2 * NS + 16 * (ct + 1)) * 4];
        }
#endif
        // six products per row group, the small ones first, BACK TO BACK on the group's accumulator, one group after the other
        // (round 3 alternated the two groups' accumulators: the same sums in the same order per accumulator, 0.6 % slower
        // in the step -- profiles/r04e_kernel_component_removal.txt); one wave-uniform branch per column tile, not per MFMA
#if defined(PCMI_X3_DIAG_NO_MFMA)  // timing diagnostic (wrong results): the operands are kept alive, the products are not issued
#define PCMI_X3_MFMA(G, AT, BT) asm volatile("" ::"v"(AT[G]), "v"(BT[rb]))
#else
#define PCMI_X3_MFMA(G, AT, BT)                                                                                             \
  acc[G][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AT[G]), __builtin_bit_cast(bf16x8, BT[rb]), \
                                                       acc[G][ct], 0, 0, 0)
#endif
#define PCMI_X3_SIX(G)      \
  PCMI_X3_MFMA(G, al, bh);  \
  PCMI_X3_MFMA(G, ah, bl);  \
  PCMI_X3_MFMA(G, am, bm);  \
  PCMI_X3_MFMA(G, am, bh);  \
  PCMI_X3_MFMA(G, ah, bm);  \
  PCMI_X3_MFMA(G, ah, bh)
        if (g0 && g1) {
          PCMI_X3_SIX(0);
          PCMI_X3_SIX(1);
        } else if (g0) {
          PCMI_X3_SIX(0);
        } else if (g1) {
          PCMI_X3_SIX(1);
        }
#undef PCMI_X3_SIX
#undef PCMI_X3_MFMA
        // the next step's operands, requested in the shadow of this step's MFMAs
        if (ct == 0) stage_meta();
        if (ct == 1) stage_b(more, (step + 1) & 1);
        if (ct == 2) va_nxt = stage_a(nxt, more);
        __builtin_amdgcn_sched_barrier(0);
      }
      PCMI_X3C_PHASE(2);  // fragment reads, MFMAs, next step's requests
#if defined(PCMI_X3_DIAG_STAMP)
      if (va_cur) ph_sum[4] += 1; else ph_sum[5] += 1;
#endif
      if (more) store_b((step + 1) & 1);
#if defined(PCMI_X3_DIAG_NO_BARRIER)  // timing diagnostic (racy): the waves of a workgroup never meet
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#else
      __syncthreads();
#endif
      PCMI_X3C_PHASE(3);  // barrier
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
#if defined(PCMI_X3_DIAG_STAMP)
  if (blockIdx.x == 17 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    for (int e = 0; e < 8; ++e) g_x3c_phase[e] = ph_sum[e];
#endif
}

template <bool SK, bool DMA>
static int launch_x3(int NT, const ConvArgs& a, dim3 grid, hipStream_t st) {
  switch (NT) {
    case 2: spconv16x_kernel<2, SK, DMA><<<grid, 256, 0, st>>>(a); break;
    case 3: spconv16x_kernel<3, SK, DMA><<<grid, 256, 0, st>>>(a); break;
    case 4: spconv16x_kernel<4, SK, DMA><<<grid, 256, 0, st>>>(a); break;
    default: set_error("spconv x3: bad NT %d", NT); return PCMI_ERR_INVALID;
  }
  PCMI_LAUNCH_CHECK();
#if defined(PCMI_X3_DIAG_STAMP)
  {
    unsigned long long h[8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_x3c_phase), sizeof(h));
    const double ns = (double)(h[4] + h[5]);
    if (ns > 0)
      fprintf(stderr, "x3c phases: NT %d sk %d C %d rows %lld grid %u,%u,%u | steps %.0f (%.0f absent) | per step: split %.0f | reads + MFMAs + requests %.0f | barrier %.0f | between steps %.0f\n",
              NT, SK ? 1 : 0, a.C, (long long)a.n_rows, grid.x, grid.y, grid.z, ns, (double)h[5], h[1] / ns, h[2] / ns, h[3] / ns, h[0] / ns);
  }
#endif
  return PCMI_OK;
}

size_t x3_pack_bytes(int K, int C, int N) { return align_up((size_t)std::max(K, 1) * C * N * 6, 256); }

int x3_pack_weights(const ConvArgs& a, int NT, void* out, hipStream_t st) {
  PCMI_REQUIRE(out && ((uintptr_t)out % 16 == 0) && a.C % kKC == 0 && a.N % (32 * NT) == 0, PCMI_ERR_INVALID,
               "spconv x3: bad pack arguments (%d -> %d, NT %d)", a.C, a.N, NT);
  // every weight slice a launch can select: wsel maps offsets to slices 0 .. K-1
  const int64_t total = (int64_t)a.K * (a.C / kKC) * a.N * 4;
  x3_pack_kernel<<<dim3((unsigned)ceil_div(total, 256)), 256, 0, st>>>(a.w, a.w_kstride, a.w_sc, a.w_sn, a.K, a.C, a.N, 32 * NT,
                                                                      reinterpret_cast<u32x4*>(out));
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int x3_pack_many(const X3PackJob* jobs_dev, int n_jobs, int64_t total_items, hipStream_t st, int64_t first_item) {
  if (n_jobs <= 0 || total_items <= first_item) return PCMI_OK;
  x3_pack_many_kernel<<<dim3((unsigned)ceil_div(total_items - first_item, 256)), 256, 0, st>>>(jobs_dev, n_jobs, total_items, first_item);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

static thread_local const X3Prepacked* t_prepacked = nullptr;
static thread_local int t_n_prepacked = 0;
void x3_set_prepacked(const X3Prepacked* table, int n) {
  t_prepacked = table;
  t_n_prepacked = table ? n : 0;
}
const void* x3_find_prepacked(const float* w, bool transposed, int NT) {
  for (int i = 0; i < t_n_prepacked; ++i)
    if (t_prepacked[i].w == w && (t_prepacked[i].transposed != 0) == transposed && t_prepacked[i].NT == NT) return t_prepacked[i].pack;
  return nullptr;
}

int x3_launch(int NT, bool sk, const ConvArgs& a, dim3 grid, hipStream_t st) {
  PCMI_REQUIRE(a.wpack && NT >= 2 && NT <= 4, PCMI_ERR_INVALID, "spconv x3: needs packed weights and NT in 2..4 (NT %d)", NT);
  // (weight blocks go global -> LDS directly; the form through staging registers -- DMA = false, 185 instead of 166
  //  VGPRs at NT = 3, 0.404 against 0.343 ms on the level-1 launch -- is kept as a template flag only)
  return sk ? launch_x3<true, true>(NT, a, grid, st) : launch_x3<false, true>(NT, a, grid, st);
}

}  // namespace pcmi
