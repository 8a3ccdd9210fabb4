// Sparse convolution backward-weight on the fp32 matrix cores:
//     gW[k][c][n] = sum over the pairs p of offset k of  X[i_p][c] * G[j_p][n]
// i.e. per offset a [cin x M_k] @ [M_k x cout] GEMM whose contraction runs over the pairs.
//
// v_mfma_f32_32x32x2_f32 contracts two pairs per instruction: lane l = (i = l&31, h = l>>5)
// supplies A[i][h] = X[row_in(p+h)][c(i)] and B[h][j=i] = G[row_out(p+h)][n(j)].  Both operands
// are read STRAIGHT from global memory, fully coalesced: a half-wave reads CT (resp. NT)
// consecutive floats per lane from ONE feature row, so 32 lanes x CT x 4 B is one contiguous
// run of the row; the CT floats of a lane feed CT different 32x32 output tiles (tile t covers
// the channels c0 + CT*i + t), so one vector load per operand feeds CT*NT MFMAs.  No LDS in
// the main loop.  The 4 waves of a workgroup take interleaved 64-pair groups of the chunk, are
// summed through LDS, and the workgroup writes one [32*CT x 32*NT] slab; a second kernel sums
// the slabs of each offset in fixed order (deterministic, no float atomics).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "internal.h"

namespace pcmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
  const float* x;   // [*, x_ld]  rows indexed by src_x
  int64_t x_ld;
  const float* g;   // [*, g_ld]  rows indexed by src_g
  int64_t g_ld;
  const int32_t* idx_x;  // [M] or nullptr (identity)
  const int32_t* idx_g;  // [M] or nullptr (identity)
  const int64_t* offs;   // [K+1] device (nullptr: single group [0, M))
  int64_t M;
  int K;
  int cin, cout;
  int chunk;        // pairs per workgroup (multiple of 128)
  int mpk;          // maps: chunk slots per offset = ceil(bound of the pairs of one offset / chunk); workgroup b takes
                    // chunk b % mpk of offset b / mpk and leaves at once if that offset has fewer chunks -- the launch is
                    // sized without the host knowing the per-offset pair counts (they are read from `offs` on the device)
  float* slabs;     // [n_chunks][cin][cout]  (maps: slot k * mpk + j)
  // how the chunks of an offset become gW[k] (chosen on the host from the per-offset chunk counts):
  //   kSlabs  : every workgroup writes its slab, wgrad_reduce_kernel sums them (many chunks per offset: level 1)
  //   kDirect : no offset has more than one chunk -> the workgroup stores / accumulates straight into gW (levels 4+)
  //   kArrive : a few chunks per offset -> slabs + the last-arriving workgroup of the (offset, tile) sums them in
  //             chunk order (deterministic) and writes gW; no second launch
  int mode;
  float* gw;        // [K][cin][cout]
  int accumulate;
  unsigned* counters;  // kArrive: one per (offset, tile-y, tile-z), zero on entry and on exit
};
enum { kSlabs = 0, kDirect = 1, kArrive = 2 };
constexpr int kWgradGroupMax = 16;  // layers per grouped launch (WgradGroup travels in the kernel arguments: 16 x 128 B)

template <int V>
struct VecLoad;
template <>
struct VecLoad<1> {
  static __device__ inline void ld(const float* p, float* v) { v[0] = *p; }
};
template <>
struct VecLoad<2> {
  static __device__ inline void ld(const float* p, float* v) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x;
    v[1] = t.y;
  }
};
typedef float v3f_unaligned __attribute__((ext_vector_type(3), aligned(4)));
template <>
struct VecLoad<3> {
  static __device__ inline void ld(const float* p, float* v) {  // one global_load_dwordx3 (dword aligned)
    const v3f_unaligned t = *reinterpret_cast<const v3f_unaligned*>(p);
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
  }
};
template <>
struct VecLoad<4> {
  static __device__ inline void ld(const float* p, float* v) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
  }
};

// V consecutive floats through a raw buffer load: an offset >= 2^31 (kAbsentRow) is out of range and returns zeros.
// (The results are taken over with `auto` + memcpy: assigning the b64 / b96 builtins to an int ext-vector compiled to
// a ONE-dword load with the first element broadcast.)
constexpr uint32_t kAbsentRow = 0x80000000u;
template <int V>
struct BufLoad;
template <>
struct BufLoad<1> {
  static __device__ inline void ld(__amdgpu_buffer_rsrc_t r, uint32_t off, float* v) {
    v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
  }
};
template <>
struct BufLoad<2> {
  static __device__ inline void ld(__amdgpu_buffer_rsrc_t r, uint32_t off, float* v) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    static_assert(sizeof(t) == 8, "b64");
    __builtin_memcpy(v, &t, 8);
  }
};
template <>
struct BufLoad<3> {
  static __device__ inline void ld(__amdgpu_buffer_rsrc_t r, uint32_t off, float* v) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b96(r, off, 0, 0);
    static_assert(sizeof(t) >= 12, "b96");
    __builtin_memcpy(v, &t, 12);
  }
};
template <>
struct BufLoad<4> {
  static __device__ inline void ld(__amdgpu_buffer_rsrc_t r, uint32_t off, float* v) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    static_assert(sizeof(t) == 16, "b128");
    __builtin_memcpy(v, &t, 16);
  }
};

// workgroup c of the launch -> its chunk: desc = {offset k (-1: nothing to do), first pair, end, first slab of that
// offset, number of chunks of that offset}.  Dense (offs == nullptr): one group of M pairs, chunk c.
__device__ inline void locate_chunk(const int64_t* offs, int mpk, int64_t M, int chunk, int64_t c, int64_t* desc) {
  if (!offs) {
    desc[0] = c * chunk < M ? 0 : -1;
    desc[1] = c * chunk;
    desc[2] = min(desc[1] + (int64_t)chunk, M);
    desc[3] = 0;
    desc[4] = (M + chunk - 1) / chunk;
    return;
  }
  const int64_t k = c / mpk, j = c - k * mpk;
  const int64_t b = offs[k], e = offs[k + 1];
  const int64_t pb = b + j * chunk;
  desc[0] = pb < e ? k : -1;
  desc[1] = pb;
  desc[2] = min(pb + (int64_t)chunk, e);
  desc[3] = k * mpk;
  desc[4] = (e - b + chunk - 1) / chunk;
}


// locate_chunk by one wave: lane k reads the bounds of offset k, the chunk counts are scanned across the lanes -- one
// memory latency instead of up to K dependent ones (measured: 6 us of a 25 us launch at the coarse levels)
__device__ inline void locate_chunk_wave(const int64_t* offs, int K, int64_t M, int chunk, int64_t c, int lane,
                                         int64_t* desc /* k, pb, pe, first_chunk, n_chunks */) {
  if (!offs) {
    if (lane == 0) {
      desc[0] = 0;
      desc[1] = c * chunk;
      desc[2] = min(desc[1] + (int64_t)chunk, M);
      desc[3] = 0;
      desc[4] = (M + chunk - 1) / chunk;
    }
    return;
  }
  const int64_t b = lane <= K ? offs[lane] : 0;
  const int64_t e = __shfl_down(b, 1, 64);
  const int64_t nc = lane < K ? (e - b + chunk - 1) / chunk : 0;
  int64_t incl = nc;  // inclusive scan over the lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  const unsigned long long hit = __ballot(lane < K && c < incl);
  if (hit == 0) {
    if (lane == 0) {
      desc[0] = -1;
      desc[1] = desc[2] = desc[3] = desc[4] = 0;
    }
    return;
  }
  const int k = __ffsll((long long)hit) - 1;
  if (lane == k) {
    const int64_t first = incl - nc, j = c - first;
    desc[0] = k;
    desc[1] = b + j * chunk;
    desc[2] = min(desc[1] + (int64_t)chunk, e);
    desc[3] = first;
    desc[4] = nc;
  }
}

// One workgroup of the pair-list kernel: chunk `bx` of the launch, channel tile (by, bz) of a (gy x gz) tiling.
// (A function of its own since round 5: wgrad_mfma_group_kernel runs it for the workgroups of SEVERAL layers.)
template <int CT, int NT, bool IDX, bool BUF>
__device__ __forceinline__ void wgrad_mfma_body(const WgradArgs& a, int bx, int by, int bz, int gy, int gz) {
  __shared__ float s_red[32 * CT][32 * NT + 1];
  __shared__ int64_t s_desc[6];
  __shared__ unsigned s_last;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int c0 = by * 32 * CT, n0 = bz * 32 * NT;
  if (t == 0) {
    locate_chunk(a.offs, a.mpk, a.M, a.chunk, bx, s_desc);
    s_desc[5] = bx;  // slab of this chunk
  }
  __syncthreads();
  if (s_desc[0] < 0) return;
  const int64_t pb = s_desc[1], pe = s_desc[2];

  f32x16 acc[CT][NT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[ct][nt][j] = 0.f;

  if constexpr (BUF) {
    // Both operands through raw buffer loads with 32-bit byte offsets: the offset of a pair's row is computed once per
    // 64-pair group (one multiply per lane), a pair past the end of the chunk gets an out-of-range offset and reads
    // zeros -- the ragged last group of a wave runs through the same D-deep request ring as the full ones.  (The
    // separate ragged loop below requested, waited and multiplied step by step: at the coarse levels, where an offset
    // has fewer pairs than one group per wave, that loop WAS the kernel -- 25-33 us for 0.6 GFLOP.)
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, 0x7FFFFFFF, 0x00020000);
    const uint32_t xlane = (uint32_t)(c0 + CT * i) * 4, glane = (uint32_t)(n0 + NT * i) * 4;
    const uint32_t xld = (uint32_t)(a.x_ld * 4), gld = (uint32_t)(a.g_ld * 4);
    constexpr int D = (CT * NT >= 6) ? 4 : 8;
    float av[D][CT], bv[D][NT];
    auto load_off = [&](int64_t g0, uint32_t& ox, uint32_t& og) {
      const int64_t p = g0 + lane;
      ox = og = kAbsentRow;
      if (p < pe) {
        ox = (uint32_t)(IDX ? a.idx_x[p] : (int32_t)p) * xld;
        og = (uint32_t)(IDX ? a.idx_g[p] : (int32_t)p) * gld;
      }
    };
#define PCMI_WGRAD_BLOAD(J, OX, OG)                                                             \
  {                                                                                             \
    const uint32_t x0 = __builtin_amdgcn_readlane(OX, 2 * (J)), x1 = __builtin_amdgcn_readlane(OX, 2 * (J) + 1); \
    const uint32_t y0 = __builtin_amdgcn_readlane(OG, 2 * (J)), y1 = __builtin_amdgcn_readlane(OG, 2 * (J) + 1); \
    BufLoad<CT>::ld(xr, (h ? x1 : x0) + xlane, av[(J) % D]);                                    \
    BufLoad<NT>::ld(gr, (h ? y1 : y0) + glane, bv[(J) % D]);                                    \
  }
    int64_t g0 = pb + (int64_t)wave * 64;
    uint32_t ox = 0, og = 0, oxn = 0, ogn = 0;
    bool primed = false;
    for (; g0 < pe; g0 += 256) {
      const bool next = g0 + 256 < pe;
      if (!primed) {
        load_off(g0, ox, og);
#pragma unroll
        for (int j = 0; j < D; ++j) PCMI_WGRAD_BLOAD(j, ox, og)
      }
      if (next) load_off(g0 + 256, oxn, ogn);
#pragma unroll
      for (int s = 0; s < 32; ++s) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[ct][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % D][ct], bv[s % D][nt], acc[ct][nt], 0, 0, 0);
        if (s + D < 32) {
          PCMI_WGRAD_BLOAD(s + D, ox, og)
        } else if (next) {
          PCMI_WGRAD_BLOAD(s + D - 32, oxn, ogn)
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the requests D steps ahead of their use
      }
      ox = oxn;
      og = ogn;
      primed = next;
    }
#undef PCMI_WGRAD_BLOAD
  } else {
    const float* xcol = a.x + c0 + CT * i;
    const float* gcol = a.g + n0 + NT * i;
    // Each wave walks 64-pair groups (group index = wave, wave+4, ...); one group is 32 MFMA contraction steps of two
    // pairs.  The operand rows of step s+D are requested while step s multiplies (register ring of D steps, running on
    // into the next group): written as load -> use per step, hipcc emitted global_load -> s_waitcnt vmcnt(0) -> MFMAs,
    // the whole memory latency in front of every CT*NT MFMAs (SQ_VALU_MFMA_BUSY_CYCLES: 38 %).
    constexpr int D = (CT * NT >= 6) ? 4 : 8;  // steps in flight: >= ~2000 cycles of matrix work (32 % D == 0)
    float av[D][CT], bv[D][NT];
    auto load_idx = [&](int64_t g0, int32_t& rx, int32_t& rg) {
      const int64_t p = g0 + lane;
      rx = IDX ? a.idx_x[p] : (int32_t)p;  // IDX is a template flag: a run-time select put an s_waitcnt vmcnt(0) here
      rg = IDX ? a.idx_g[p] : (int32_t)p;
    };
    // operands of contraction step j of a FULL group: lane (i, h) takes pair 2j+h
  #define PCMI_WGRAD_LOAD(J, RX, RG)                                                            \
    {                                                                                           \
      const int32_t ix0 = __builtin_amdgcn_readlane(RX, 2 * (J)), ix1 = __builtin_amdgcn_readlane(RX, 2 * (J) + 1); \
      const int32_t ig0 = __builtin_amdgcn_readlane(RG, 2 * (J)), ig1 = __builtin_amdgcn_readlane(RG, 2 * (J) + 1); \
      VecLoad<CT>::ld(xcol + (int64_t)(h ? ix1 : ix0) * a.x_ld, av[(J) % D]);                   \
      VecLoad<NT>::ld(gcol + (int64_t)(h ? ig1 : ig0) * a.g_ld, bv[(J) % D]);                   \
    }
    int64_t g0 = pb + (int64_t)wave * 64;
    int32_t rx = 0, rg = 0, rxn = 0, rgn = 0;
    bool primed = false;  // ring holds steps 0..D-1 of the group at g0
    for (; g0 + 64 <= pe; g0 += 256) {
      const bool next_full = g0 + 256 + 64 <= pe;
      if (!primed) {
        load_idx(g0, rx, rg);
  #pragma unroll
        for (int j = 0; j < D; ++j) PCMI_WGRAD_LOAD(j, rx, rg)
      }
      if (next_full) load_idx(g0 + 256, rxn, rgn);
  #pragma unroll
      for (int s = 0; s < 32; ++s) {
  #pragma unroll
        for (int ct = 0; ct < CT; ++ct)
  #pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[ct][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % D][ct], bv[s % D][nt], acc[ct][nt], 0, 0, 0);
        if (s + D < 32) {
          PCMI_WGRAD_LOAD(s + D, rx, rg)
        } else if (next_full) {
          PCMI_WGRAD_LOAD(s + D - 32, rxn, rgn)
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the requests D steps ahead of their use
      }
      rx = rxn;
      rg = rgn;
      primed = next_full;
    }
  #undef PCMI_WGRAD_LOAD
    // ragged last group of this wave
    if (g0 < pe) {
      const int64_t p = g0 + lane;
      int32_t tx = -1, tg = -1;
      if (p < pe) {
        tx = IDX ? a.idx_x[p] : (int32_t)p;
        tg = IDX ? a.idx_g[p] : (int32_t)p;
      }
      const int npairs = (int)(pe - g0);
      for (int s = 0; 2 * s < npairs; ++s) {
        const int32_t ix = __shfl(tx, 2 * s + h, 64);
        const int32_t ig = __shfl(tg, 2 * s + h, 64);
        float ta[CT], tb[NT];
        if (ix >= 0) {
          VecLoad<CT>::ld(xcol + (int64_t)ix * a.x_ld, ta);
          VecLoad<NT>::ld(gcol + (int64_t)ig * a.g_ld, tb);
        } else {
  #pragma unroll
          for (int ct = 0; ct < CT; ++ct) ta[ct] = 0.f;
  #pragma unroll
          for (int nt = 0; nt < NT; ++nt) tb[nt] = 0.f;
        }
  #pragma unroll
        for (int ct = 0; ct < CT; ++ct)
  #pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[ct][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[ct], tb[nt], acc[ct][nt], 0, 0, 0);
      }
    }

  }
  // reduce the 4 waves one after the other through ONE LDS tile, wave 0 writes the slab
  for (int src = 1; src < 4; ++src) {
    if (wave == src) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
            s_red[ct * 32 + row][nt * 32 + i] = acc[ct][nt][j];
          }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
            acc[ct][nt][j] += s_red[ct * 32 + row][nt * 32 + i];
          }
    }
    __syncthreads();
  }
  // ---- write-out: the summed tile goes through LDS so that all 256 threads store (or accumulate) it, coalesced
  // along cout, with every read of a read-modify-write issued before the first store ---------------------------
  const int64_t per_k = (int64_t)a.cin * a.cout;
  const int k_off = (int)s_desc[0];
  constexpr int TW = 32 * NT, TE = 32 * CT * TW, EPT = TE / 256;  // tile width, elements, elements per thread
  if (wave == 0) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          // tile (ct, nt): row = lane index of the A role (cin), col = lane index i of the B role (cout)
          const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
          s_red[CT * row + ct][NT * i + nt] = acc[ct][nt][j];
        }
  }
  __syncthreads();
  const bool direct = a.mode == kDirect;
  float* base = direct ? a.gw + (int64_t)k_off * per_k : a.slabs + s_desc[5] * per_k;
  float v[EPT];
  int64_t el[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int e = t + 256 * j, r = e / TW, col = e - r * TW;
    el[j] = (int64_t)(c0 + r) * a.cout + n0 + col;
    v[j] = s_red[r][col];
  }
  if (direct && a.accumulate) {
    float old[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) old[j] = base[el[j]];
#pragma unroll
    for (int j = 0; j < EPT; ++j) v[j] += old[j];
  }
#pragma unroll
  for (int j = 0; j < EPT; ++j) base[el[j]] = v[j];
  if (a.mode != kArrive) return;
  // ---- the last workgroup of this (offset, tile) to arrive sums the slabs in chunk order -------------------------
  const int64_t first = s_desc[3], count = s_desc[4];
  unsigned* counter = a.counters + ((int64_t)k_off * gy + by) * gz + bz;
  if (!arrive_last(counter, (unsigned)count, &s_last)) return;
  float sum[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) sum[j] = 0.f;
  for (int64_t q = 0; q < count; ++q) {  // chunk order (deterministic); EPT independent loads in flight per chunk
    const float* sl = a.slabs + (first + q) * per_k;
#pragma unroll
    for (int j = 0; j < EPT; ++j) sum[j] += sl[el[j]];
  }
  float* dst = a.gw + (int64_t)k_off * per_k;
  if (a.accumulate) {
    float old[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) old[j] = dst[el[j]];
#pragma unroll
    for (int j = 0; j < EPT; ++j) sum[j] += old[j];
  }
#pragma unroll
  for (int j = 0; j < EPT; ++j) dst[el[j]] = sum[j];
}

template <int CT, int NT, bool IDX, bool BUF>
// min 2 waves/SIMD keeps the 9 accumulator tiles in VGPRs (198 registers); unbounded, hipcc used 190 + 144 AGPRs = 1 wave/SIMD
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WgradArgs a) {
  wgrad_mfma_body<CT, NT, IDX, BUF>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.y, (int)gridDim.z);
}

// The weight gradients of SEVERAL layers in one launch (round 5).  The coarse levels of the U-Net (strides 8 and 16: 800-
// 2700 rows per pair tensor, 128-256 channels) have 24 3^3 layers whose gradients are latency-, not matrix-bound: every
// launch is a few hundred workgroups that each walk a few hundred pairs, and the 27 launches of a step took 62 us each on
// the weight-gradient stream, one after the other (1.7 ms per step, 10-18 TFLOP/s).  All layers of a level are independent once their output
// gradients exist, so the executor collects them (csrc/engine.hip) and launches one grid over all their workgroups:
// workgroup b -> job (first[j] <= b), then the layer's own (offset, tile) numbering; every offset is ONE chunk ("direct"
// mode: the workgroup adds its tile straight into the flat gradient), so the sums differ from the single launches' (which
// cut long offsets into chunks) by their order only.  The jobs travel in the kernel arguments (no table upload): <= kWgradGroupMax.
struct WgradGroup {
  int n;
  int first[kWgradGroupMax + 1];  // first[j]: first workgroup of job j; first[n]: grid size
  WgradArgs job[kWgradGroupMax];
};

template <int CT, int NT>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_group_kernel(WgradGroup g) {
  int j = 0;
  for (int q = 1; q < g.n; ++q)
    if ((int)blockIdx.x >= g.first[q]) j = q;  // (uniform)
  const WgradArgs a = g.job[j];
  const int local = (int)blockIdx.x - g.first[j];
  const int gy = a.cin / (32 * CT), gz = a.cout / (32 * NT);
  const int nchunks = a.K * a.mpk;  // maps only (a.offs != nullptr)
  const int bx = local % nchunks, r = local / nchunks;
  wgrad_mfma_body<CT, NT, true, true>(a, bx, r % gy, r / gy, gy, gz);
}

// gW[k][e] = sum over the chunks of offset k of slab[chunk][e] (fixed order -> deterministic).
// LANES == 1: one thread per output element (few chunks per offset: small levels).
// LANES == 8: 32 elements x 8 chunk lanes per workgroup, folded through LDS (many chunks per offset:
//             level 1, dense 1x1 convs), so the serial chain of dependent loads is 8x shorter.
template <int LANES>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs,
                                                           const int64_t* __restrict__ offs, int mpk, int64_t M,
                                                           int chunk, int64_t per_k /* cin*cout */,
                                                           float* __restrict__ gw, int accumulate) {
  __shared__ float s_part[LANES][33];
  const int k = blockIdx.y;
  constexpr int EPB = 256 / LANES;  // elements per block
  const int el = threadIdx.x % EPB, cl = threadIdx.x / EPB;
  const int64_t e = (int64_t)blockIdx.x * EPB + el;
  int64_t first = 0, count;
  if (!offs) {
    count = (M + chunk - 1) / chunk;
  } else {  // the slabs of offset k sit in slots k * mpk ... (see WgradArgs::mpk)
    first = (int64_t)k * mpk;
    count = (offs[k + 1] - offs[k] + chunk - 1) / chunk;
  }
  float s = 0.f;
  if (e < per_k && cl < count) {
    // chunks cl, cl + LANES, ... in that order, requested EIGHT at a time from clamped indices (a chunk past the last is
    // the last again and is not added).  Round 6: one dependent load per chunk was 85 L2 round trips per thread for the
    // head's 683 chunks -- 38 us for 8 MB (hipcc -S: load, s_waitcnt vmcnt(0), add).  Same additions in the same order.
    const float* base = slabs + first * per_k + e;
    const int64_t last = cl + ((count - 1 - cl) / LANES) * LANES;
    for (int64_t c0 = cl; c0 < count; c0 += 8 * LANES) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[min(c0 + (int64_t)u * LANES, last) * per_k];
#pragma unroll
      for (int u = 0; u < 8; ++u) s = (c0 + (int64_t)u * LANES < count) ? s + v[u] : s;
    }
  }
  if (LANES > 1) {
    s_part[cl][el] = s;
    __syncthreads();
    if (cl != 0) return;
#pragma unroll
    for (int q = 1; q < LANES; ++q) s += s_part[q][el];
  }
  if (e < per_k) {
    float* dst = gw + (int64_t)k * per_k + e;
    *dst = accumulate ? *dst + s : s;
  }
}

// tiny-channel stem (cin = 3): slab[chunk][c][n] = sum_p x[i_p][c] * g[j_p][n]
template <int CIN>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(WgradArgs a) {
  __shared__ float s_part[8][CIN][33];
  __shared__ int64_t s_desc[5];
  const int t = threadIdx.x;
  if (t == 0) locate_chunk(a.offs, a.mpk, a.M, a.chunk, blockIdx.x, s_desc);
  __syncthreads();
  if (s_desc[0] < 0) return;
  const int64_t pb = s_desc[1], pe = s_desc[2];
  const int n = t & 31, sub = t >> 5;
  float* slab = a.slabs + (int64_t)blockIdx.x * a.cin * a.cout;
  for (int nb = 0; nb < a.cout; nb += 32) {
    float acc[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) acc[c] = 0.f;
    for (int64_t p = pb + sub; p < pe; p += 8) {
      const int32_t ix = a.idx_x ? a.idx_x[p] : (int32_t)p;
      const int32_t ig = a.idx_g ? a.idx_g[p] : (int32_t)p;
      const float gv = a.g[(int64_t)ig * a.g_ld + nb + n];
      const float* xp = a.x + (int64_t)ix * a.x_ld;
#pragma unroll
      for (int c = 0; c < CIN; ++c) acc[c] = fmaf(xp[c], gv, acc[c]);
    }
#pragma unroll
    for (int c = 0; c < CIN; ++c) s_part[sub][c][n] = acc[c];
    __syncthreads();
    if (sub == 0) {
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += s_part[q][c][n];
        slab[(int64_t)c * a.cout + nb + n] = s;
      }
    }
    __syncthreads();
  }
}

// The stem's weight gradient (3 input channels) as ONE [K*3 x rows] @ [rows x 32] product on the fp32 matrix cores
// (round 5).  stem_wgrad_kernel above walks the pair list of one offset per workgroup and reads the 128-byte gradient
// row of every PAIR -- ~17 times per row, 380 MB for 22 MB of gradient rows, 165 us at 175k rows -- and this is the LAST
// weight gradient of a backward pass: the chain has finished, the optimiser waits for it.  Here a workgroup takes
// 64-row blocks: its 256 threads read the block's 27 x 64 entries of the offset-major neighbour table (64 consecutive
// rows per wave: coalesced) and gather the 3 input channels of every present neighbour into an LDS tile
// X81[row][3k + c] (absent: 0; columns 81..95 stay 0); wave w multiplies rows 16w..16w+15 -- v_mfma_f32_32x32x2_f32,
// lane (i, h) supplies A = X81[row + h][32 mt + i] from LDS and B = g[row + h][n0 + i] straight from global memory, a
// gradient row is read ONCE -- into three 32x32 accumulators (m = 3k + c padded to 96).  The four waves are added
// through LDS in wave order, the workgroup leaves one [81][cout] slab, stem_slab_reduce_kernel adds the slabs in
// workgroup order.  Deterministic; fp32 products and sums (the instruction wgrad_mfma_kernel uses).
// (A first row-stationary form on the vector ALUs -- a thread owning one output channel of every 8th row and all 81
//  sums in registers -- took 436 us: profiles/r05k_stem_weight_gradient_row_stationary_ab.txt.)
template <int CIN, int KV>
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const float* __restrict__ x, int64_t x_ld, const float* __restrict__ g,
                                                              int64_t g_ld, const int32_t* __restrict__ nbr, int64_t n_rows,
                                                              int cout, int64_t n_blocks, float* __restrict__ slabs) {
  constexpr int E = KV * CIN;          // 81 rows of the gradient slice [K][CIN] x cout
  constexpr int MT = (E + 31) / 32;    // 32-row accumulator tiles
  constexpr int LD = 32 * MT + 1;      // (odd: lanes = consecutive rows write one column conflict-free)
  constexpr int RB = 64;               // rows per block
  constexpr int KJ = (KV + 3) / 4;     // table entries per thread and block (wave w: offsets w, w + 4, ...)
  static_assert(96 * 33 <= RB * LD, "the wave reduction tile aliases the operand tile");
  __shared__ float s_a[RB][LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 31, h = lane >> 5;
  const int n0 = (int)blockIdx.y * 32;
  for (int e = t; e < RB * (LD - E); e += 256) s_a[e / (LD - E)][E + e % (LD - E)] = 0.f;
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[mt][j] = 0.f;
  for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
    const int64_t r0 = b * RB;
    float bv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int64_t row = r0 + 16 * wave + 2 * s + h;
      bv[s] = row < n_rows ? g[row * g_ld + n0 + i] : 0.f;
    }
    int32_t ix[KJ];
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int k = wave + 4 * j;
      ix[j] = (k < KV && r0 + lane < n_rows) ? nbr[(int64_t)k * n_rows + r0 + lane] : -1;
    }
    float xv[KJ][CIN];
#pragma unroll
    for (int j = 0; j < KJ; ++j)
#pragma unroll
      for (int c = 0; c < CIN; ++c) xv[j][c] = ix[j] >= 0 ? x[(int64_t)ix[j] * x_ld + c] : 0.f;
    __syncthreads();  // the previous block's products have read the tile (first block: the zero columns are written)
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int k = wave + 4 * j;
      if (k < KV) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) s_a[lane][CIN * k + c] = xv[j][c];
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float* ar = &s_a[16 * wave + 2 * s + h][i];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[32 * mt], bv[s], acc[mt], 0, 0, 0);
    }
  }
  __syncthreads();
  float (*s_red)[33] = reinterpret_cast<float (*)[33]>(&s_a[0][0]);
  for (int src = 0; src < 4; ++src) {  // wave order: the same sum on every run
    if (wave == src) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int row = 32 * mt + (j & 3) + 8 * (j >> 2) + 4 * h;  // (A role: m = 3k + c; column i: output channel)
          s_red[row][i] = src == 0 ? acc[mt][j] : s_red[row][i] + acc[mt][j];
        }
    }
    __syncthreads();
  }
  float* slab = slabs + (int64_t)blockIdx.x * E * cout;  // [E][cout]
  for (int e = t; e < E * 32; e += 256) slab[(int64_t)(e >> 5) * cout + n0 + (e & 31)] = s_red[e >> 5][e & 31];
}

// gw[e] (+)= sum over the workgroups of slab[b][e]: 8 elements x 32 slab lanes per workgroup (lane q adds slabs q, q + 32,
// ... in that order, the lanes are added in lane order: the same sum on every run).  512 slabs of 10 KB: 16 dependent adds
// per thread instead of the 64 of an 8-lane split -- this launch is the last thing the optimiser waits for.
__global__ __launch_bounds__(256) void stem_slab_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int64_t per,
                                                               float* __restrict__ gw, int accumulate) {
  __shared__ float s_p[32][9];
  const int el = threadIdx.x & 7, cl = threadIdx.x >> 3;
  const int64_t e = (int64_t)blockIdx.x * 8 + el;
  float s = 0.f;
  if (e < per)
    for (int b = cl; b < n_slabs; b += 32) s += slabs[(int64_t)b * per + e];
  s_p[cl][el] = s;
  __syncthreads();
  if (cl != 0 || e >= per) return;
#pragma unroll
  for (int q = 1; q < 32; ++q) s += s_p[q][el];
  gw[e] = accumulate ? gw[e] + s : s;
}
constexpr int kStemSlabs = 512;  // workgroups (= slabs) of stem_wgrad_mfma_kernel: two per CU

// column sums (bias gradient): two-level, deterministic.  A workgroup sums its row block with 256 / c row lanes per
// column (round 4 used one thread per column: 32 of 256 threads at c = 32, each walking its 171 rows alone -- 111 us
// for the 22 MB of the final layer's gradient) and folds the lanes through LDS in lane order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ g, int64_t g_ld,
                                                             int64_t n, int c, int rows_per_block,
                                                             float* __restrict__ part) {
  __shared__ float s_p[256];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(r0 + rows_per_block, n);
  const int cw = min(c, 256), lanes = 256 / cw;
  const int t = threadIdx.x, cl = t % cw, rl = t / cw;
  for (int c0 = 0; c0 < c; c0 += cw) {
    const int col = c0 + cl;
    float s = 0.f;
    if (rl < lanes && col < c && r0 + rl < r1) {
      // rows r0 + rl, + lanes, ... in that order, eight requests at a time from clamped rows (round 6: one dependent load per
      // row was 21 memory round trips per thread; a row past the block is the last one again and is not added)
      const int64_t last = r0 + rl + ((r1 - 1 - r0 - rl) / lanes) * lanes;
      for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)8 * lanes) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = g[min(rb + (int64_t)u * lanes, last) * g_ld + col];
#pragma unroll
        for (int u = 0; u < 8; ++u) s = (rb + (int64_t)u * lanes < r1) ? s + v[u] : s;
      }
    }
    s_p[t] = s;
    __syncthreads();
    if (rl == 0 && col < c) {
      for (int q = 1; q < lanes; ++q) s += s_p[q * cw + cl];
      part[(int64_t)blockIdx.x * c + col] = s;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nblocks, int c,
                                                           float* __restrict__ out, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per column
  if (col >= c) return;
  float s = 0.f;
  for (int b = lane; b < nblocks; b += 64) s += part[(int64_t)b * c + col];
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) s += __shfl_xor(s, d, 64);
  if (lane == 0) out[col] = accumulate ? out[col] + s : s;
}

// buf: both operands are < 2 GiB (32-bit byte offsets); else the 64-bit-address form
template <int CT, int NT>
static void launch_wg(const WgradArgs& a, dim3 grid, bool buf, hipStream_t st) {
  if (buf) {
    if (a.idx_x)
      wgrad_mfma_kernel<CT, NT, true, true><<<grid, 256, 0, st>>>(a);
    else
      wgrad_mfma_kernel<CT, NT, false, true><<<grid, 256, 0, st>>>(a);
  } else {
    if (a.idx_x)
      wgrad_mfma_kernel<CT, NT, true, false><<<grid, 256, 0, st>>>(a);
    else
      wgrad_mfma_kernel<CT, NT, false, false><<<grid, 256, 0, st>>>(a);
  }
}

template <int CT>
static int launch_wg_nt(int NT, const WgradArgs& a, dim3 grid, bool buf, hipStream_t st) {
  switch (NT) {
    case 1: launch_wg<CT, 1>(a, grid, buf, st); break;
    case 2: launch_wg<CT, 2>(a, grid, buf, st); break;
    case 3: launch_wg<CT, 3>(a, grid, buf, st); break;
    default: set_error("wgrad: bad NT %d", NT); return PCMI_ERR_INVALID;
  }
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

// resident workgroups per CU of one variant (cached; only steers the chunk size)
template <int CT, int NT>
static int occupancy_wg(bool idx) {
  static int cache[2] = {0, 0};
  int& v = cache[idx ? 1 : 0];
  if (v == 0) {
    int n = 0;
    hipError_t e = idx ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_mfma_kernel<CT, NT, true, true>, 256, 0)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_mfma_kernel<CT, NT, false, true>, 256, 0);
    v = (e == hipSuccess && n > 0) ? std::min(n, 8) : 2;
    (void)hipGetLastError();
  }
  return v;
}

static int wgrad_occupancy(int CT, int NT, bool idx) {
  switch (CT * 4 + NT) {
    case 1 * 4 + 1: return occupancy_wg<1, 1>(idx);
    case 1 * 4 + 2: return occupancy_wg<1, 2>(idx);
    case 1 * 4 + 3: return occupancy_wg<1, 3>(idx);
    case 2 * 4 + 1: return occupancy_wg<2, 1>(idx);
    case 2 * 4 + 2: return occupancy_wg<2, 2>(idx);
    case 2 * 4 + 3: return occupancy_wg<2, 3>(idx);
    case 3 * 4 + 1: return occupancy_wg<3, 1>(idx);
    case 3 * 4 + 2: return occupancy_wg<3, 2>(idx);
    default: return occupancy_wg<3, 3>(idx);
  }
}

static int tiles_per_wg(int tiles32) {  // tiles (of 32 channels) a workgroup covers per axis
  if (tiles32 % 3 == 0) return 3;
  if (tiles32 % 2 == 0) return 2;
  return 1;
}

// Pairs per workgroup.  Every chunk costs the same, so the launch runs in ceil(workgroups / resident slots) rounds of
// equal length, and a round lasts as long as the wave with the most 64-pair groups: ceil(chunk / 256) groups.  Pick
// the chunk with the smallest rounds x groups (ties: the larger chunk, fewer slabs).  Per-wave cycle accounting
// (scripts/wgrad_prof.py) of the rule this replaces -- "the largest chunk that fills >= 92 % of a whole number of
// rounds" -- on the level-1 96->96 gradient: 3200 pairs = 12.5 groups per wave, i.e. half the waves of every workgroup
// waited one whole group (8 % of the loop) at the reduction barrier, and 480 workgroups on 512 slots.
constexpr int kWgradMinChunk = 256, kWgradMaxChunks = 4096;
// Longest chunk (pairs).  A workgroup of the level-1 gradients runs ~80 us per 1024 pairs and is never pre-empted: while
// it holds its CU, the latency-critical kernels of the bwd-data chain (higher stream priority, but priority only orders
// the DISPATCH of new workgroups) wait for a slot.
static int wgrad_max_chunk() { return 4096; }

static int wgrad_min_chunk(int64_t M) {
  return (int)std::max<int64_t>(kWgradMinChunk, align_up((size_t)ceil_div(M, kWgradMaxChunks), 128));
}

// Pairs one offset of a map can have at most: a row takes part in at most one pair per offset on either side.
static int64_t wgrad_offset_bound(int64_t n_in, int64_t n_out) { return std::min(n_in, n_out); }

// Pairs of offset k as far as the host knows (only steers the chunk size): exact once the map's counts have arrived,
// otherwise the typical occupancy of a surface (17 of 27 neighbours) / an even split of the fine rows over the 8 children.
static int64_t wgrad_offset_len(const pcmi_kmap_t* map, int k) {
  if (map->M >= 0) return map->offs_host[k + 1] - map->offs_host[k];
  return map->stride == 1 ? map->n_out * 17 / 27 : ceil_div(map->n_in, map->K);
}

static int64_t wgrad_num_chunks(const pcmi_kmap_t* map, int64_t M, int chunk) {
  if (!map) return ceil_div(M, chunk);
  int64_t n = 0;
  for (int k = 0; k < map->K; ++k) n += ceil_div(wgrad_offset_len(map, k), chunk);
  return n;
}

// lo: smallest admissible chunk (keeps the number of chunk slots bounded, see spconv_wgrad_workspace)
static int wgrad_chunk(const pcmi_kmap_t* map, int64_t M, int lo, int64_t wgs_per_chunk, int64_t slots) {
  int64_t best_cost = -1;
  for (int c = std::max(lo, wgrad_max_chunk()); c >= lo; c -= 128) {
    const int64_t cost = ceil_div(wgrad_num_chunks(map, M, c) * wgs_per_chunk, slots) * ceil_div(c, 256);
    if (best_cost < 0 || cost < best_cost) best_cost = cost;
  }
  // the largest chunk within 4 % of the best: fewer slabs to write and to sum
  for (int c = std::max(lo, wgrad_max_chunk()); c >= lo; c -= 128) {
    const int64_t cost = ceil_div(wgrad_num_chunks(map, M, c) * wgs_per_chunk, slots) * ceil_div(c, 256);
    if (cost * 100 <= best_cost * 104) return c;
  }
  return lo;
}

// Slab slots of a launch: K * ceil(offset bound / chunk) for a map (WgradArgs::mpk), ceil(M / chunk) for the dense case;
// the smallest chunk a launch may pick keeps them under ~4096 + K.
static int wgrad_lo_chunk(int64_t n_in, int64_t n_out, int K, int64_t M) {
  return K > 1 ? wgrad_min_chunk(wgrad_offset_bound(n_in, n_out) * K) : wgrad_min_chunk(M);
}

size_t spconv_wgrad_workspace(int64_t n_in, int64_t n_out, int cin, int cout, int K, int64_t M) {
  if (M <= 0) M = std::max(n_in, n_out);
  const int lo = wgrad_lo_chunk(n_in, n_out, K, M);
  const int64_t nchunks = K > 1 ? (int64_t)K * ceil_div(wgrad_offset_bound(n_in, n_out), lo) : ceil_div(M, lo);
  const size_t pairwise = (size_t)std::max<int64_t>(nchunks, 1) * cin * cout * sizeof(float);
  // spconv_wgrad_x3.hip (K = 1: the dense 1x1 form of the same kernel, up to 128 row blocks of one slab each)
  const size_t tiled = (K == 27 && n_in == n_out) ? wgrad_x3t_workspace(n_out, cin, cout)
                                                  : ((K == 1 && n_in == n_out) ? (size_t)128 * cin * cout * sizeof(float) : 0);
  const size_t stem_slabs = cin == 3 ? (size_t)kStemSlabs * K * cin * cout * sizeof(float) : 0;  // stem_wgrad_mfma_kernel
  return std::max(std::max(pairwise, tiled), stem_slabs) + (size_t)1024 * cout * sizeof(float);
}

}  // namespace pcmi

using namespace pcmi;

namespace pcmi {
size_t spconv_fwd_bwd_workspace(int64_t n_in, int64_t n_out, int cin, int cout, int K);
}

namespace pcmi {
size_t spconv_workspace_m32(int64_t n_in, int64_t n_out, int cin, int cout, int K, int64_t M) {
  return std::max(spconv_fwd_bwd_workspace(n_in, n_out, cin, cout, K),
                  spconv_wgrad_workspace(n_in, n_out, cin, cout, K, M)) + 256;
}
}  // namespace pcmi

namespace pcmi {
// Everything a pair-list launch needs besides the launch itself: operands, chunk size, chunk slots per offset, how the
// chunks of an offset become gW (mode), the tile shape.  Shared by the single launch and the grouped one.
static int wgrad_plan(const float* in, int64_t in_ld, int64_t n_in, int cin, const float* gout, int64_t gout_ld, int64_t n_out,
                      int cout, const pcmi_kmap_t* map, int transpose, float* gweight, int accumulate, int64_t M, hipStream_t st,
                      WgradArgs* out, int* CT_out, int* NT_out, int64_t* nchunks_out) {
  const int K = map ? map->K : 1;
  WgradArgs a;
  a.x = in;
  a.x_ld = in_ld;
  a.g = gout;
  a.g_ld = gout_ld;
  a.idx_x = map ? (transpose ? map->pair_out : map->pair_in) : nullptr;
  a.idx_g = map ? (transpose ? map->pair_in : map->pair_out) : nullptr;
  a.offs = map ? map->offs : nullptr;
  a.M = M;
  a.K = K;
  a.cin = cin;
  a.cout = cout;
  const bool stem = cin < 8;
  int CT = 1, NT = 1;
  int64_t slots = 8 * (int64_t)num_cu(), wgs_per_chunk = 1;
  if (stem) {
    PCMI_REQUIRE(cin == 3 && cout % 32 == 0, PCMI_ERR_UNSUPPORTED, "spconv_bwd_weight: cin=%d only supported as the 3-channel stem", cin);
  } else {
    PCMI_REQUIRE(cin % 32 == 0 && cout % 32 == 0, PCMI_ERR_UNSUPPORTED,
                 "spconv_bwd_weight: channels (%d, %d) must be multiples of 32", cin, cout);
    PCMI_REQUIRE(in_ld % 4 == 0 && gout_ld % 4 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)gout % 16 == 0,
                 PCMI_ERR_INVALID, "spconv_bwd_weight: operands must be 16-byte aligned with ld %% 4 == 0");
    CT = tiles_per_wg(cin / 32);
    NT = tiles_per_wg(cout / 32);
    wgs_per_chunk = (int64_t)(cin / (32 * CT)) * (cout / (32 * NT));
    slots = (int64_t)wgrad_occupancy(CT, NT, a.idx_x != nullptr) * num_cu();
  }
  // The launch is sized WITHOUT the per-offset pair counts (pcmi_coords_plan_unet leaves them on the device): every
  // offset gets mpk = ceil(bound / chunk) chunk slots and a workgroup whose slot lies behind the offset's last pair
  // leaves at once (WgradArgs::mpk).  The counts, when the host has them, only steer the chunk size.
  a.chunk = wgrad_chunk(map, M, wgrad_lo_chunk(n_in, n_out, K, M), wgs_per_chunk, slots);
  a.mpk = map ? (int)ceil_div(wgrad_offset_bound(n_in, n_out), a.chunk) : 0;
  const int64_t nchunks = map ? (int64_t)K * a.mpk : ceil_div(M, a.chunk);
  const int64_t max_per_k = map ? a.mpk : nchunks;  // bound of the chunks of one offset
  constexpr int kArriveMax = 8;  // most chunks per offset the in-kernel (last-arriver) reduction takes
  a.mode = stem ? kSlabs : (max_per_k <= 1 ? kDirect : (max_per_k <= kArriveMax ? kArrive : kSlabs));
  a.gw = gweight;
  a.accumulate = accumulate;
  a.counters = nullptr;
  if (a.mode == kArrive) {
    a.counters = stream_counters(st, (size_t)K * (cin / (32 * CT)) * (cout / (32 * NT)));
    if (!a.counters) return PCMI_ERR_HIP;
  }
  *out = a;
  *CT_out = CT;
  *NT_out = NT;
  *nchunks_out = nchunks;
  return PCMI_OK;
}

// ---- grouped launches (wgrad_mfma_group_kernel) ------------------------------------------------------------------------
// PCMI_WGRAD_GROUP=0: every layer its own launch (round 4's form; A/B).  Read per call.
static bool wgrad_group_on() {
  const char* e = getenv("PCMI_WGRAD_GROUP");
  return !(e && e[0] == '0');
}

struct WgradGroupBuilder {
  WgradGroup g;
  int CT = 0, NT = 0;
};

WgradGroupBuilder* wgrad_group_create() {
  WgradGroupBuilder* b = new WgradGroupBuilder();
  b->g.n = 0;
  b->g.first[0] = 0;
  return b;
}
void wgrad_group_destroy(WgradGroupBuilder* b) { delete b; }
int wgrad_group_size(const WgradGroupBuilder* b) { return b ? b->g.n : 0; }
void wgrad_group_drop(WgradGroupBuilder* b) {
  if (b) {
    b->g.n = 0;
    b->g.first[0] = 0;
  }
}

// true: the layer's weight gradient will be enqueued by the next wgrad_group_flush (its operands must stay as they are
// until then); false: not a candidate (the caller launches it by itself -- after flushing if the order matters to it).
// Candidates: 3^3 / stride-1 table maps in "direct" mode (every offset one chunk: the coarse levels), accumulating,
// 32-multiples of channels, 32-bit addressable operands, not taken by the split-precision tile kernel, and the tile
// shape of the layers already collected.
bool wgrad_group_add(WgradGroupBuilder* b, const float* in, int64_t in_ld, int64_t n_in, int cin, const float* gout,
                     int64_t gout_ld, int64_t n_out, int cout, const pcmi_kmap_t* map, float* gweight, int accumulate,
                     hipStream_t st) {
  if (!b || !wgrad_group_on() || !map || !in || !gout || !gweight || !accumulate) return false;
  if (map->kernel_size != 3 || map->stride != 1 || map->K > PCMI_MAX_KERNEL_VOLUME || n_in != n_out || map->n_in != n_in) return false;
  if (cin < 32 || cin % 32 != 0 || cout % 32 != 0 || in_ld % 4 != 0 || gout_ld % 4 != 0 || (uintptr_t)in % 16 != 0 ||
      (uintptr_t)gout % 16 != 0)
    return false;
  if (n_in * in_ld * 4 > 0x7FFFFF00ll || n_out * gout_ld * 4 > 0x7FFFFF00ll) return false;
  if (wgrad_x3t_eligible(map, n_in, n_out, cin, cout, in_ld, gout_ld)) return false;
  const int64_t M = kmap_pairs_bound(*map);
  if (M <= 0 || b->g.n >= kWgradGroupMax) return false;
  WgradArgs a;
  int CT = 1, NT = 1;
  int64_t nchunks = 0;
  if (wgrad_plan(in, in_ld, n_in, cin, gout, gout_ld, n_out, cout, map, 0, gweight, accumulate, M, st, &a, &CT, &NT, &nchunks) != PCMI_OK)
    return false;
  if (!a.idx_x) return false;
  // In a group every offset is ONE chunk (the workgroup adds its tile straight into the flat gradient: no slabs, no
  // arrival counters): a single launch cuts a 2700-row offset into several chunks to get enough workgroups, a group has
  // them from its other layers.  Up to 4096 pairs per offset (the longest chunk a single launch may take, wgrad_max_chunk).
  const int64_t bound = wgrad_offset_bound(n_in, n_out);
  if (bound > wgrad_max_chunk()) return false;
  a.chunk = (int)align_up((size_t)bound, 128);
  a.mpk = 1;
  a.mode = kDirect;
  a.counters = nullptr;
  nchunks = a.K;
  if (b->g.n > 0 && (CT != b->CT || NT != b->NT)) return false;
  a.slabs = nullptr;
  const int64_t wgs = nchunks * (cin / (32 * CT)) * (cout / (32 * NT));
  if ((int64_t)b->g.first[b->g.n] + wgs > 0x3FFFFFFF) return false;
  b->CT = CT;
  b->NT = NT;
  b->g.job[b->g.n] = a;
  b->g.first[b->g.n + 1] = b->g.first[b->g.n] + (int)wgs;
  ++b->g.n;
  return true;
}

template <int CT>
static int launch_group_nt(int NT, const WgradGroup& g, unsigned grid, hipStream_t st) {
  switch (NT) {
    case 1: wgrad_mfma_group_kernel<CT, 1><<<grid, 256, 0, st>>>(g); break;
    case 2: wgrad_mfma_group_kernel<CT, 2><<<grid, 256, 0, st>>>(g); break;
    case 3: wgrad_mfma_group_kernel<CT, 3><<<grid, 256, 0, st>>>(g); break;
    default: set_error("wgrad group: bad NT %d", NT); return PCMI_ERR_INVALID;
  }
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int wgrad_group_flush(WgradGroupBuilder* b, hipStream_t st) {
  if (!b || b->g.n == 0) return PCMI_OK;
  const unsigned grid = (unsigned)b->g.first[b->g.n];
  int rc;
  switch (b->CT) {
    case 1: rc = launch_group_nt<1>(b->NT, b->g, grid, st); break;
    case 2: rc = launch_group_nt<2>(b->NT, b->g, grid, st); break;
    default: rc = launch_group_nt<3>(b->NT, b->g, grid, st); break;
  }
  b->g.n = 0;
  b->g.first[0] = 0;
  return rc;
}

int spconv_backward_weight_m32(const float* in, int64_t in_ld, int64_t n_in, int cin, const float* gout,
                               int64_t gout_ld, int64_t n_out, int cout, const pcmi_kmap_t* map, int transpose,
                               float* gweight, float* gbias, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  PCMI_REQUIRE(in && gout && gweight && cin > 0 && cout > 0, PCMI_ERR_INVALID, "spconv_bwd_weight: bad argument");
  const int K = map ? map->K : 1;
  int64_t M;
  if (map) {
    const int64_t mi = transpose ? map->n_out : map->n_in, mo = transpose ? map->n_in : map->n_out;
    PCMI_REQUIRE(mi == n_in && mo == n_out, PCMI_ERR_INVALID, "spconv_bwd_weight: rows do not match the map");
    M = kmap_pairs_bound(*map);  // exact once the map's counts are on the host; only sizing / early-outs use it
  } else {
    PCMI_REQUIRE(n_in == n_out, PCMI_ERR_INVALID, "spconv_bwd_weight: dense path needs n_in == n_out");
    M = n_in;
  }
  const int64_t per_k = (int64_t)cin * cout;
  if (M == 0) {
    if (!accumulate) {
      PCMI_HIP_CHECK(hipMemsetAsync(gweight, 0, sizeof(float) * K * per_k, st));
      if (gbias) PCMI_HIP_CHECK(hipMemsetAsync(gbias, 0, sizeof(float) * cout, st));
    }
    return PCMI_OK;
  }
  if (!transpose && !gbias && wgrad_x3t_eligible(map, n_in, n_out, cin, cout, in_ld, gout_ld) && in_ld % 4 == 0 &&
      gout_ld % 4 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)gout % 16 == 0)
    return wgrad_x3t_run(in, in_ld, gout, gout_ld, n_out, cin, cout, map, gweight, accumulate, ws, ws_bytes, st);
  // PCMI_STEM_WGRAD_MFMA=0: the pair-list form of the stem's gradient (A/B); read per call
  const bool stem_mfma = [] {
    const char* e = getenv("PCMI_STEM_WGRAD_MFMA");
    return !(e && e[0] == '0');
  }();
  if (stem_mfma && cin == 3 && K == 27 && map && map->nbr && !transpose && map->stride == 1 && n_in == n_out && cout % 32 == 0 &&
      !gbias) {
    const int64_t n_blocks = ceil_div(n_out, 64);
    int slabs_max = kStemSlabs;
    if (const char* e = getenv("PCMI_STEM_SLABS")) slabs_max = std::max(1, std::min(kStemSlabs, atoi(e)));  // (tuning; <= the workspace's 512)
    const int n_wg = (int)std::min<int64_t>(slabs_max, n_blocks);
    PCMI_REQUIRE(ws && ws_bytes >= (size_t)n_wg * K * per_k * sizeof(float), PCMI_ERR_WORKSPACE,
                 "spconv_bwd_weight (stem): workspace %zu < %zu bytes", ws_bytes, (size_t)n_wg * K * per_k * sizeof(float));
    stem_wgrad_mfma_kernel<3, 27><<<dim3((unsigned)n_wg, (unsigned)(cout / 32)), 256, 0, st>>>(in, in_ld, gout, gout_ld, map->nbr, n_out,
                                                                                            cout, n_blocks, (float*)ws);
    PCMI_LAUNCH_CHECK();
    stem_slab_reduce_kernel<<<(unsigned)ceil_div(K * per_k, 8), 256, 0, st>>>((const float*)ws, n_wg, K * per_k, gweight, accumulate);
    PCMI_LAUNCH_CHECK();
    return PCMI_OK;
  }
  WgradArgs a;
  int CT = 1, NT = 1;
  int64_t nchunks = 0;
  const bool stem = cin < 8;
  {
    const int rc = wgrad_plan(in, in_ld, n_in, cin, gout, gout_ld, n_out, cout, map, transpose, gweight, accumulate, M, st, &a, &CT,
                              &NT, &nchunks);
    if (rc) return rc;
  }
  if (a.mode != kSlabs && map && !accumulate)  // an offset without pairs gets no workgroup: its slice must read zero
    PCMI_HIP_CHECK(hipMemsetAsync(gweight, 0, sizeof(float) * K * per_k, st));
  const size_t slab_bytes = a.mode == kDirect ? 0 : (size_t)nchunks * per_k * sizeof(float);
  const size_t bias_bytes = gbias ? (size_t)1024 * cout * sizeof(float) : 0;
  PCMI_REQUIRE(ws && ws_bytes >= slab_bytes + bias_bytes, PCMI_ERR_WORKSPACE,
               "spconv_bwd_weight: workspace %zu < %zu bytes", ws_bytes, slab_bytes + bias_bytes);
  a.slabs = (float*)ws;
  const int64_t grid_x = nchunks;
  if (stem) {
    stem_wgrad_kernel<3><<<dim3((unsigned)nchunks), 256, 0, st>>>(a);
    PCMI_LAUNCH_CHECK();
  } else {
    dim3 grid((unsigned)grid_x, (unsigned)(cin / (32 * CT)), (unsigned)(cout / (32 * NT)));
    const bool buf_ok = [] {  // PCMI_WGRAD_BUF=0: always the 64-bit-address form (A/B, parity of the two; read per call)
      const char* e = getenv("PCMI_WGRAD_BUF");
      return !(e && e[0] == '0');
    }();
    const bool buf = buf_ok && n_in * in_ld * 4 <= 0x7FFFFF00ll && n_out * gout_ld * 4 <= 0x7FFFFF00ll;
    int rc;
    switch (CT) {
      case 1: rc = launch_wg_nt<1>(NT, a, grid, buf, st); break;
      case 2: rc = launch_wg_nt<2>(NT, a, grid, buf, st); break;
      default: rc = launch_wg_nt<3>(NT, a, grid, buf, st); break;
    }
    if (rc) return rc;
  }
  if (a.mode != kSlabs) {
    // gW is complete when the launch is
  } else if (nchunks > 16 * (int64_t)K)
    wgrad_reduce_kernel<8><<<dim3((unsigned)ceil_div(per_k, 32), (unsigned)K), 256, 0, st>>>(a.slabs, a.offs, a.mpk, M, a.chunk,
                                                                                          per_k, gweight, accumulate);
  else
    wgrad_reduce_kernel<1><<<dim3((unsigned)ceil_div(per_k, 256), (unsigned)K), 256, 0, st>>>(a.slabs, a.offs, a.mpk, M, a.chunk,
                                                                                           per_k, gweight, accumulate);
  PCMI_LAUNCH_CHECK();
  if (gbias) {
    float* part = (float*)((char*)ws + align_up(slab_bytes, 256));
    const int nblocks = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n_out, 64)));
    const int rows_per_block = (int)ceil_div(n_out, nblocks);
    colsum_partial_kernel<<<nblocks, 256, 0, st>>>(gout, gout_ld, n_out, cout, rows_per_block, part);
    PCMI_LAUNCH_CHECK();
    colsum_final_kernel<<<dim3((unsigned)ceil_div(cout, 4)), 256, 0, st>>>(part, nblocks, cout, gbias, accumulate);
    PCMI_LAUNCH_CHECK();
  }
  return PCMI_OK;
}

}  // namespace pcmi

extern "C" {


int pcmi_spconv_bwd_weight(const float* in, int64_t in_ld, int64_t n_in, int cin, const float* gout,
                           int64_t gout_ld, int64_t n_out, int cout, const pcmi_kmap_t* map, int transpose,
                           float* gweight, float* gbias, void* ws, size_t ws_bytes, pcmi_stream_t stream) {
  return spconv_backward_weight(in, in_ld, n_in, cin, gout, gout_ld, n_out, cout, map, transpose, gweight, gbias, 0, ws,
                                ws_bytes, as_stream(stream));
}

}  // extern "C"
