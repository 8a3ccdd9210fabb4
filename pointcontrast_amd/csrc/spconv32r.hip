// 32 -> 32 channel table convolutions (3^3 / stride 1 of block1, the 2^3 / stride-2 down-samplings of the first two
// levels; forward and backward-data) with the WEIGHTS RESIDENT in LDS.
//
// These launches are bound by the gathered rows, not by the matrix cores (0.36 GFLOP per 29 MB at 2^3, 1.3 GFLOP per
// 91 MB at 3^3: far left of the ridge).  The 128-row-tile kernels (spconv.hip) re-stage a 4 KiB weight block per
// (tile, offset) behind a workgroup barrier -- at 32 channels a step is only 8 MFMAs per row group, so the barrier and
// the staging dominate -- and need a 3-way split of the offsets plus a reduction pass to fill the chip at 40k rows
// (+38 % traffic).  Here
//   * all K weight slices (27 x 32 x 32 fp32 = 108 KiB, rows padded to 36 floats: conflict-free B fragments) are
//     loaded into LDS ONCE per workgroup; one workgroup of 16 waves per CU;
//   * a wave owns a 16-row group of the (mask-sorted) processing order at a time, with no barrier after the first:
//     it looks up its rows' neighbours for all offsets at once, compacts the occupied offsets into a wave-private LDS
//     list, and walks that list with the gathers of kDepth offsets in flight (raw buffer loads; an absent row is an
//     out-of-range offset that returns zeros);
//   * v_mfma_f32_16x16x4_f32 as in spconv16p_kernel: lane (i, kk) gathers the float4 channels 16 blk + 4 kk .. + 3 of
//     row i, step s contracts channels {16 blk + 4 kk + s}; fp32 throughout, offsets in table order -> deterministic.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "internal.h"
#include "spconv_args.h"
#include "x3_split.h"

namespace pcmi {

constexpr int kR32Waves = 12;  // waves per workgroup (one workgroup per CU: the weights take most of its LDS; 3 per SIMD = 170 registers)
constexpr int kR32LDB = 36;    // floats per staged weight row (32 + 4: 4 LDB = 16 mod 32 banks)
constexpr int kR32Depth = 8;   // offsets whose gathers a wave keeps in flight (a wave has ONE group at 40k rows: its time is rounds of kDepth gathers)

template <bool WT, int KMAX>
__global__ __launch_bounds__(kR32Waves * 64, 1) void spconv32r_kernel(ConvArgs a) {
  constexpr int LDB = kR32LDB, W = kR32Waves, D = kR32Depth;
  constexpr int QN = (KMAX + 3) / 4;  // table look-ups of a lane: offsets kk, kk + 4, ...
  constexpr uint32_t kAbsent = 0x80000000u;
  constexpr int kRsrcFlags = 0x00020000;  // raw buffer, 32-bit data format
  __shared__ __attribute__((aligned(16))) float s_w[KMAX * 32 * LDB];
  __shared__ uint32_t s_off[W][KMAX][16];  // per wave: byte offsets of the 16 rows' neighbours, occupied offsets only
  __shared__ int32_t s_ks[W][32];          // per wave: weight slice of the j-th occupied offset

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i = lane & 15, kk = lane >> 4;

  // ---- all weight slices -> LDS: B_ks[c][n] at s_w[(ks * 32 + c) * LDB + n].  The loads are issued here, all at once
  // (a load-store loop paid one L2 round trip per iteration: 7 in a row for 27 slices), and stored behind the first
  // group's table look-up and gathers, which do not depend on them.
  constexpr int WL = (KMAX * 256 + W * 64 - 1) / (W * 64);  // float4s of a thread
  float4 wreg[WL];
#pragma unroll
  for (int u = 0; u < WL; ++u) {
    const int e = t + u * (W * 64);
    const int ks = e >> 8, r = e & 255;
    wreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ks < a.K) {
      const float* wk = a.w + (int64_t)ks * a.w_kstride;
      wreg[u] = !WT ? *reinterpret_cast<const float4*>(wk + (int64_t)(r >> 3) * a.w_sc + (r & 7) * 4)   // memory [c][n]
                    : *reinterpret_cast<const float4*>(wk + (int64_t)(r >> 3) * a.w_sn + (r & 7) * 4);  // memory [n][c]
    }
  }
  auto store_weights = [&]() {
#pragma unroll
    for (int u = 0; u < WL; ++u) {
      const int e = t + u * (W * 64);
      const int ks = e >> 8, r = e & 255;
      if (ks < a.K) {
        if (!WT) {  // a float4 of n
          *reinterpret_cast<float4*>(&s_w[(ks * 32 + (r >> 3)) * LDB + (r & 7) * 4]) = wreg[u];
        } else {  // a float4 of c, scattered over four staged rows
          float* d = &s_w[(ks * 32 + (r & 7) * 4) * LDB + (r >> 3)];
          d[0] = wreg[u].x;
          d[LDB] = wreg[u].y;
          d[2 * LDB] = wreg[u].z;
          d[3 * LDB] = wreg[u].w;
        }
      }
    }
  };

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7FFFFFFF, kRsrcFlags);
  const uint32_t ld_bytes = (uint32_t)(a.x_ld * 4);
  const int64_t n_groups = (a.n_rows + 15) / 16;
  bool first = true;
  // group -> wave: wave w of workgroup b takes the groups w * G + b, (w + W) * G + b, ...  In the mask-sorted order
  // neighbouring groups have the same number of occupied offsets (8 .. 27); consecutive groups per workgroup gave one
  // CU all the heavy groups and another all the light ones.
  for (int64_t g = (int64_t)wave * gridDim.x + blockIdx.x;; g += (int64_t)gridDim.x * W) {
    const bool active = g < n_groups;  // (wave-uniform)
    if (!active && !first) break;
    const int64_t pos = g * 16 + i;
    const bool valid = active && pos < a.n_rows;
    // ---- the 16 rows' neighbours at every offset; occupied offsets (uniform mask) ------------------------------------
    int32_t nb[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int k = 4 * q + kk;
      nb[q] = (valid && k < a.K) ? a.nbr[(int64_t)k * a.n_rows + pos] : -1;
    }
    uint32_t occ = 0;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const uint64_t b = __ballot(nb[q] >= 0);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (4 * q + c < KMAX) occ |= (((b >> (16 * c)) & 0xFFFFull) != 0ull ? 1u : 0u) << (4 * q + c);
    }
    occ = (uint32_t)__builtin_amdgcn_readfirstlane((int)occ);
    const int cnt = __popc(occ);
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int k = 4 * q + kk;
      if (k < KMAX && ((occ >> k) & 1u)) {
        const int j = __popc(occ & ((1u << k) - 1u));
        s_off[wave][j][i] = nb[q] >= 0 ? (uint32_t)nb[q] * ld_bytes : kAbsent;
        if (i == 0) s_ks[wave][j] = a.wsel[k];
      }
    }
    // (wave-private lists: LDS operations of one wave complete in order, no barrier)
    v4f ab[D][2];
    auto issue = [&](int j, v4f (&dst)[2]) {
      const uint32_t off = (j < cnt ? s_off[wave][j][i] : kAbsent) + 16u * (uint32_t)kk;  // absent stays out of range
      dst[0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
      dst[1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, off + 64u, 0, 0));
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, ab[d]);
    if (first) {  // every wave of the workgroup passes here exactly once, with or without a group of its own
      store_weights();
      __syncthreads();
      first = false;
    }
    if (!active) break;
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    for (int j0 = 0; j0 < cnt; j0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int j = j0 + d;
        if (j < cnt) {  // (uniform; no memory loads inside: see below)
          const int ks = __builtin_amdgcn_readfirstlane(s_ks[wave][j]);
          const float* sb = s_w + (ks * 32 + 4 * kk) * LDB + i;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float av = ab[d][q >> 2][q & 3];
            const int row = 16 * (q >> 2) + (q & 3);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sb[row * LDB], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sb[row * LDB + 16], acc[1], 0, 0, 0);
          }
        }
        // the refill is issued on EVERY path (past the end of the list it asks for out-of-range offsets, which cost no
        // memory traffic): the number of loads in flight is then the same whatever the branch above did, and the wait
        // the compiler puts in front of a slot's first use is for THAT slot's loads only -- with the refill inside the
        // branch it waited for everything in flight (one memory round trip per offset instead of per kDepth offsets)
        issue(j + D, ab[d]);
      }
    }
    // ---- D[row = 4 kk + r][col = i] of the two 16x16 tiles -----------------------------------------------------------
    const int32_t orow_i = valid ? (a.perm ? a.perm[pos] : (int32_t)pos) : -1;
    const float b0 = a.bias ? a.bias[i] : 0.f, b1 = a.bias ? a.bias[16 + i] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int32_t orow = __shfl(orow_i, 4 * kk + r, 64);
      if (orow >= 0) {
        float* op = a.out + (int64_t)orow * a.out_ld + i;
        if (a.accumulate) {
          op[0] += acc[0][r] + b0;
          op[16] += acc[1][r] + b1;
        } else {
          op[0] = acc[0][r] + b0;
          op[16] = acc[1][r] + b1;
        }
      }
    }
  }
}

// PCMI_CONV32R: minimum number of output rows for this kernel (0 = never).  Read per call (parity tests compare both).
static int64_t conv32r_min_rows() {
  const char* e = getenv("PCMI_CONV32R");
  return e ? (int64_t)atoll(e) : (int64_t)8192;
}

bool conv32r_eligible(const ConvArgs& a, int64_t x_bytes) {
  const int64_t mr = conv32r_min_rows();
  return mr > 0 && a.n_rows >= mr && a.C == 32 && a.N == 32 && a.nbr && a.K > 1 && a.K <= 27 && x_bytes <= 0x7FFFFF00ll &&
         (a.w_sn == 1 || a.w_sc == 1);
}

int conv32r_launch(bool w_transposed, const ConvArgs& a, hipStream_t st) {
  const int64_t n_groups = ceil_div(a.n_rows, 16);
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(num_cu(), ceil_div(n_groups, kR32Waves)));
  if (a.K <= 8) {
    if (w_transposed)
      spconv32r_kernel<true, 8><<<grid, kR32Waves * 64, 0, st>>>(a);
    else
      spconv32r_kernel<false, 8><<<grid, kR32Waves * 64, 0, st>>>(a);
  } else {
    if (w_transposed)
      spconv32r_kernel<true, 27><<<grid, kR32Waves * 64, 0, st>>>(a);
    else
      spconv32r_kernel<false, 27><<<grid, kR32Waves * 64, 0, st>>>(a);
  }
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

}  // namespace pcmi
