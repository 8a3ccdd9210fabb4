// Mask-sorted row order of a 3^3 / stride-1 kernel map.
//
// The output-stationary conv kernel issues the MFMAs of offset k for a 32-row wave group unless NONE of the 32
// rows has that neighbour.  In loader order (depth-image scan order) nearly every group has some row with every
// offset, so 27 offsets are issued where 17.1 are occupied on average: x1.51 redundant MFMA work (measured:
// SQ_INSTS_VALU_MFMA_MOPS_F32 = 1.575 x algorithmic).  Processing the rows in the order of their 27-bit
// occupancy mask puts rows with the same neighbourhood shape (same surface orientation) into the same group:
// x1.13 on the bench batch.  The sort is a radix sort (hipCUB, plan-time plumbing on the side stream);
// perm[i] = row processed at position i, nbr_perm[k][i] = nbr[k][perm[i]] keeps the table reads coalesced.
#include <hipcub/hipcub.hpp>

#include "internal.h"

namespace pcmi {

// key = (spatial chunk << K) | occupancy mask: rows are sorted by mask only INSIDE the contiguous row range that
// one XCD processes (see the tile swizzle in spconv.hip), so the loader's scan-order locality -- 98 % of the
// gathers of a 1/8 chunk stay inside that chunk -- keeps working for that XCD's L2.
__global__ void row_mask_kernel(const int32_t* __restrict__ nbr, int K, int64_t n, int64_t chunk_rows,
                                uint32_t* __restrict__ mask, int32_t* __restrict__ iota) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t m = 0;
  for (int k = 0; k < K; ++k) m |= (nbr[(int64_t)k * n + j] >= 0 ? 1u : 0u) << k;
  mask[j] = m | ((uint32_t)(j / chunk_rows) << K);
  iota[j] = (int32_t)j;
}

__global__ void permute_table_kernel(const int32_t* __restrict__ nbr, int K, int64_t n, const int32_t* __restrict__ perm,
                                     int32_t* __restrict__ nbr_perm) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)K * n) return;
  const int64_t k = idx / n, i = idx - k * n;
  nbr_perm[idx] = nbr[k * n + perm[i]];
}

// tile_mask[t] = OR of the masks of the 128 rows at sorted positions [128 t, 128 t + 128), cnt[t] = its popcount
__global__ __launch_bounds__(128) void tile_mask_kernel(const uint32_t* __restrict__ sorted_key, int K, int64_t n,
                                                        uint32_t* __restrict__ tile_mask, int32_t* __restrict__ cnt) {
  __shared__ uint32_t s_m[2];
  const int64_t i = (int64_t)blockIdx.x * 128 + threadIdx.x;
  uint32_t m = i < n ? (sorted_key[i] & ((1u << K) - 1u)) : 0u;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) m |= __shfl_xor(m, d, 64);
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = s_m[0] | s_m[1];
    tile_mask[blockIdx.x] = m;
    cnt[blockIdx.x] = __popc(m);
  }
}

// pref[t] = sum of cnt[0..t), pref[n_tiles] = total; one workgroup (a level has at most a few thousand tiles)
__global__ __launch_bounds__(1024) void tile_prefix_kernel(const int32_t* __restrict__ cnt, int64_t n_tiles,
                                                           int32_t* __restrict__ pref) {
  __shared__ int32_t s_sum[1024];
  const int t = threadIdx.x;
  const int64_t per = (n_tiles + 1023) / 1024;
  const int64_t b = (int64_t)t * per, e = b + per < n_tiles ? b + per : n_tiles;
  int32_t s = 0;
  for (int64_t i = b; i < e; ++i) s += cnt[i];
  s_sum[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan of the per-thread sums
    const int32_t v = t >= d ? s_sum[t - d] : 0;
    __syncthreads();
    s_sum[t] += v;
    __syncthreads();
  }
  int32_t run = t > 0 ? s_sum[t - 1] : 0;
  for (int64_t i = b; i < e; ++i) {
    pref[i] = run;
    run += cnt[i];
  }
  if (t == 1023) pref[n_tiles] = s_sum[1023];
}

// sorted_key: the keys sort_rows_by_mask left in mask_out.  cnt: scratch [n_tiles].
int tile_units(const uint32_t* sorted_key, int K, int64_t n, uint32_t* tile_mask, int32_t* cnt, int32_t* tile_pref,
               hipStream_t st) {
  const int64_t n_tiles = ceil_div(n, 128);
  if (n_tiles == 0) return PCMI_OK;
  tile_mask_kernel<<<dim3((unsigned)n_tiles), 128, 0, st>>>(sorted_key, K, n, tile_mask, cnt);
  PCMI_LAUNCH_CHECK();
  tile_prefix_kernel<<<1, 1024, 0, st>>>(cnt, n_tiles, tile_pref);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

size_t sort_rows_temp_bytes(int64_t n) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                           (const int32_t*)nullptr, (int32_t*)nullptr, (int)n, 0, 30, (hipStream_t)0);
  return bytes + 256;
}

// scratch: mask_in[n], mask_out[n] (uint32), iota[n] (int32), temp (sort_rows_temp_bytes(n))
int sort_rows_by_mask(const int32_t* nbr, int K, int64_t n, int64_t chunk_rows, uint32_t* mask_in, uint32_t* mask_out,
                      int32_t* iota, void* temp, size_t temp_bytes, int32_t* perm, int32_t* nbr_perm, hipStream_t st) {
  if (n == 0) return PCMI_OK;
  PCMI_REQUIRE(K <= 27 && ceil_div(n, chunk_rows) <= 8, PCMI_ERR_INVALID, "sort_rows: key does not fit 30 bits");
  row_mask_kernel<<<dim3((unsigned)ceil_div(n, 256)), 256, 0, st>>>(nbr, K, n, chunk_rows, mask_in, iota);
  PCMI_LAUNCH_CHECK();
  PCMI_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, mask_in, mask_out, iota, perm, (int)n, 0, K + 3, st));
  permute_table_kernel<<<dim3((unsigned)ceil_div((int64_t)K * n, 256)), 256, 0, st>>>(nbr, K, n, perm, nbr_perm);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

}  // namespace pcmi
