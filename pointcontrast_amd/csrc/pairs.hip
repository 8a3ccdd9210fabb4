// Positive-pair selection of the PointInfoNCE step on the device (pc/lib/ddp_trainer.py:400-417):
//   q_unique, count = pos_pairs[:, 0].unique(return_counts=True)          (pos_pairs sorted by column 0: the loader's contract)
//   off  = floor(uniform * count)                                          (fp32 product of the host-side draws)
//   k_sel = pos_pairs[:, 1][off + cumsum(count) - count];  optional sub-sample [sampled_inds] of both
// The reference runs unique / cumsum / the gathers as torch CUDA kernels around host-side random draws; round 2 of this
// library did the whole selection on the host (1.7 ms of the enqueueing thread per 840k correspondences, ~8 ms at 1 cm).
// Here the correspondences are uploaded once and three small kernels find the run starts of column 0 (count per block ->
// one-workgroup scan -> write), a fourth picks the pairs; the host keeps only what consumes its random-number streams
// (torch.rand(n_unique), np.random.choice) -- for which it needs n_unique: pcmi_pairs_scan_host, one pass over column 0.
#include <algorithm>
#include <thread>
#include <vector>

#include "common.h"
#include "internal.h"

namespace pcmi {

constexpr int kRunItems = 8, kRunBlock = 256 * kRunItems;  // pairs per thread / per workgroup

__device__ inline int run_flag(const int32_t* __restrict__ pairs, int64_t i, int64_t P) {
  if (i >= P) return 0;
  return (i == 0 || pairs[2 * i] != pairs[2 * (i - 1)]) ? 1 : 0;
}

// block sum of per-thread counts; returns the exclusive prefix of this thread and the block total
__device__ inline int block_exclusive(int v, int* s_wave /* [4] */, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) base += s_wave[w];
    tot += s_wave[w];
  }
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(256) void runs_count_kernel(const int32_t* __restrict__ pairs, int64_t P, int32_t* __restrict__ block_counts) {
  __shared__ int s_wave[4];
  const int64_t i0 = (int64_t)blockIdx.x * kRunBlock + (int64_t)threadIdx.x * kRunItems;
  int c = 0;
#pragma unroll
  for (int e = 0; e < kRunItems; ++e) c += run_flag(pairs, i0 + e, P);
  int tot;
  (void)block_exclusive(c, s_wave, &tot);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = tot;
}

// exclusive scan of the block counts by one workgroup (<= a few thousand blocks); *n_runs = total
__global__ __launch_bounds__(256) void runs_scan_kernel(const int32_t* __restrict__ block_counts, int n_blocks, int32_t* __restrict__ block_offs,
                                                        int64_t* __restrict__ n_runs) {
  __shared__ int s_wave[4];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < n_blocks; b0 += 256) {
    const int b = b0 + (int)threadIdx.x;
    const int v = b < n_blocks ? block_counts[b] : 0;
    int tot;
    const int ex = block_exclusive(v, s_wave, &tot);
    const int carry = s_carry;
    if (b < n_blocks) block_offs[b] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) s_carry = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_runs = s_carry;
}

__global__ __launch_bounds__(256) void runs_write_kernel(const int32_t* __restrict__ pairs, int64_t P, const int32_t* __restrict__ block_offs,
                                                         int32_t* __restrict__ starts) {
  __shared__ int s_wave[4];
  const int64_t i0 = (int64_t)blockIdx.x * kRunBlock + (int64_t)threadIdx.x * kRunItems;
  int f[kRunItems], c = 0;
#pragma unroll
  for (int e = 0; e < kRunItems; ++e) {
    f[e] = run_flag(pairs, i0 + e, P);
    c += f[e];
  }
  int tot;
  int pos = block_offs[blockIdx.x] + block_exclusive(c, s_wave, &tot);
#pragma unroll
  for (int e = 0; e < kRunItems; ++e)
    if (f[e]) starts[pos++] = (int32_t)(i0 + e);
}

// j-th selected pair: run qi = sampled[j] (or j), key = the floor(u * count)-th pair of that run
__global__ __launch_bounds__(256) void pair_pick_kernel(const int32_t* __restrict__ pairs, int64_t P, const int32_t* __restrict__ starts,
                                                        int64_t nq, const float* __restrict__ uniform, const int64_t* __restrict__ sampled,
                                                        int64_t n_sel, int64_t* __restrict__ q_idx, int64_t* __restrict__ k_idx) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_sel) return;
  const int64_t qi = sampled ? sampled[j] : j;
  if (qi < 0 || qi >= nq) {  // (an index the host drew for a different n_unique: never with consistent arguments)
    q_idx[j] = k_idx[j] = 0;
    return;
  }
  const int64_t s = starts[qi], e = qi + 1 < nq ? (int64_t)starts[qi + 1] : P;
  // torch: floor(float32 uniform * count) with count converted to float32 (exact below 2^24), one rounding
  const int64_t off = (int64_t)floorf(__fmul_rn(uniform[qi], (float)(e - s)));
  q_idx[j] = pairs[2 * s];
  k_idx[j] = pairs[2 * min(s + off, P - 1) + 1];
}

}  // namespace pcmi

using namespace pcmi;

extern "C" {

size_t pcmi_pair_select_workspace_bytes(int64_t n_pairs) {
  const int64_t blocks = ceil_div(std::max<int64_t>(n_pairs, 1), kRunBlock);
  return align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(n_pairs, 1), 256) + 2 * align_up(sizeof(int32_t) * (size_t)blocks, 256) + 256;
}

int pcmi_pair_select(const int32_t* pairs, int64_t n_pairs, int64_t n_unique, const float* uniform, const int64_t* sampled,
                     int64_t n_sel, int64_t* q_idx, int64_t* k_idx, void* ws, size_t ws_bytes, pcmi_stream_t stream) {
  PCMI_REQUIRE(pairs && uniform && q_idx && k_idx && n_pairs > 0 && n_pairs < (1ll << 31) && n_unique > 0 && n_sel >= 0, PCMI_ERR_INVALID,
               "pair_select: bad argument");
  PCMI_REQUIRE(ws && ws_bytes >= pcmi_pair_select_workspace_bytes(n_pairs), PCMI_ERR_WORKSPACE, "pair_select: workspace too small");
  if (n_sel == 0) return PCMI_OK;
  hipStream_t st = as_stream(stream);
  const int64_t blocks = ceil_div(n_pairs, kRunBlock);
  char* p = (char*)ws;
  int32_t* starts = (int32_t*)p;
  p += align_up(sizeof(int32_t) * (size_t)n_pairs, 256);
  int32_t* block_counts = (int32_t*)p;
  p += align_up(sizeof(int32_t) * (size_t)blocks, 256);
  int32_t* block_offs = (int32_t*)p;
  p += align_up(sizeof(int32_t) * (size_t)blocks, 256);
  int64_t* n_runs = (int64_t*)p;
  runs_count_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(pairs, n_pairs, block_counts);
  PCMI_LAUNCH_CHECK();
  runs_scan_kernel<<<1, 256, 0, st>>>(block_counts, (int)blocks, block_offs, n_runs);
  PCMI_LAUNCH_CHECK();
  runs_write_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(pairs, n_pairs, block_offs, starts);
  PCMI_LAUNCH_CHECK();
  pair_pick_kernel<<<dim3((unsigned)ceil_div(n_sel, 256)), 256, 0, st>>>(pairs, n_pairs, starts, n_unique, uniform, sampled, n_sel, q_idx, k_idx);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

// One pass over column 0 of the host-side correspondences [n_pairs, 2]: the number of runs (= unique queries when the
// column is sorted) and whether it is sorted at all.  A few threads: the pass is memory-bound (8 bytes per pair).
int pcmi_pairs_scan_host(const int32_t* pairs_host, int64_t n_pairs, int64_t* n_runs_host, int* sorted_host) {
  PCMI_REQUIRE((pairs_host || n_pairs == 0) && n_pairs >= 0 && n_runs_host && sorted_host, PCMI_ERR_INVALID, "pairs_scan_host: bad argument");
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(4, n_pairs / (1 << 16)));
  std::vector<int64_t> runs(nt, 0);
  std::vector<int> bad(nt, 0);
  auto work = [&](int t) {
    const int64_t b = n_pairs * t / nt, e = n_pairs * (t + 1) / nt;
    int64_t r = 0;
    int unsorted = 0;
    for (int64_t i = std::max<int64_t>(b, 1); i < e; ++i) {
      const int32_t a = pairs_host[2 * (i - 1)], c = pairs_host[2 * i];
      r += a != c;
      unsorted |= c < a;
    }
    runs[t] = r;
    bad[t] = unsorted;
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  int64_t r = n_pairs > 0 ? 1 : 0;
  int unsorted = 0;
  for (int t = 0; t < nt; ++t) {
    r += runs[t];
    unsorted |= bad[t];
  }
  *n_runs_host = r;
  *sorted_host = unsorted ? 0 : 1;
  return PCMI_OK;
}

}  // extern "C"
